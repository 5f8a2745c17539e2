#!/bin/bash
# round 4, lease 18: gradient fill of a tensor with few dynamic columns: dense + store gate (default) or around the stored columns without a gate
mkdir -p gpurun_out
for rnd in 1 2 3; do for g in 0 1; do
  echo "== HBVX_FILL_GAPS=$g round $rnd"
  HBVX_FILL_GAPS=$g timeout -k 10 200 python tools/bench_configs.py cfg2dyn cfg4 dmg 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms']; print(d['config'], d['ms_per_step'], {a:b for a,b in k.items() if 'forward' in a or 'backward' in a or 'zero' in a})"
done; done > gpurun_out/r04_fill_gaps.txt 2>&1
cat gpurun_out/r04_fill_gaps.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
