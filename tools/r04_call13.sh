#!/bin/bash
# round 4, lease 13: when and how the gradient fill runs (cfg2, cfg2dyn, cfg4): beside the adjoint (default) or from the end of the
# forward kernel on; one store per thread or a persistent grid
mkdir -p gpurun_out
for rnd in 1 2; do
for after in 0 1; do for wgs in 0 512 2048; do
  echo "== FILL_AFTER_FWD=$after ZERO_WGS=$wgs round $rnd"
  HBVX_FILL_AFTER_FWD=$after HBVX_ZERO_WGS=$wgs timeout -k 10 200 python tools/bench_configs.py cfg2 cfg2dyn cfg4 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms']; print(d['config'], d['ms_per_step'], {a:b for a,b in k.items() if 'forward' in a or 'backward' in a or 'zero' in a})"
done; done; done > gpurun_out/r04_fill_schedule.txt 2>&1
cat gpurun_out/r04_fill_schedule.txt
( time python bench.py --steps 20 --warmup 5 > /dev/null 2>&1 ) 2>&1 | grep real
