#!/bin/bash
# lease 40: where the 3.8 GB gradient fill runs now that the adjoint is short: beside the adjoint (default) or beside the forward
set -e
mkdir -p gpurun_out
: > gpurun_out/r04_c40_fill.txt
for rnd in 1 2 3; do
for ez in 0 1; do
  echo "HBVX_EARLY_ZERO=$ez" >> gpurun_out/r04_c40_fill.txt
  HBVX_EARLY_ZERO=$ez timeout -k 10 200 python tools/bench_configs.py cfg2 cfg2dyn cfg4 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(' ', j['config'], j['ms_per_step'], {k: v for k, v in j['kernel_ms'].items() if 'route' not in k and 'bfi' not in k})" >> gpurun_out/r04_c40_fill.txt
done; done
cat gpurun_out/r04_c40_fill.txt
