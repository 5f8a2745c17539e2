#!/bin/bash
# lease 47: the gradient fill beside the FORWARD on a capped grid (HBVX_EARLY_ZERO=1, HBVX_ZERO_BLOCKS)
set -e
mkdir -p gpurun_out
: > gpurun_out/r04_c47_fill.txt
for rnd in 1 2; do
for cfg in "0 0" "1 0" "1 64" "1 128" "1 256" "1 512" "1 2048"; do
  set -- $cfg
  echo "HBVX_EARLY_ZERO=$1 HBVX_ZERO_BLOCKS=$2" >> gpurun_out/r04_c47_fill.txt
  HBVX_EARLY_ZERO=$1 HBVX_ZERO_BLOCKS=$2 timeout -k 10 200 python tools/bench_configs.py cfg2 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(' ', j['config'], j['ms_per_step'], {k: v for k, v in j['kernel_ms'].items() if 'route' not in k and 'bfi' not in k})" >> gpurun_out/r04_c47_fill.txt
done; done
cat gpurun_out/r04_c47_fill.txt
