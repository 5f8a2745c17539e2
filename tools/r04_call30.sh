#!/bin/bash
# lease 30: waves per workgroup of the pipelined forward with lean helpers (config 2)
set -e
mkdir -p gpurun_out
: > gpurun_out/r04_c30_threads.txt
for rnd in 1 2 3; do
for th in 1024 512 640 768 896; do
  echo "threads $th" >> gpurun_out/r04_c30_threads.txt
  HBVX_PIPE_THREADS=$th timeout -k 10 120 python tools/bench_configs.py cfg2 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['config'], j['ms_per_step'], {k: v for k, v in j['kernel_ms'].items() if 'forward' in k})" >> gpurun_out/r04_c30_threads.txt
done; done
cat gpurun_out/r04_c30_threads.txt
