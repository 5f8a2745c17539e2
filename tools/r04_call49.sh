#!/bin/bash
# lease 49: longer tiles for the pipelined forward WITHOUT dynamic parameters (8 / 10 / 11 days), six rounds
set -e
mkdir -p gpurun_out
for k in 1 2; do
AB_CFGS="cfg2" timeout -k 10 400 python tools/ab_libs.py libhbvx_base.so libhbvx_kt10s.so libhbvx_kt11s.so >> gpurun_out/r04_c49_ab.txt 2>&1
done
cat gpurun_out/r04_c49_ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
