#!/bin/bash
# lease 44: at least three waves per SIMD for the chunk kernels (the hourly sweep: 181 -> 168 registers with 16 spilled)
set -e
mkdir -p gpurun_out
AB_CFGS="hourly cfg5 cfg3" timeout -k 10 600 python tools/ab_libs.py libhbvx_base.so libhbvx_minw3.so > gpurun_out/r04_c44_ab.txt 2>&1
cat gpurun_out/r04_c44_ab.txt
