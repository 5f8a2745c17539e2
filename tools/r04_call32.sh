#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 python tools/micro/wgrad_gemm.py > gpurun_out/r04_c32_wgrad.txt 2>&1
cat gpurun_out/r04_c32_wgrad.txt
