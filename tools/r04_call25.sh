#!/bin/bash
# lease 25: upper bounds: pipelined forward without reducer / drainer work (wrong results; timing only)
set -e
mkdir -p gpurun_out
AB_CFGS="cfg2 hourly" timeout -k 10 600 python tools/ab_libs.py libhbvx_base.so libhbvx_nored.so libhbvx_nodrain.so libhbvx_noboth.so > gpurun_out/r04_c25_ab.txt 2>&1
cat gpurun_out/r04_c25_ab.txt
