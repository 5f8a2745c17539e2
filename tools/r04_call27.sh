#!/bin/bash
# lease 27: which part of a reducer pass costs the forward its 16 %: the LDS reads or the store (wrong results; timing only)
set -e
mkdir -p gpurun_out
AB_CFGS="cfg2" timeout -k 10 600 python tools/ab_libs.py libhbvx_base.so libhbvx_rednostore.so libhbvx_rednoread.so libhbvx_nored.so > gpurun_out/r04_c27_ab.txt 2>&1
cat gpurun_out/r04_c27_ab.txt
