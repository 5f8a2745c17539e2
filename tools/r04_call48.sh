#!/bin/bash
# lease 48: days per tile of the pipelined forward (one barrier per tile): 6 / 8 / 10
set -e
mkdir -p gpurun_out
AB_CFGS="cfg2 cfg2dyn dmg" timeout -k 10 600 python tools/ab_libs.py libhbvx_base.so libhbvx_kt10.so libhbvx_kt6.so > gpurun_out/r04_c48_ab.txt 2>&1
cat gpurun_out/r04_c48_ab.txt
