#!/bin/bash
# lease 36: two raw-value sets alone (global-form loads) against the previous library and against both changes
set -e
mkdir -p gpurun_out
AB_CFGS="cfg2 cfg2dyn cfg3 hourly dmg" timeout -k 10 700 python tools/ab_libs.py libhbvx_prev.so libhbvx_gld.so libhbvx_base.so > gpurun_out/r04_c36_ab.txt 2>&1
cat gpurun_out/r04_c36_ab.txt
