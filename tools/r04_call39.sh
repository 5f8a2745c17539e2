#!/bin/bash
# lease 39: occupancy targets (waves per SIMD) for the chunk-parallel adjoint kernels
set -e
mkdir -p gpurun_out
AB_CFGS="cfg2 cfg2dyn cfg3 hourly dmg" timeout -k 10 900 python tools/ab_libs.py libhbvx_base.so libhbvx_occ5.so libhbvx_occ6.so libhbvx_occ8.so > gpurun_out/r04_c39_ab.txt 2>&1
cat gpurun_out/r04_c39_ab.txt
