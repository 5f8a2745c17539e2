#!/bin/bash
# lease 42: static-gradient accumulators of the adjoint sweep in LDS (13-19 registers fewer): parity + A/B
set -e
mkdir -p gpurun_out
cp hydrodl2_amd/csrc/libhbvx.so hydrodl2_amd/csrc/libhbvx_keep.so
cp hydrodl2_amd/csrc/libhbvx_glds.so hydrodl2_amd/csrc/libhbvx.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_mts.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r04_c42_tests.log 2>&1 || { tail -30 gpurun_out/r04_c42_tests.log; cp hydrodl2_amd/csrc/libhbvx_keep.so hydrodl2_amd/csrc/libhbvx.so; exit 1; }
tail -2 gpurun_out/r04_c42_tests.log
cp hydrodl2_amd/csrc/libhbvx_keep.so hydrodl2_amd/csrc/libhbvx.so
AB_CFGS="cfg2 cfg2dyn hourly dmg" timeout -k 10 600 python tools/ab_libs.py libhbvx_base.so libhbvx_glds.so > gpurun_out/r04_c42_ab.txt 2>&1
cat gpurun_out/r04_c42_ab.txt
