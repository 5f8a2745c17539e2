#!/bin/bash
# lease 46: four chunks in flight in the chunk scan: parity + A/B
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_hbv_adj.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r04_c46_tests.log 2>&1 || { tail -30 gpurun_out/r04_c46_tests.log; exit 1; }
tail -2 gpurun_out/r04_c46_tests.log
AB_CFGS="cfg2 cfg2dyn cfg3 cfg4 hourly dmg" timeout -k 10 700 python tools/ab_libs.py libhbvx_prev.so libhbvx_base.so > gpurun_out/r04_c46_ab.txt 2>&1
cat gpurun_out/r04_c46_ab.txt
