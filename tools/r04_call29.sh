#!/bin/bash
# lease 29: lean tile loops of reducers / drainers: parity, probe, A/B against the previous library
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_mts.py tests/test_hbv_adj.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r04_c29_tests.log 2>&1 || { tail -30 gpurun_out/r04_c29_tests.log; exit 1; }
tail -2 gpurun_out/r04_c29_tests.log
timeout -k 10 300 python tools/pipe_probe.py > gpurun_out/r04_c29_probe_cfg2.txt 2>&1
grep -A18 "^traj" gpurun_out/r04_c29_probe_cfg2.txt
AB_CFGS="cfg2 cfg2dyn cfg3 cfg4 hourly dmg" timeout -k 10 600 python tools/ab_libs.py libhbvx_prev.so libhbvx_base.so > gpurun_out/r04_c29_ab.txt 2>&1
cat gpurun_out/r04_c29_ab.txt
