#!/bin/bash
# round 4, lease 15: one-source addressing of the all-dynamic chunk kernels (cfg3), A/B + the parity cases that use them
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not cfg5_full" > gpurun_out/r04_parity_lease15.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r04_parity_lease15.log | cut -c1-200
AB_CFGS="cfg3" timeout -k 10 300 python tools/ab_libs.py libhbvx_prev.so libhbvx_base.so > gpurun_out/r04_ab_onesrc.txt 2>&1
cat gpurun_out/r04_ab_onesrc.txt
