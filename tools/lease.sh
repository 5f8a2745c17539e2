#!/bin/bash
# scratch lease script (rewritten per lease): the large-tensor tests with the in-launch fill forced on for every model
mkdir -p gpurun_out
HBVX_EARLY_ZERO=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py tests/test_graphed.py tests/test_hbv_adj.py -m gpu -x -q > gpurun_out/r05_earlyzero1_tests.log 2>&1
echo "rc $?"; tail -4 gpurun_out/r05_earlyzero1_tests.log
