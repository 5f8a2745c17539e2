#!/bin/bash
# lease 22: few-mode implicit adjoint: parity + A/B against the previous library
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_hbv_adj.py -m gpu -x -q > gpurun_out/r04_c22_tests.log 2>&1 || { tail -30 gpurun_out/r04_c22_tests.log; exit 1; }
tail -2 gpurun_out/r04_c22_tests.log
AB_CFGS="cfg4" timeout -k 10 400 python tools/ab_libs.py libhbvx_prev.so libhbvx_base.so > gpurun_out/r04_c22_ab.txt 2>&1
cat gpurun_out/r04_c22_ab.txt
