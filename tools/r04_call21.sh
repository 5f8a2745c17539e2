#!/bin/bash
# round 4, lease 21: straight-code ensemble sums for ensembles of 4 / 8 members (reducers of the pipelined / tiled forward): parity + A/B
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gage_route.py tests/test_mts.py -m gpu -x -q > gpurun_out/r04_parity_lease21.log 2>&1; echo "parity rc=$?"; tail -2 gpurun_out/r04_parity_lease21.log | cut -c1-200
AB_CFGS="grid:hbv:1344:7300:8 grid:hbv:2688:7300:4 grid:hbv_2:2688:7300:4 hourly" timeout -k 10 400 python tools/ab_libs.py libhbvx_prev.so libhbvx_base.so > gpurun_out/r04_ab_enssmall.txt 2>&1; cat gpurun_out/r04_ab_enssmall.txt
