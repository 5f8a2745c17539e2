#!/bin/bash
# round 4, lease 16: pipelined forward whose steppers store the trajectory themselves (no drainer waves), A/B + parity
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_hbv_adj.py -m gpu -x -q > gpurun_out/r04_parity_lease16.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r04_parity_lease16.log | cut -c1-200
AB_CFGS="cfg2 cfg2dyn cfg3 cfg4 dmg hourly" timeout -k 10 600 python tools/ab_libs.py libhbvx_drain.so libhbvx_base.so > gpurun_out/r04_ab_direct.txt 2>&1
cat gpurun_out/r04_ab_direct.txt
