#!/bin/bash
# lease 37: no arithmetic on freshly issued loads in the adjoint kernels (prefetch restored): A/B, then the GPU tier
set -e
mkdir -p gpurun_out
AB_CFGS="cfg2 cfg2dyn cfg3 cfg4 cfg5 hourly dmg" timeout -k 10 800 python tools/ab_libs.py libhbvx_prev.so libhbvx_base.so > gpurun_out/r04_c37_ab.txt 2>&1
cat gpurun_out/r04_c37_ab.txt
timeout -k 10 300 tools/ab_cfg5full.sh libhbvx_prev.so libhbvx_base.so > gpurun_out/r04_c37_ab_cfg5full.txt 2>&1 || true
cat gpurun_out/r04_c37_ab_cfg5full.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputier_lease37.log 2>&1; rc=$?
echo "gputier rc=$rc" | tee -a gpurun_out/r04_gputier_lease37.log
tail -4 gpurun_out/r04_gputier_lease37.log | cut -c1-300
