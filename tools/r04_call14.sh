#!/bin/bash
# round 4, lease 14: fused multiply-adds in the adjoint's own arithmetic (A/B) and the GPU tier on it
mkdir -p gpurun_out
AB_CFGS="cfg2 cfg2dyn cfg3 cfg4 cfg5 dmg hourly" timeout -k 10 700 python tools/ab_libs.py libhbvx_nofma.so libhbvx_base.so > gpurun_out/r04_ab_adjfma.txt 2>&1
cat gpurun_out/r04_ab_adjfma.txt
timeout -k 10 300 tools/ab_cfg5full.sh libhbvx_nofma.so libhbvx_base.so > gpurun_out/r04_ab_adjfma_cfg5full.txt 2>&1; cat gpurun_out/r04_ab_adjfma_cfg5full.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputier_lease14.log 2>&1; echo "gputier rc=$?" | tee -a gpurun_out/r04_gputier_lease14.log
tail -4 gpurun_out/r04_gputier_lease14.log | cut -c1-300; cat gpurun_out/fuzz_flips.txt | cut -c1-400
