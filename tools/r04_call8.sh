#!/bin/bash
# round 4, lease 8: GPU tier on the current tree; shorter soil chain (A/B against the previous commit and the approximate quotient)
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputier_lease8.log 2>&1; echo "gputier rc=$?" | tee -a gpurun_out/r04_gputier_lease8.log
tail -6 gpurun_out/r04_gputier_lease8.log | cut -c1-600
AB_CFGS="cfg2 cfg2dyn cfg3 cfg5 dmg" timeout -k 10 500 python tools/ab_libs.py libhbvx_chain0.so libhbvx_base.so libhbvx_qapprox.so > gpurun_out/r04_ab_chain.txt 2>&1
cat gpurun_out/r04_ab_chain.txt
