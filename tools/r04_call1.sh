#!/bin/bash
# round 4, lease 1: GPU tier on the no-aux build, then A/B against the saved-powers build
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputier_lease1.log 2>&1; echo "gputier rc=$?" | tee -a gpurun_out/r04_gputier_lease1.log
tail -3 gpurun_out/r04_gputier_lease1.log
AB_CFGS="cfg2 cfg2dyn cfg3 cfg4 cfg5 dmg" timeout -k 10 600 python tools/ab_libs.py libhbvx_base.so libhbvx_savepow.so > gpurun_out/r04_ab_savepow.txt 2>&1
cat gpurun_out/r04_ab_savepow.txt
timeout -k 10 400 tools/ab_cfg5full.sh libhbvx_base.so libhbvx_savepow.so > gpurun_out/r04_ab_savepow_cfg5full.txt 2>&1
cat gpurun_out/r04_ab_savepow_cfg5full.txt
