#!/bin/bash
# round 4, lease 10: the library built without the SLP vectorizer (packed fp32 issues at half rate on gfx950: the pairs
# buy nothing and cost register shuffles) against the regular build, every configuration
mkdir -p gpurun_out
AB_CFGS="cfg2 cfg2dyn cfg3 cfg4 cfg5 dmg hourly" timeout -k 10 700 python tools/ab_libs.py libhbvx_base.so libhbvx_noslp.so > gpurun_out/r04_ab_noslp.txt 2>&1
cat gpurun_out/r04_ab_noslp.txt
timeout -k 10 300 tools/ab_cfg5full.sh libhbvx_base.so libhbvx_noslp.so > gpurun_out/r04_ab_noslp_cfg5full.txt 2>&1; cat gpurun_out/r04_ab_noslp_cfg5full.txt
