#!/bin/bash
# round 4, lease 11: instruction-scheduler strategies (max-ilp / iterative-ilp for the pipelined forward only, max-ilp everywhere)
mkdir -p gpurun_out
AB_CFGS="cfg2 cfg2dyn cfg3 cfg4 cfg5 dmg hourly" timeout -k 10 900 python tools/ab_libs.py libhbvx_base.so libhbvx_pmaxilp.so libhbvx_piterilp.so libhbvx_allmaxilp.so > gpurun_out/r04_ab_sched.txt 2>&1
cat gpurun_out/r04_ab_sched.txt
