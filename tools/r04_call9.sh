#!/bin/bash
# round 4, lease 9: evaporation factor off the excess chain (A/B against lease 8's "before"), per-wave probe of the pipelined forward
mkdir -p gpurun_out
AB_CFGS="cfg2 cfg2dyn cfg3 cfg5 dmg" timeout -k 10 500 python tools/ab_libs.py libhbvx_chain0.so libhbvx_base.so > gpurun_out/r04_ab_chain2.txt 2>&1
cat gpurun_out/r04_ab_chain2.txt
timeout -k 10 300 python tools/pipe_probe.py > gpurun_out/r04_pipe_probe_cfg2.txt 2>&1; tail -45 gpurun_out/r04_pipe_probe_cfg2.txt
