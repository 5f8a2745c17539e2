#!/bin/bash
# lease 26: one-slot plan (fewer branches on the reducers' path): probe + A/B
set -e
mkdir -p gpurun_out
timeout -k 10 300 python tools/pipe_probe.py > gpurun_out/r04_c26_probe_cfg2.txt 2>&1
grep -A18 "^traj" gpurun_out/r04_c26_probe_cfg2.txt
AB_CFGS="cfg2" timeout -k 10 600 python tools/ab_libs.py libhbvx_base.so libhbvx_slots1.so libhbvx_prev.so > gpurun_out/r04_c26_ab.txt 2>&1
cat gpurun_out/r04_c26_ab.txt
