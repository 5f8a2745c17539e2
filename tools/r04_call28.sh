#!/bin/bash
# lease 28: where a reducer wave's tile goes (segment timing, probe build)
set -e
mkdir -p gpurun_out
timeout -k 10 300 python tools/pipe_probe.py > gpurun_out/r04_c28_probe_cfg2.txt 2>&1
grep -B2 -A26 "^traj" gpurun_out/r04_c28_probe_cfg2.txt
