#!/bin/bash
# round 4, lease 19: per-wave probe of the pipelined forward at config 3 (14 dynamic rows) and config 2 with 2 dynamic
mkdir -p gpurun_out
PROBE_MODEL=hbv_1_1p:Hbv_1_1p PROBE_DYN=all timeout -k 10 300 python tools/pipe_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_pipe_probe_cfg3.txt; head -24 gpurun_out/r04_pipe_probe_cfg3.txt
PROBE_DYN=parBETA,parBETAET timeout -k 10 300 python tools/pipe_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_pipe_probe_cfg2dyn.txt; head -24 gpurun_out/r04_pipe_probe_cfg2dyn.txt
