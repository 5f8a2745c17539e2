#!/bin/bash
# lease 23: one-rank RCCL test + pipelined-forward probe on the current tree
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_rccl_world1.py -m gpu -x -q > gpurun_out/r04_c23_rccl.log 2>&1 || { tail -40 gpurun_out/r04_c23_rccl.log; exit 1; }
tail -2 gpurun_out/r04_c23_rccl.log
timeout -k 10 300 python tools/pipe_probe.py > gpurun_out/r04_c23_probe_cfg2.txt 2>&1
cat gpurun_out/r04_c23_probe_cfg2.txt
