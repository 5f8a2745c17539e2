#!/bin/bash
# scratch lease script (rewritten per lease)
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/repro_case327.py > gpurun_out/r05_case327.txt 2>&1
echo "rc $?"; cat gpurun_out/r05_case327.txt | tail -40
