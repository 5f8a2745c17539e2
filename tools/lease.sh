#!/bin/bash
# scratch lease script (rewritten per lease): parity soak of the final library
mkdir -p gpurun_out
timeout -k 10 500 python3 tools/fuzz_parity.py 500 601 > gpurun_out/r05_soak2_default.txt 2>&1
echo "default rc $?"; tail -1 gpurun_out/r05_soak2_default.txt
HBVX_STREAM_MIN=1 timeout -k 10 500 python3 tools/fuzz_parity.py 500 602 > gpurun_out/r05_soak2_stream.txt 2>&1
echo "stream rc $?"; tail -1 gpurun_out/r05_soak2_stream.txt
grep -c "ERROR" gpurun_out/r05_soak2_default.txt gpurun_out/r05_soak2_stream.txt
