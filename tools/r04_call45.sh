#!/bin/bash
# lease 45: randomised soaks on the final tree (not part of the test tiers)
mkdir -p gpurun_out
timeout -k 10 500 python tools/fuzz_parity.py 400 41 > gpurun_out/r04_soak_parity.txt 2>&1; echo "fuzz_parity rc=$?" >> gpurun_out/r04_soak_parity.txt
tail -4 gpurun_out/r04_soak_parity.txt | cut -c1-300
timeout -k 10 300 python tools/fuzz_adj.py 120 42 > gpurun_out/r04_soak_adj.txt 2>&1; echo "fuzz_adj rc=$?" >> gpurun_out/r04_soak_adj.txt
tail -3 gpurun_out/r04_soak_adj.txt | cut -c1-300
timeout -k 10 200 python tools/fuzz_lstm.py 80 43 > gpurun_out/r04_soak_lstm.txt 2>&1; echo "fuzz_lstm rc=$?" >> gpurun_out/r04_soak_lstm.txt
tail -2 gpurun_out/r04_soak_lstm.txt | cut -c1-300
timeout -k 10 200 python tools/fuzz_gage.py 60 44 > gpurun_out/r04_soak_gage.txt 2>&1; echo "fuzz_gage rc=$?" >> gpurun_out/r04_soak_gage.txt
tail -2 gpurun_out/r04_soak_gage.txt | cut -c1-300
