#!/bin/bash
# round 4, lease 20: seven (six) filler waves with many dynamic rows: parity, A/B at cfg3, probe
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r04_parity_lease20.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r04_parity_lease20.log | cut -c1-200
AB_CFGS="cfg3 grid:hbv_2:672:7300:16" timeout -k 10 300 python tools/ab_libs.py libhbvx_prev.so libhbvx_f7.so libhbvx_base.so > gpurun_out/r04_ab_fillers.txt 2>&1; cat gpurun_out/r04_ab_fillers.txt
PROBE_MODEL=hbv_1_1p:Hbv_1_1p PROBE_DYN=all timeout -k 10 300 python tools/pipe_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_pipe_probe_cfg3_b.txt; head -22 gpurun_out/r04_pipe_probe_cfg3_b.txt
