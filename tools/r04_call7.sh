#!/bin/bash
# round 4, lease 7: streaming adjoint with incremental descriptors (A/B), single-wave adjoint probe at cfg3, grid sweep
mkdir -p gpurun_out
AB_CFGS="cfg5" timeout -k 10 300 python tools/ab_libs.py libhbvx_armmul.so libhbvx_base.so > gpurun_out/r04_ab_arm.txt 2>&1; cat gpurun_out/r04_ab_arm.txt
timeout -k 10 300 tools/ab_cfg5full.sh libhbvx_armmul.so libhbvx_base.so > gpurun_out/r04_ab_arm_cfg5full.txt 2>&1; cat gpurun_out/r04_ab_arm_cfg5full.txt
echo "cfg3 with the serial tiled adjoint (one stepper wave per 64 lanes runs the whole day's adjoint):"
HBVX_BWD=tiled timeout -k 10 200 python tools/bench_configs.py cfg3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'])" | tee gpurun_out/r04_cfg3_tiled_adjoint.txt
timeout -k 10 900 python tools/grid_sweep.py 730 7300 > gpurun_out/r04_grid_sweep.jsonl 2> gpurun_out/r04_grid_sweep.log; echo "sweep rc=$?"; wc -l gpurun_out/r04_grid_sweep.jsonl; tail -3 gpurun_out/r04_grid_sweep.log
