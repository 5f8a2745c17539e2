#!/bin/bash
# round 4, lease 5: which parts of the new soil-moisture solve pay on the GPU (cfg4 forward), and the timing floor
mkdir -p gpurun_out
AB_CFGS="cfg4" timeout -k 10 600 python tools/ab_libs.py libhbvx_oldsoil.so libhbvx_base.so libhbvx_vkink.so libhbvx_vlin.so libhbvx_vkinklin.so libhbvx_vhh.so libhbvx_vklfma.so libhbvx_voldfma.so > gpurun_out/r04_ab_soil_parts.txt 2>&1
cat gpurun_out/r04_ab_soil_parts.txt
echo "floor: exactly one update per day (newton_max_iter = 0), old solve and new"
cp hydrodl2_amd/csrc/libhbvx.so /tmp/keep.so
for l in libhbvx_oldsoil.so libhbvx_vkinklin.so; do cp hydrodl2_amd/csrc/$l hydrodl2_amd/csrc/libhbvx.so; HBVX_CFG4_MAXITER=0 python tools/bench_configs.py cfg4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', d['ms_per_step'], d['kernel_ms'])"; done
cp /tmp/keep.so hydrodl2_amd/csrc/libhbvx.so
