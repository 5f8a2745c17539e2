#!/bin/bash
# Profile bench.py on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag>
# Three rocprofv3 passes of bench.py: kernel trace + stats (the whole command: headline, secondary configurations,
# LSTM and training step), then the two TCC counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one
# pass), then the same two counters for cfg3 and the cfg5 share (tools/bench_configs.py).  Raw output under
# gpurun_out/prof_<tag>/; tools/profile_collect.py turns it into profiles/<tag>_*.
set -e
TAG=${1:-r02}
R=$PWD
export TMPDIR=/tmp
export HBVX_BENCH_TUNE_GEMM=0   # the tuner's trial launches would flood the kernel trace
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/kt -o kt -- python3 $R/bench.py --steps 5 --warmup 2 > $R/gpurun_out/prof_$TAG.bench.json 2> $R/gpurun_out/prof_$TAG.kt.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG/fetch -o fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_$TAG.fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG/write -o write -- python3 $R/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_$TAG.write.log
# HBM traffic of the cfg3 and cfg5-share kernels: the same two counters, one process per
# configuration so that the shared kernels (routing, zero fill) are attributed to the right one
for c in cfg3 cfg5; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG/fetch_$c -o fetch -- python3 $R/tools/bench_configs.py $c > /dev/null 2> $R/gpurun_out/prof_$TAG.fetch_$c.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG/write_$c -o write -- python3 $R/tools/bench_configs.py $c > /dev/null 2> $R/gpurun_out/prof_$TAG.write_$c.log
done
cd $R
# SQ counters (VALU busy, instruction mix) of the cfg2 / cfg3 / cfg5-share kernels, four passes each: PROFILE_SQ=1
if [ -n "$PROFILE_SQ" ]; then
bash tools/diag_pmc.sh ${TAG}_cfg3 cfg3 > gpurun_out/prof_$TAG.sq_cfg3.txt 2>&1 || true
bash tools/diag_pmc.sh ${TAG}_cfg5 cfg5 > gpurun_out/prof_$TAG.sq_cfg5.txt 2>&1 || true
bash tools/diag_pmc.sh ${TAG}_cfg2 cfg2 > gpurun_out/prof_$TAG.sq_cfg2.txt 2>&1 || true
fi
find gpurun_out/prof_$TAG -name "*.csv" | head -20
tail -1 gpurun_out/prof_$TAG.bench.json
