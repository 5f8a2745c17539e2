#!/bin/bash
# Profile bench.py on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag>
# Three rocprofv3 passes of bench.py: kernel trace + stats, then the two TCC counters in
# their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Raw output under
# gpurun_out/prof_<tag>/; tools/profile_collect.py turns it into profiles/<tag>_*.
set -e
TAG=${1:-r02}
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/kt -o kt -- python3 $R/bench.py --steps 5 --warmup 2 > $R/gpurun_out/prof_$TAG.bench.json 2> $R/gpurun_out/prof_$TAG.kt.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG/fetch -o fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_$TAG.fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG/write -o write -- python3 $R/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_$TAG.write.log
# the kernels of the secondary configurations, one stats pass
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/cfgs -o cfgs -- python3 $R/tools/bench_configs.py cfg2dyn cfg3 cfg4 cfg5 dmg > $R/gpurun_out/prof_$TAG.cfgs.jsonl 2> $R/gpurun_out/prof_$TAG.cfgs.log
# the caller side: sequence LSTM beside torch's, and the end-to-end training step
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/lstm -o lstm -- python3 $R/tools/bench_lstm.py > $R/gpurun_out/prof_$TAG.lstm.json 2> $R/gpurun_out/prof_$TAG.lstm.log
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/dpl -o dpl -- python3 $R/examples/train_dpl.py --lstm fused --steps 10 > $R/gpurun_out/prof_$TAG.dpl.json 2> $R/gpurun_out/prof_$TAG.dpl.log
cd $R
# SQ counters (VALU busy, instruction mix) of the cfg3 and cfg5-share kernels: four passes each
bash tools/diag_pmc.sh ${TAG}_cfg3 cfg3 > gpurun_out/prof_$TAG.sq_cfg3.txt 2>&1 || true
bash tools/diag_pmc.sh ${TAG}_cfg5 cfg5 > gpurun_out/prof_$TAG.sq_cfg5.txt 2>&1 || true
bash tools/diag_pmc.sh ${TAG}_cfg2 cfg2 > gpurun_out/prof_$TAG.sq_cfg2.txt 2>&1 || true
find gpurun_out/prof_$TAG -name "*.csv" | head -20
tail -1 gpurun_out/prof_$TAG.bench.json
