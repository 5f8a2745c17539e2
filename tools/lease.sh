#!/bin/bash
# scratch lease script (rewritten per lease): driver-style bench + the round's profile passes
set -o pipefail
mkdir -p gpurun_out
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_final3.json 2> gpurun_out/r05_bench_final3.err && \
timeout -k 10 600 bash tools/profile_round.sh r05g > gpurun_out/r05g_profile.log 2>&1
echo "rc $?"
grep "^\[bench\]" gpurun_out/r05_bench_final3.err | tail -18
du -sh gpurun_out
