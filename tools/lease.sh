#!/bin/bash
# scratch lease script (rewritten per lease): root cause of the round-4 outlier + the round's profile passes
set -o pipefail
mkdir -p gpurun_out
python3 tools/gc_stall.py cfg5share > gpurun_out/r05_gc_stall.jsonl 2> gpurun_out/r05_gc_stall.err && \
timeout -k 10 900 bash tools/profile_round.sh r05f > gpurun_out/r05f_profile.log 2>&1
echo "exit $?"
tail -5 gpurun_out/r05_gc_stall.jsonl
