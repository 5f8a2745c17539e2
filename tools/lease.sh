#!/bin/bash
# scratch lease script (rewritten per lease)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/bench_one.py dpl dplgraph --rounds 2 --steps 10 > gpurun_out/r05_dpl_graph3.jsonl 2> gpurun_out/r05_dpl_graph3.err
echo "bench rc $?"
cut -c1-200 gpurun_out/r05_dpl_graph3.jsonl
tail -3 gpurun_out/r05_dpl_graph3.err
