#!/bin/bash
# scratch lease script (rewritten per lease): full GPU tier, smoke, driver-style bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_gputier_16.log 2>&1
rc=$?
tail -4 gpurun_out/r05_gputier_16.log
[ $rc -eq 0 ] && python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1 && tail -1 gpurun_out/r05_smoke.log && \
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_final2.json 2> gpurun_out/r05_bench_final2.err
echo "rc $?"
tail -16 gpurun_out/r05_bench_final2.err
