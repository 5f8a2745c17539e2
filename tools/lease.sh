#!/bin/bash
# scratch lease script (rewritten per lease): full GPU tier, smoke, driver-style bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_gputier_19.log 2>&1
rc=$?
tail -3 gpurun_out/r05_gputier_19.log
[ $rc -eq 0 ] && python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1 && tail -1 gpurun_out/r05_smoke.log && \
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_final4.json 2> gpurun_out/r05_bench_final4.err
echo "rc $?"
grep "^\[bench\]" gpurun_out/r05_bench_final4.err | tail -6
du -sh gpurun_out
