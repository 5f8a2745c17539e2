#!/bin/bash
# scratch lease script (rewritten per lease)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "persistent or zero_fill or fill_schedules or gated or golden" > gpurun_out/r05_memo_tests.log 2>&1
echo "rc $?"; tail -4 gpurun_out/r05_memo_tests.log
timeout -k 10 300 python3 tools/bench_one.py cfg2dyn cfg2dynpersist dmg --steps 15 > gpurun_out/r05_memo_bench.jsonl 2> gpurun_out/r05_memo_bench.err
cut -c1-230 gpurun_out/r05_memo_bench.jsonl
