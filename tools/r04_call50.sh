#!/bin/bash
# lease 50: ten-day tiles for the pipelined forward without dynamic parameters: GPU tier + bench line
set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputier_lease50.log 2>&1; rc=$?
echo "gputier rc=$rc" | tee -a gpurun_out/r04_gputier_lease50.log
tail -4 gpurun_out/r04_gputier_lease50.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_lease50.json 2> gpurun_out/r04_bench_lease50.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_bench_lease50.json').read().strip().splitlines()[-1])
print('headline', round(d['ms_per_step'], 3), 'ms  frac', d['roofline']['frac'], ' kernel_ms', d['roofline']['whole_step']['kernel_ms'])
for s in d['secondary']:
    print(s.get('config'), s.get('ms_per_step'), s.get('kernel_ms'))
PY
