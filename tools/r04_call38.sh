#!/bin/bash
# lease 38: records on the current tree: driver-style bench line, kernel stats + HBM counters + SQ counters
set -e
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_lease38.json 2> gpurun_out/r04_bench_lease38.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_bench_lease38.json').read().strip().splitlines()[-1])
print('headline', round(d['ms_per_step'], 3), 'ms  frac', d['roofline']['frac'], ' kernel_ms', d['roofline']['whole_step']['kernel_ms'])
for s in d['secondary']:
    print(s.get('config'), s.get('ms_per_step'), s.get('kernel_ms'))
PY
PROFILE_SQ=1 timeout -k 10 900 bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1 || { tail -20 gpurun_out/r04_profile_round.log; exit 1; }
tail -3 gpurun_out/r04_profile_round.log | cut -c1-300
