#!/bin/bash
# round 4, lease 2: the driver-style bench line with the new secondary entries, then the profile round
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.log; echo "bench rc=$?"
tail -5 gpurun_out/r04_bench_a.log
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_bench_a.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['limited_by'])
for e in d['secondary']: print(e.get('config'), e.get('ms_per_step'), e.get('limited_by'), e.get('error'))
PY
timeout -k 10 700 bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1; echo "profile rc=$?"
tail -3 gpurun_out/r04_profile_round.log
