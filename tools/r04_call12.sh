#!/bin/bash
# round 4, lease 12: the records of the round on the final tree: GPU tier, driver-style bench line, profile round (kernel stats,
# counter traffic, SQ counters)
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputier_final.log 2>&1; echo "gputier rc=$?" | tee -a gpurun_out/r04_gputier_final.log
tail -4 gpurun_out/r04_gputier_final.log | cut -c1-300
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_style.json 2> gpurun_out/r04_bench_driver_style.log; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_bench_driver_style.json') if l.startswith('{')][-1])
print('headline', round(d['ms_per_step'],3), d['roofline']['frac'], d['roofline']['avg_ms'])
for e in d['secondary']: print(e.get('config'), e.get('ms_per_step'), e.get('error'))
PY
PROFILE_SQ=1 timeout -k 10 1000 bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1; echo "profile rc=$?"
