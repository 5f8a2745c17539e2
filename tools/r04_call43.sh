#!/bin/bash
# lease 43: kernel list of the hourly step (what fills the 0.5 ms between the named kernels)
set -e
mkdir -p gpurun_out
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hourly -o h -- python3 $R/tools/bench_configs.py hourly > $R/gpurun_out/r04_c43_hourly.json 2> $R/gpurun_out/r04_c43.log
cd $R
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_hourly/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/1e6:8.3f} ms  {r['Calls']:>5}  {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
print('total', tot/1e6)
PY
