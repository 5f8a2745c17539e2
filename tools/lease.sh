mkdir -p gpurun_out
python tools/bench_one.py cfg2 cfg2graph cfg2 cfg2graph --steps 20 > gpurun_out/r05_cfg2graph.jsonl 2> gpurun_out/r05_cfg2graph.err
python - <<'PY'
import json
for l in open('gpurun_out/r05_cfg2graph.jsonl'):
    r=json.loads(l); print(r['config'], r['ms_median'], r['ms_min'], r['ms_max'], r['host_enqueue_ms_median'])
PY
tail -3 gpurun_out/r05_cfg2graph.err
