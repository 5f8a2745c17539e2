mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py tests/test_hbv_adj.py tests/test_graphed.py -m gpu -q -p no:cacheprovider -k "persistent or adj or Adj" > gpurun_out/r05_gputier_13.log 2>&1
tail -3 gpurun_out/r05_gputier_13.log
grep -E "^E  +" gpurun_out/r05_gputier_13.log | cut -c1-300 | head
python tools/bench_one.py cfg4 cfg4persist cfg4 cfg4persist --steps 20 > gpurun_out/r05_persist_adj_ab.jsonl 2> gpurun_out/r05_persist_ab.err
python - <<'PY'
import json
for l in open('gpurun_out/r05_persist_adj_ab.jsonl'):
    r=json.loads(l); k=r['kernel_ms']; print(r['config'], 'med',r['ms_median'],'mean',r['ms_mean_region'],'max',r['ms_max'],'host',r['host_enqueue_ms_median'], {a.replace('hbvx_',''):round(b,3) for a,b in k.items() if b>0.05})
PY
