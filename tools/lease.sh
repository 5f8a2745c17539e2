mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py tests/test_hbv_adj.py -m gpu -q -p no:cacheprovider -k "persistent or fill or adj" > gpurun_out/r05_gputier_12.log 2>&1
tail -3 gpurun_out/r05_gputier_12.log
grep -E "^E  +" gpurun_out/r05_gputier_12.log | cut -c1-300 | head
python tools/bench_one.py cfg2 cfg2persist cfg2 cfg2persist --steps 20 > gpurun_out/r05_persist_ab.jsonl 2> gpurun_out/r05_persist_ab.err
python - <<'PY'
import json
for l in open('gpurun_out/r05_persist_ab.jsonl'):
    r=json.loads(l); k=r['kernel_ms']; print(r['config'], 'med',r['ms_median'],'mean',r['ms_mean_region'],'max',r['ms_max'],'mallocs',r['device_mallocs'], {a.replace('hbvx_',''):round(b,3) for a,b in k.items() if b>0.05})
PY
