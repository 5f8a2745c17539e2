mkdir -p gpurun_out
python tools/bench_one.py cfg2dyn cfg2dynpersist cfg2dyn cfg2dynpersist --steps 20 > gpurun_out/r05_persist_dyn_ab.jsonl 2> gpurun_out/r05_persist_ab.err
python - <<'PY'
import json
for l in open('gpurun_out/r05_persist_dyn_ab.jsonl'):
    r=json.loads(l); k=r['kernel_ms']; print(r['config'], 'med',r['ms_median'],'mean',r['ms_mean_region'],'max',r['ms_max'],'mallocs',r['device_mallocs'], {a.replace('hbvx_',''):round(b,3) for a,b in k.items() if b>0.05})
PY
