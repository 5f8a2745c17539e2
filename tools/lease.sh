mkdir -p gpurun_out
python tools/bench_one.py cfg2 cfg2dyn cfg4 cfg5full --steps 20 --rounds 2 > gpurun_out/r05_earlyzero_default2.jsonl 2> gpurun_out/r05_earlyzero.err
HBVX_EARLY_ZERO=0 python tools/bench_one.py cfg2 cfg2dyn cfg4 cfg5full --steps 20 --rounds 2 > gpurun_out/r05_earlyzero_off3.jsonl 2>> gpurun_out/r05_earlyzero.err
python - <<'PY'
import json
for f in ('default2','off3'):
    for l in open(f'gpurun_out/r05_earlyzero_{f}.jsonl'):
        r=json.loads(l); k=r['kernel_ms']; print(f, r['config'], 'med',r['ms_median'],'mean',r['ms_mean_region'],'max',r['ms_max'],'mallocs',r['device_mallocs'], {a.replace('hbvx_',''):round(b,3) for a,b in k.items() if b>0.3})
PY
