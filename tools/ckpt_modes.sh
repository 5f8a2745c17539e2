for mode in "HBVX_CKPT_DAYS=0" "HBVX_CKPT_DAYS=8" "HBVX_CKPT_DAYS=8 HBVX_CKPT_BLOCKWISE=0" "HBVX_CKPT_DAYS=8 HBVX_CKPT_BLOCK=64" "HBVX_CKPT_DAYS=8 HBVX_CKPT_BLOCK=128" "HBVX_CKPT_DAYS=16 HBVX_CKPT_BLOCKWISE=0" "HBVX_CKPT_DAYS=4 HBVX_CKPT_BLOCKWISE=0"; do
  echo "== $mode"; env $mode python tools/bench_configs.py cfg5 cfg2 2>&1 | grep config | python -c "
import sys, json
for ln in sys.stdin:
    r=json.loads(ln); print(r['config'], r['ms_per_step'], {k:round(v,3) for k,v in r['kernel_ms'].items() if 'ward' in k and 'route' not in k})"
done
