mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -k "ensemble_weights or muwts or hip_matches_oracle or fuzz or variants" > gpurun_out/r05_gputier_9.log 2>&1
tail -3 gpurun_out/r05_gputier_9.log
grep -E "^(FAILED|ERROR)" gpurun_out/r05_gputier_9.log | cut -c1-200 | head
python tools/bench_one.py cfg2 cfg2mu cfg2dyn --steps 10 --rounds 2 > gpurun_out/r05_mu_ab2.jsonl 2> gpurun_out/r05_mu_ab.err
python - <<'PY'
import json
for l in open('gpurun_out/r05_mu_ab2.jsonl'):
    r=json.loads(l); print(r['config'], r['ms_median'], r['kernel_ms'].get('hbvx_forward'), r['kernel_ms'].get('hbvx_backward'), r['dispatch'])
PY
