mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_uh_routing.py tests/test_gpu_fullsize.py tests/test_mts.py tests/test_graphed.py tests/test_gpu_fuzz.py tests/test_api_and_abi.py -m gpu -q -p no:cacheprovider > gpurun_out/r05_gputier_2.log 2>&1
tail -3 gpurun_out/r05_gputier_2.log
python tools/bench_one.py cfg5share cfg5share_ck4 cfg5full cfg5full_ck4 cfg5full_ck8 --steps 10 --rounds 2 > gpurun_out/r05_ckpt_ab.jsonl 2> gpurun_out/r05_ckpt_ab.err
tail -4 gpurun_out/r05_ckpt_ab.jsonl | cut -c1-400
