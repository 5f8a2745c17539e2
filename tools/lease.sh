mkdir -p gpurun_out
python -m pytest tests/test_graphed.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "graph or stream2 or long_golden or checkpointed" > gpurun_out/r05_gputier_3.log 2>&1
tail -3 gpurun_out/r05_gputier_3.log
HBVX_CKPT_ONCHIP=0 python tools/bench_one.py cfg5full_ck4 --steps 10 > gpurun_out/r05_ckpt_block.jsonl 2>> gpurun_out/r05_ckpt_ab.err
cut -c1-300 gpurun_out/r05_ckpt_block.jsonl
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_1.json 2> gpurun_out/r05_bench_1.err
tail -15 gpurun_out/r05_bench_1.err
