mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05_gputier_14.log 2>&1
tail -3 gpurun_out/r05_gputier_14.log
grep -E "^(FAILED|ERROR)" gpurun_out/r05_gputier_14.log | cut -c1-200 | head
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; tail -1 gpurun_out/r05_smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
grep "\[bench\]" gpurun_out/r05_bench_final.err | tail -20
