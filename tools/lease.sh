mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05_gputier_10.log 2>&1
tail -3 gpurun_out/r05_gputier_10.log
grep -E "^(FAILED|ERROR)" gpurun_out/r05_gputier_10.log | cut -c1-200 | head
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; tail -1 gpurun_out/r05_smoke.log
