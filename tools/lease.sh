mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r05_gputier_7.log 2>&1
tail -3 gpurun_out/r05_gputier_7.log
grep -E "^(FAILED|ERROR)" gpurun_out/r05_gputier_7.log | cut -c1-200 | head
