#!/bin/bash
# lease 33: split-K weight gradients of the sequence LSTM: parity and timing
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_lstm.py tests/test_example_dpl.py -m gpu -x -q > gpurun_out/r04_c33_tests.log 2>&1 || { tail -30 gpurun_out/r04_c33_tests.log; exit 1; }
tail -2 gpurun_out/r04_c33_tests.log
timeout -k 10 300 python tools/bench_lstm.py > gpurun_out/r04_c33_lstm.txt 2>&1 || true
cut -c1-600 gpurun_out/r04_c33_lstm.txt
