#!/usr/bin/env python3
"""Randomised soak of the sequence LSTM (hydrodl2_amd.lstm.SeqLSTM on the GPU) against torch.nn.LSTM in
fp64, with the tolerances of tests/test_lstm.py:  python tools/fuzz_lstm.py [n_cases] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_lstm import _run  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for case in range(n):
        T = rng.choice([1, 2, 3, 9, 17, 40, 100, 365])
        B = rng.choice([1, 5, 15, 16, 17, 33, 100, 128, 129, 300, 700])
        I = rng.choice([1, 3, 8, 35, 64])
        H = rng.choice([64, 128, 256])
        env = rng.choice([{}, {}, {"HBVX_LSTM_UNITS": "16"}, {"HBVX_LSTM_WGS_PER_CU": str(rng.randint(1, 3))}])
        for k in ("HBVX_LSTM_UNITS", "HBVX_LSTM_WGS_PER_CU"):
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            _run("cuda", T, B, I, H, seed=rng.randint(0, 10 ** 6))
            status = "ok"
        except AssertionError as e:
            bad += 1
            status = "MISMATCH " + str(e)[:200]
        print(f"[{case:3d}] {status} T={T} B={B} I={I} H={H} {env}", flush=True)
    print(f"{n - bad}/{n} cases agree", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
