#!/usr/bin/env python3
"""Sequence LSTM: hydrodl2_amd.lstm.SeqLSTM (include/hbvx_lstm.h) beside torch.nn.LSTM (MIOpen) on the
same weights and input, forward and forward+backward, at the delta-MG shape by default.

    python tools/bench_lstm.py [T B I H] [--steps 20]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from hydrodl2_amd import ops  # noqa: E402
from hydrodl2_amd.lstm import SeqLSTM  # noqa: E402


def timed(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
    T, B, I, H = (nums + [730, 100, 256, 256])[:4] if len(nums) < 4 else nums[:4]
    steps = 20
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    fused = SeqLSTM(I, H).to(dev)
    ref = torch.nn.LSTM(I, H).to(dev)
    ref.load_state_dict(fused.state_dict())
    x = torch.randn(T, B, I, device=dev, requires_grad=True)
    gh = torch.randn(T, B, H, device=dev)

    def fwd(m):
        with torch.no_grad():
            return m(x)[0]

    def fwdbwd(m):
        for p in m.parameters():
            p.grad = None
        x.grad = None
        (m(x)[0] * gh).sum().backward()

    res = {"T": T, "B": B, "I": I, "H": H}
    for name, m in (("fused", fused), ("torch", ref)):
        res[f"{name}_fwd_ms"] = round(timed(lambda: fwd(m), steps), 3)
        res[f"{name}_fwdbwd_ms"] = round(timed(lambda: fwdbwd(m), steps), 3)
    ops.KERNEL_EVENTS = []
    for _ in range(steps):
        fwdbwd(fused)
    torch.cuda.synchronize()
    ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    for n in ("hbvx_lstm_forward", "hbvx_lstm_backward"):
        ms = sum(a.elapsed_time(b) for k, a, b in ev if k == n) / steps
        res[n + "_ms"] = round(ms, 3)
        res[n + "_us_per_step"] = round(ms * 1e3 / T, 2)
    # what bounds the recurrence kernels: MFMA work and hand-off traffic per launch against the chip's peaks
    tiles = (B + 15) // 16
    flops = 2.0 * T * tiles * 16 * H * 4 * H                   # padded rows; forward = backward
    res["mfma"] = {d: {"tflops": round(flops / (res[f"hbvx_lstm_{d}_ms"] * 1e-3) / 1e12, 2),
                       "frac_of_fp32_mfma_peak": round(flops / (res[f"hbvx_lstm_{d}_ms"] * 1e-3) / 157.3e12, 4)}
                   for d in ("forward", "backward")}
    wgs = H // 16
    xch = {"forward": T * tiles * wgs * H * 16 * 4, "backward": T * tiles * wgs * H * 16 * 4 * 4}
    res["hand_off_load_TBps"] = {d: round(xch[d] / (res[f"hbvx_lstm_{d}_ms"] * 1e-3) / 1e12, 3) for d in xch}
    res["hbm_algorithmic_GB"] = {"forward": round(T * B * H * 4 * (4 + 4 + 1 + 1) / 1e9, 3),
                                 "backward": round(T * B * H * 4 * (4 + 4 + 2 + 1) / 1e9, 3)}
    err = (fwd(fused) - fwd(ref)).abs().max().item()
    res["max_abs_diff_vs_torch"] = err
    print(json.dumps(res))


if __name__ == "__main__":
    main()
