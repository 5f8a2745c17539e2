#!/usr/bin/env python3
"""Dev micro-benchmark: the LSTM's weight-gradient GEMMs (dg^T x: [4H, T*B] x [T*B, I], fp32, T*B = 73 000) through the
library in several call shapes -- plain, transposed operands, split-K as a batched product."""
import torch

dev = torch.device("cuda:0")
TB, G, I = 73000, 1024, 256
dg = torch.randn(TB, G, device=dev)
x = torch.randn(TB, I, device=dev)
h = torch.randn(TB, I, device=dev)
ref = dg.t() @ x


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def splitk(S):
    a = dg.view(S, TB // S, G).transpose(1, 2)          # [S, G, K/S]
    return torch.bmm(a, x.view(S, TB // S, I)).sum(0)


cases = {
    "dg.t() @ x": lambda: dg.t() @ x,
    "(x.t() @ dg).t()": lambda: (x.t() @ dg).t(),
    "mm(out=) of x.t() @ dg": lambda: torch.mm(x.t(), dg),
    "einsum kg,ki->gi": lambda: torch.einsum("kg,ki->gi", dg, x),
}
for S in (4, 8, 10, 20, 40, 73, 100):
    if TB % S == 0:
        cases[f"split-K bmm S={S}"] = (lambda S=S: splitk(S))
xh = torch.cat([x, h], 1)
cases["both at once: dg.t() @ [x|h]"] = lambda: dg.t() @ xh
cases["both at once: ([x|h].t() @ dg)"] = lambda: xh.t() @ dg
flops = 2.0 * TB * G * I
for name, f in cases.items():
    ms = timeit(f)
    out = f()
    both = "both" in name
    if not both:
        o = out if out.shape == ref.shape else out.t()
        err = float((o - ref).abs().max() / ref.abs().max())
    else:
        err = float("nan")
    print(f"{name:36s} {ms:7.3f} ms  {flops * (2 if both else 1) / ms / 1e9:7.1f} TFLOP/s  rel err {err:.1e}")
