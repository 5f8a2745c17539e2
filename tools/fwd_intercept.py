#!/usr/bin/env python3
"""Dev probe: fixed cost of the pipelined forward -- hbvx_forward's HIP-event time at several record lengths
(671 basins x 16, static parameters, trajectory kept), least-squares line t = a + b T."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from hydrodl2_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
Ts, ms = [], []
for T in (365, 730, 1825, 3650, 7300):
    wl = bench.Workload("cfg2", dev, seed=3, T=T)
    for _ in range(3):
        wl.step()
    ops.KERNEL_EVENTS = []
    for _ in range(10):
        wl.step()
    torch.cuda.synchronize()
    ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    v = sorted(a.elapsed_time(b) for n, a, b in ev if n == "hbvx_forward")
    Ts.append(T); ms.append(v[len(v) // 2])
    print(f"T={T:5d}  hbvx_forward {ms[-1] * 1e3:8.1f} us  ({ms[-1] * 1e6 / T:6.1f} ns/day)")
    del wl
    torch.cuda.empty_cache()
b, a = np.polyfit(Ts, ms, 1)
print(f"fit: {a * 1e3:.1f} us + {b * 1e6:.1f} ns/day")
