#!/usr/bin/env python3
"""Host-side cost of one training step at the deltaMG minibatch shape (100 basins x 16, 365 + 365 days):
wall per step against the GPU time of its kernels, and a cProfile of the Python side.

    python tools/host_overhead.py [steps]
"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
name = sys.argv[2] if len(sys.argv) > 2 else "dmg"          # dmg | dmggraph
dev = torch.device("cuda:0")
wl = bench.Workload(name, dev, seed=7)
for _ in range(20):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    wl.step()
t_cpu = time.perf_counter() - t0          # host time to ENQUEUE the steps
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{name}: enqueue {1e3 * t_cpu / steps:.3f} ms/step, wall {1e3 * t_all / steps:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    wl.step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
