#!/usr/bin/env python3
"""Dev probe: where the WALL time of a bench step goes that the GPU timeline does not show -- host time of the
forward / loss / backward calls with a device sync after each, and the caching allocator's device mallocs per step.

    python tools/step_phases.py cfg5 [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
wl = bench.Workload(name, dev, seed=7)
sync = torch.cuda.synchronize
for i in range(steps):
    st0 = torch.cuda.memory_stats()
    sync(); t0 = time.perf_counter()
    for leaf in wl.leaves:
        leaf.grad = None
    out = wl.model(wl.xd, wl.params)
    ta = time.perf_counter(); sync(); t1 = time.perf_counter()
    loss = (out[wl.key] * wl.w).sum()
    sync(); t2 = time.perf_counter()
    loss.backward()
    tb = time.perf_counter(); sync(); t3 = time.perf_counter()
    del out, loss
    sync(); t4 = time.perf_counter()
    st1 = torch.cuda.memory_stats()
    print(f"step {i}: forward {1e3*(t1-t0):7.2f} ms (host {1e3*(ta-t0):6.2f})  loss {1e3*(t2-t1):5.2f}  "
          f"backward {1e3*(t3-t2):7.2f} (host {1e3*(tb-t2):6.2f})  free {1e3*(t4-t3):5.2f}   "
          f"device mallocs {st1['num_device_alloc'] - st0['num_device_alloc']}  frees {st1['num_device_free'] - st0['num_device_free']}  "
          f"retries {st1['num_alloc_retries'] - st0['num_alloc_retries']}  reserved {st1['reserved_bytes.all.current']/1e9:.1f} GB", flush=True)
