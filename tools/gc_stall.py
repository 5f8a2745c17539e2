#!/usr/bin/env python3
"""What one pass of Python's cyclic collector costs a short timed region.

Round 4's driver run recorded cfg5share at 11.9 ms per step (mean of 5) against 2.99 ms of kernels: 44.6 ms that no
kernel accounts for, once, inside 5 steps.  This tool replays that protocol (3 warm-up + 5 timed steps, whole-region
mean, collector enabled, the workload built after another one was torn down) three ways and prints one JSON line
each:

    plain      collector enabled, nothing forced
    forced     a full collection (gc.collect()) in the middle of the third timed step's enqueue
    disabled   collector off inside the region (bench.py's protocol since round 5)

plus the cost of a full collection in this process on its own.  A full pass walks every tracked container of the
process (~170 000 with torch, numpy and the package imported): tens of ms of host time during which nothing is
enqueued; at a 3 ms step that is the whole difference.
"""
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


PASSES = []          # (generation, ms) of every collector pass, from gc.callbacks
_t = [0.0]


def _watch(phase, info):
    if phase == "start":
        _t[0] = time.perf_counter()
    else:
        PASSES.append((info["generation"], round(1e3 * (time.perf_counter() - _t[0]), 2)))


def region(wl, steps, force_at=None):
    torch.cuda.synchronize()
    del PASSES[:]
    per = []
    t0 = time.perf_counter()
    for i in range(steps):
        h0 = time.perf_counter()
        if force_at == i:
            gc.collect()
        wl.step()
        per.append(round(1e3 * (time.perf_counter() - h0), 3))
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, per


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg5share"
    dev = torch.device("cuda:0")
    gc.callbacks.append(_watch)
    prev = bench.Workload("cfg4joint", dev, seed=7)        # the workload the driver's run had just finished
    for _ in range(3):
        prev.step()
    torch.cuda.synchronize()
    del prev
    torch.cuda.empty_cache()
    wl = bench.Workload(name, dev, seed=7)
    for _ in range(3):
        wl.step()
    t0 = time.perf_counter()
    n = gc.collect()
    full_ms = 1e3 * (time.perf_counter() - t0)
    print(json.dumps({"full_collection_ms": round(full_ms, 2), "unreachable": n, "tracked": len(gc.get_objects()),
                      "thresholds": gc.get_threshold()}), flush=True)
    for mode in ("plain", "forced", "disabled", "plain"):
        if mode == "disabled":
            gc.disable()
        mean, per = region(wl, 5, force_at=2 if mode == "forced" else None)
        gc.enable()
        print(json.dumps({"config": name, "mode": mode, "ms_mean_of_5": round(mean, 3), "host_ms_per_step": per,
                          "collector_passes_gen_ms": list(PASSES),
                          "gc_counts_after": gc.get_count()}), flush=True)


if __name__ == "__main__":
    main()
