#!/usr/bin/env python3
"""Time named workloads of bench.py (bench.WORKLOADS) with its own protocol -- W warm-up + K timed steps, per-step
HIP-event samples, per-launch kernel times -- one JSON line each.  For A/B runs under environment knobs:

    HBVX_CKPT_ONCHIP=0 python tools/bench_one.py cfg5full_ck4 cfg5full --steps 10
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="+")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=1, help="repeat the whole list (interleaved A/B)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    from hydrodl2_amd import _lib
    for rnd in range(args.rounds):
        for name in args.names:
            wl = bench.make_workload(name, dev, 7)
            dt, kms = bench.timed_steps(wl, args.steps, args.warmup, dev, 1)
            lib = _lib.get_library()
            smp = bench.timed_steps.samples
            rec = {"config": name, "round": rnd, "ms_median": smp["ms_median"], "ms_min": smp["ms_min"], "ms_max": smp["ms_max"],
                   "ms_mean_region": smp["ms_mean_region"], "host_enqueue_ms_median": smp["host_enqueue_ms_median"],
                   "kernel_ms": {k: round(v, 4) for k, v in kms.items()},
                   "dispatch": [lib.last_dispatch(0), lib.last_dispatch(1)],
                   "lane_steps_per_s": (wl.lane_steps / (smp["ms_median"] * 1e-3)) if hasattr(wl, "lane_steps") else None,
                   "device_mallocs": bench.timed_steps.device_mallocs,
                   "env": {k: v for k, v in os.environ.items() if k.startswith("HBVX_")}}
            print(json.dumps(rec), flush=True)
            del wl
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
