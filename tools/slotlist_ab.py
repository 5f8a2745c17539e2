#!/usr/bin/env python3
"""Run-time slot lists against the compiled dynamic sets of the streaming kernels: forward /
adjoint call times at W = 4096 wavefronts of state (16 384 basins x 16 members, 730 days) for sibling dynamic sets.

    python tools/slotlist_ab.py [--steps 10]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

CASES = {
    # name: (model file, class, T, B, M, dynamic set)
    "hbv_static": ("hbv", "Hbv", 730, 16384, 16, []),
    "hbv_BETA_BETAET(compiled)": ("hbv", "Hbv", 730, 16384, 16, ["parBETA", "parBETAET"]),
    "hbv_BETA(list)": ("hbv", "Hbv", 730, 16384, 16, ["parBETA"]),
    "hbv_K0_BETAET(list)": ("hbv", "Hbv", 730, 16384, 16, ["parK0", "parBETAET"]),
    "hbv_FC_K1_TT(list)": ("hbv", "Hbv", 730, 16384, 16, ["parFC", "parK1", "parTT"]),
    "hbv2_BETA_K0_BETAET(compiled)": ("hbv_2", "Hbv_2", 730, 16384, 16, ["parBETA", "parK0", "parBETAET"]),
    "hbv2_BETA_BETAET(list)": ("hbv_2", "Hbv_2", 730, 16384, 16, ["parBETA", "parBETAET"]),
    "hbv2_BETA_K0_BETAET(list, forced)": ("hbv_2", "Hbv_2", 730, 16384, 16, ["parBETA", "parK0", "parBETAET"]),
    "hbv_5dyn": ("hbv", "Hbv", 730, 16384, 16, ["parBETA", "parFC", "parK0", "parLP", "parBETAET"]),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    from hydrodl2_amd import _lib
    for name, spec in CASES.items():
        bench.WORKLOADS[name] = spec
        if "forced" in name:
            os.environ["HBVX_STREAM_SLOTLIST"] = "1"
        wl = bench.Workload(name, dev, seed=7)
        dt, kms = bench.timed_steps(wl, args.steps, 3, dev, 1)
        os.environ.pop("HBVX_STREAM_SLOTLIST", None)
        lib = _lib.get_library()
        smp = bench.timed_steps.samples
        print(json.dumps({"case": name, "ms_median": smp["ms_median"], "forward_ms": round(kms.get("hbvx_forward", 0), 4),
                          "backward_ms": round(kms.get("hbvx_backward", 0), 4),
                          "dispatch": [lib.last_dispatch(0), lib.last_dispatch(1)]}), flush=True)
        del wl
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
