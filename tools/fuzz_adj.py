#!/usr/bin/env python3
"""Randomised soak of the implicit scheme's staged forward (not part of the test tiers):

    python tools/fuzz_adj.py [n_cases] [seed]

For random shapes (T 32-400, B 1-70, nmul 1-40), dynamic sets (0-3 parameters from all blocks), dy_drop masks and
warm-ups, the three-wave pipeline (k_fwd_pipe<HBVADJ>) must give the SAME BITS as the one-wave stepper that solves
the three blocks in sequence (HBVX_FWD=tiled) -- values and gradients -- and both must satisfy the scheme's own
acceptance test: the float64 residual of the returned trajectory (checked through the streamflow only here; the
block residuals are checked on the host build in tests/test_step_math_host.py)."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hydrodl2_amd  # noqa: E402
from tests import synth  # noqa: E402

NAMES = ["parBETA", "parFC", "parK0", "parK1", "parK2", "parLP", "parPERC", "parUZL", "parTT", "parCFMAX", "parCFR",
         "parCWH", "parBETAET"]


def run(cfg, x, p, w, env):
    for k in ("HBVX_FWD",):
        os.environ.pop(k, None)
    os.environ.update(env)
    H = hydrodl2_amd.load_model("hbv_adj", "HbvAdj")
    m = H(cfg, torch.device("cuda:0"))
    pp = p.clone().requires_grad_(True)
    torch.manual_seed(11)                      # the per-lane dy_drop masks come from the CPU generator
    out = m({"x_phy": x}, pp)["flow_sim"]
    (out * w[-out.shape[0]:]).sum().backward()
    torch.cuda.synchronize()
    return out.detach(), pp.grad


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    t0 = time.time()
    for case in range(n_cases):
        T = rng.choice([32, 33, 40, 64, 65, 100, 129, 400])
        B = rng.choice([1, 2, 3, 4, 5, 7, 16, 37, 70])
        M = rng.choice([1, 2, 3, 4, 5, 8, 16, 16, 32, 40])
        nd = rng.randint(0, 3)
        dyn = rng.sample(NAMES, nd)
        be = "parBETAET" in dyn
        n = 13 if be else 12
        wu = rng.choice([0, 0, 0, 5, 33]) if T > 70 else 0
        cfg = {"nmul": M, "warm_up": wu, "dy_drop": rng.choice([0.0, 0.0, 0.4]), "dynamic_params": {"HbvAdj": dyn}}
        seed = rng.randint(1, 10 ** 6)
        x = torch.from_numpy(synth.forcing(T, B, seed, cold=rng.random() < 0.3)).cuda()
        p = torch.from_numpy(synth.raw_parameters(T, B, n * M + 2, seed, rng.choice([1.0, 1.0, 2.5]))).cuda()
        w = torch.from_numpy(synth.loss_weights((T, B, 1), seed, 70)).cuda()
        a, ga = run(cfg, x, p, w, {})
        b, gb = run(cfg, x, p, w, {"HBVX_FWD": "tiled"})
        ok = torch.equal(a, b) and torch.equal(ga, gb) and bool(torch.isfinite(a).all()) and bool(torch.isfinite(ga).all())
        bad += not ok
        print(f"[{case:3d}] {'ok      ' if ok else 'MISMATCH'} T={T} B={B} M={M} dyn={dyn} warm_up={wu} drop={cfg['dy_drop']} "
              f"seed={seed}" + ("" if ok else f" max|dq|={float((a - b).abs().max()):.3g} max|dg|={float((ga - gb).abs().max()):.3g}"),
              flush=True)
    os.environ.pop("HBVX_FWD", None)
    print(f"{n_cases - bad}/{n_cases} cases bit-identical, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
