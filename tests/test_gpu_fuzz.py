"""GPU tier: a fixed-seed slice of tools/fuzz_parity.py -- 200 random (model, shape, ensemble size,
dynamic set, dy_drop, muwts, warm-up offset, routing) draws, HIP against the oracle.

Fluxes, routed series and final storages must agree in EVERY draw.  For gradients one class of
difference is expected and bounded here instead of ignored: the model is not differentiable where a
min/clamp argument ties (e.g. P = 0 on a dry day), the GPU's pow / exp differ from libm's by ulps, so
an isolated element can land on the other one-sided slope ("threshold flip").  A draw
whose gradient mismatch touches fewer than 1e-3 of the elements counts as a flip draw; anything
larger, or any non-gradient mismatch, fails outright.  The count is printed and bounded."""
import random
import re

import numpy as np
import pytest

from . import golden_cases as gc
from .abi_util import compare_runs, make_problem, run_problem

pytestmark = pytest.mark.gpu

N_DRAWS, SEED, MAX_FLIP_DRAWS = 200, 20261004, 12   # measured on MI355X: 6 draws, 1-9 elements each, all in g_x


def _draw(rng):
    model = rng.choice(["Hbv", "Hbv", "Hbv_1_1p", "Hbv_2", "Hbv_2_hourly"])
    names = list(gc.PHY_NAMES[model]) + (["parBETAET"] if model == "Hbv" else [])
    mode = rng.choice(["static", "few", "default", "many", "all"])
    if mode == "static":
        dyn = []
    elif mode == "few":
        dyn = rng.sample(names, rng.randint(1, 3))
    elif mode == "default":
        dyn = [n for n in ("parBETA", "parK0", "parBETAET") if n in names and (n != "parK0" or rng.random() < 0.5)]
    elif mode == "many":
        dyn = rng.sample(names, rng.randint(4, len(names) - 1))
    else:
        dyn = list(names)
    if model == "Hbv" and "parBETAET" not in dyn and rng.random() < 0.5:
        names.remove("parBETAET")
    dyn = [n for n in names if n in dyn]
    M = rng.choice([1, 2, 3, 4, 5, 8, 16, 16, 16, 32, 64])
    B = rng.choice([1, 2, 3, 7, 19, 37, 64, 130])
    T = rng.choice([2, 5, 31, 33, 64, 65, 100, 129, 257])
    kw = dict(model=model, T=T, B=B, M=M, dyn=tuple(dyn), seed=rng.randint(1, 10 ** 6),
              drop_frac=rng.choice([0.0, 0.0, 0.3]) if dyn else 0.0,
              muwts=(rng.random() < 0.2 and model == "Hbv"), cold=rng.random() < 0.3,
              betaet=("parBETAET" in names and model == "Hbv"),
              routing=rng.random() < 0.8, raw_scale=rng.choice([1.0, 1.0, 2.5]))
    t0 = rng.randint(1, max(1, T // 2)) if (rng.random() < 0.3 and T > 4) else 0
    grad = rng.random() < 0.75
    return kw, t0, grad


def test_fixed_seed_fuzz_bounds_threshold_flips(hip_backend, oracle_path, capsys):
    rng = random.Random(SEED)
    flips, detail = 0, []
    for case in range(N_DRAWS):
        kw, t0, grad = _draw(rng)
        prob = make_problem(**kw)
        got = run_problem(prob, None, device="cuda:0", x_grad=grad, backward=grad, t0=t0)
        want = run_problem(prob, oracle_path, device="cpu", x_grad=grad, backward=grad, t0=t0)
        compare_runs(prob, got, want, label=f"draw {case} {kw}", keys=("flux", "routed", "state_out"))
        flipped = False
        for k in ("g_params", "g_x", "g_muwts"):
            if not (grad and k in want and want[k] is not None):
                continue
            try:
                compare_runs(prob, got, want, label=f"draw {case} [gradient; threshold flips are counted, not refused]", keys=(k,))
            except AssertionError as e:
                m = re.search(r"(\d+)/(\d+) outside tol", str(e))
                nbad, size = int(m.group(1)), int(m.group(2))
                assert nbad <= max(1e-3 * size, 3), f"draw {case} {kw} t0={t0}: {e}"
                flipped = True
                detail.append((case, kw["model"], k, nbad, size))
        flips += flipped
    msg = f"fuzz: {N_DRAWS} draws, {flips} with isolated gradient threshold flips (bound {MAX_FLIP_DRAWS}): {detail}"
    with capsys.disabled():          # in the output the driver tails, pass or fail
        print("\n" + msg)
    try:                             # and on file for the round's records
        import os
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        if os.path.isdir(out):
            open(os.path.join(out, "fuzz_flips.txt"), "w").write(msg + "\n")
    except OSError:
        pass
    assert flips <= MAX_FLIP_DRAWS, detail
