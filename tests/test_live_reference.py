"""CPU tier, authoring container only: the CPU oracle against the LIVE reference on freshly drawn cases.

The committed fixtures pin the oracle on 39 hand-picked configurations; this test draws more of them every run from a
fixed seed list -- model class, ensemble size, dynamic-parameter set, warm-up, dy_drop, routing, cold start, parameter
spread -- runs the reference itself (imported from /root/reference exactly as tests/golden/make_golden.py does) and this
package through the oracle, and compares every output, the final states and every gradient at the stated tolerances
(tests/helpers.py::compare).  Skipped where the reference is not present (the GPU box).

One allowance the committed fixtures do not need: up to 0.5 % of a gradient tensor's elements (or four) may miss the per-GROUP
tolerance as long as they meet BASELINE.md's original proposal, 2e-6 of the TENSOR's largest entry.  Measured: draw 2 (Hbv,
4 members, static, cold start) -- the gradient of parFC for one lane is the sum of 40 daily terms of up to 1.9e-3 that
cancel to 1.3e-6, so float32 evaluations scatter around the float64 value 1.9462e-4: reference 1.9448e-4, a second torch
float32 evaluation of the same equations 1.8665e-4, the oracle 2.0047e-4; draw 138 (all 13 parameters dynamic, cold start)
-- parK2's per-day gradients of the first days, 5e-5 in a tensor whose largest entry is 22.6: reference -0.44 %, oracle
+0.34 %, second float32 evaluation +0.66 % from the float64 value.  (Seeds 0-299 under these rules: 297 agree; the other three are
single small elements of the same kind.  The 40 seeds below are the committed, deterministic set.)
"""
import os
import random
import sys
import warnings

import pytest

from . import golden_cases as gc
from .helpers import compare, run_case

REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


class _Rec(dict):
    files = property(lambda self: list(self.keys()))


def _draw(seed: int) -> dict:
    rng = random.Random(seed)
    model = rng.choice(["Hbv", "Hbv", "Hbv_1_1p", "Hbv_2", "Hbv_2_hourly"])
    names = list(gc.PHY_NAMES[model]) + (["parBETAET"] if model == "Hbv" else [])
    mode = rng.choice(["static", "few", "few", "all"])
    dyn = [] if mode == "static" else (list(names) if mode == "all" else rng.sample(names, rng.randint(1, 3)))
    if model == "Hbv" and "parBETAET" not in dyn and rng.random() < 0.5:
        names.remove("parBETAET")
    dyn = tuple(n for n in names if n in dyn)
    kw = {}
    if model in ("Hbv", "Hbv_1_1p") and rng.random() < 0.4:
        kw["warm_up"] = rng.choice([5, 12])
        kw["warm_up_states"] = rng.random() < 0.7
    if dyn and rng.random() < 0.4:
        kw["dy_drop"] = 0.3
    if model in ("Hbv_2", "Hbv_2_hourly"):
        kw["routing"] = rng.random() < 0.5       # (routing=False on Hbv / Hbv_1_1p crashes upstream: hbv.py:550-567)
    spec = dict(model=model, config=gc._cfg(model, rng.choice([2, 3, 4, 16]), dyn, **kw),
                T=rng.choice([24, 40, 64]), B=rng.choice([3, 5, 7]), seed=1000 + seed, loss="all",
                torch_seed=seed, cold=rng.random() < 0.3, raw_scale=rng.choice([1.0, 1.0, 2.5]))
    if model == "Hbv_2_hourly":
        spec["G"] = rng.choice([2, 3])
        # a day or two of hours from a cold start moves no water: every gradient is a 1e-12 rounding residue.  Not a case
        spec["cold"] = False
        spec["T"] = max(spec["T"], 64)
    return spec


@pytest.mark.parametrize("seed", list(range(40)))
def test_oracle_equals_the_live_reference_on_a_fresh_case(seed, oracle_backend):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    name = f"_live_{seed}"
    gc.CASES[name] = _draw(seed)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            hydrodl2 = make_golden._import_reference()
            ref = _Rec(make_golden._run_case(hydrodl2, name))
            res = run_case(name, "cpu")
        compare(name, res, ref, grad_outlier_frac=5e-3, grad_outlier_atol_rel=2e-6)
    finally:
        del gc.CASES[name]
