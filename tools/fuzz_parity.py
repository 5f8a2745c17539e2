#!/usr/bin/env python3
"""Randomised parity sweep: the HIP path against the CPU oracle (tests/abi_util.py helpers and the
tolerances of tests/test_gpu_parity.py) over random models, shapes, member counts, dynamic-parameter
sets, dy_drop masks and ensemble weights.  Not part of the test tiers: a longer soak for the GPU box.

    python tools/fuzz_parity.py [n_cases] [seed]
"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from tests import golden_cases as gc  # noqa: E402
from tests.abi_util import compare_runs, make_problem, run_problem  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    if not os.path.exists(ge.ORACLE_LIB):
        ge.build_oracle()
    bad = 0
    t_start = time.time()
    for case in range(n_cases):
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
        dyn = [n for n in names if n in dyn]                     # table order
        M = rng.choice([1, 2, 3, 4, 5, 8, 16, 16, 16, 32, 64])
        B = rng.choice([1, 2, 3, 7, 19, 37, 64, 130, 200])
        T = rng.choice([2, 5, 31, 33, 64, 65, 100, 129, 257, 400, 730])
        kw = dict(model=model, T=T, B=B, M=M, dyn=tuple(dyn), seed=rng.randint(1, 10 ** 6),
                  drop_frac=rng.choice([0.0, 0.0, 0.3]) if dyn else 0.0,
                  muwts=(rng.random() < 0.2 and model == "Hbv"), cold=rng.random() < 0.3,
                  betaet=("parBETAET" in names and model == "Hbv"),
                  routing=rng.random() < 0.8, raw_scale=rng.choice([1.0, 1.0, 2.5]),
                  channels=rng.choice([(0, 1, 2)] * 4 + [(2, 0, 1), (1, 2, 0), (0, 2, 1)]))
        t0 = rng.randint(1, max(1, T // 2)) if (rng.random() < 0.3 and T > 4) else 0   # warm-up offset
        try:
            prob = make_problem(**kw)
            grad = rng.random() < 0.75            # the rest: inference (no trajectory kept, other kernel variants)
            got = run_problem(prob, None, device="cuda:0", x_grad=grad, backward=grad, t0=t0)
            want = run_problem(prob, ge.ORACLE_LIB, device="cpu", x_grad=grad, backward=grad, t0=t0)
            compare_runs(prob, got, want, keys=("flux", "routed", "state_out", "g_params", "g_x", "g_muwts"))
            from hydrodl2_amd import _lib
            lib = _lib.get_library()
            status = "ok " + lib.last_dispatch(0) + "/" + (lib.last_dispatch(1) if grad else "-")
        except AssertionError as e:
            bad += 1
            status = "MISMATCH " + str(e)[:300]
        except Exception as e:  # noqa: BLE001
            bad += 1
            status = "ERROR " + repr(e)[:300]
        print(f"[{case:3d}] {status:28.300s} {model} T={T} B={B} M={M} dyn={len(dyn)}:{mode} drop={kw['drop_frac']} "
              f"muwts={kw['muwts']} cold={kw['cold']} routing={kw['routing']} scale={kw['raw_scale']} t0={t0} grad={grad} seed={kw['seed']}", flush=True)
    print(f"{n_cases - bad}/{n_cases} cases agree, {time.time() - t_start:.0f} s", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
