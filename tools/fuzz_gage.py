#!/usr/bin/env python3
"""Randomised soak of the gage routing (hbvx_gage_route_*) on the GPU against the CPU oracle, with the
tolerances of tests/test_gage_route.py:  python tools/fuzz_gage.py [n_cases] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from tests.abi_util import assert_close, assert_grad_close  # noqa: E402
from tests.test_gage_route import _problem, _run  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for case in range(n):
        T = rng.choice([1, 2, 5, 37, 71, 72, 73, 100, 255, 256, 257, 1023, 1024, 1025, 1500, 2200])
        U = rng.choice([1, 2, 5, 17, 64, 150, 400])
        G = rng.choice([1, 2, 3, 7, 20])
        G = min(G, U)
        lag = rng.random() < 0.7
        dense = rng.choice([0.05, 0.4, 1.0])
        pb = _problem(T, U, G, seed=rng.randint(1, 10 ** 6), dense=dense)
        try:
            want = _run(pb, lag, ge.ORACLE_LIB, "cpu")
            got = _run(pb, lag, None, "cuda")
            assert_close("out", got[0], want[0])
            assert_grad_close("grad_qs", got[1], want[1])
            assert_grad_close("grad_dp", got[2], want[2], list(range(want[2].shape[-1])))
            status = "ok"
        except AssertionError as e:
            bad += 1
            status = "MISMATCH " + str(e)[:250]
        except Exception as e:  # noqa: BLE001
            bad += 1
            status = "ERROR " + repr(e)[:250]
        print(f"[{case:3d}] {status} T={T} U={U} G={G} lag={lag} dense={dense} pairs={int(pb['topo'].sum())}", flush=True)
    print(f"{n - bad}/{n} cases agree", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
