"""CPU tier: how far the reference's OWN float32 trajectory moves when the last bits of its pow change.

Why this test exists (round 5): with the parity tier at the stated tolerances (fluxes and storages rtol 1e-4 + an
absolute 1e-5; tests/abi_util.py) the GPU against the oracle left a handful of isolated elements of the state
trajectory outside -- 3 of 461 840 at 250 days, 6 of 1 752 240 at 7 300 days, worst 3.4 x the tolerance -- all in
storages, where a last-bit difference of one day's recharge stays for weeks.  Is that the GPU's pow being poor, or the
model being ill-conditioned at that level?  The oracle can answer without a GPU: `hbvo_set_pow_noise(n)` perturbs the
step's powers by +-n ulp (sign from a hash of the operands), everything else identical.  Measured here: +-1 ulp moves
the trajectory by up to half the stated tolerance, +-2 ulp (the GPU pow's error class; torch's vectorised CPU pow is a
1-ulp pow) by up to 1.4 x, with isolated elements outside.  Hence the bounded outlier allowance of
tests/abi_util.py::assert_close (OUTLIER_FRAC, OUTLIER_FACTOR): it is sized by this measurement, and this test fails
if the measurement stops supporting it (no element moves: the allowance is unfounded; many move: the tolerance itself
is wrong)."""
import ctypes

import numpy as np
import pytest

from . import golden_cases as gc
from .abi_util import FLUX_ATOL, FLUX_RTOL, OUTLIER_FACTOR, OUTLIER_FRAC, make_problem, run_problem

CASE = dict(model="Hbv_1_1p", T=250, B=23, M=16, dyn=tuple(gc.PHY_NAMES["Hbv_1_1p"]))     # = ORACLE_CASES[5]


@pytest.fixture()
def noisy(oracle_path):
    dll = ctypes.CDLL(oracle_path)
    dll.hbvo_set_pow_noise.argtypes = [ctypes.c_int]
    yield dll.hbvo_set_pow_noise
    dll.hbvo_set_pow_noise(0)


def _ratios(got, want, key):
    a, b = got[key].astype(np.float64), want[key].astype(np.float64)
    return np.abs(a - b) / (FLUX_ATOL + FLUX_RTOL * np.abs(b))


def test_a_two_ulp_pow_moves_the_reference_trajectory_past_the_stated_tolerance(noisy, oracle_path):
    prob = make_problem(seed=7, **CASE)
    want = run_problem(prob, oracle_path, device="cpu")        # (with the adjoint: the trajectory is only kept for it)
    worst = {}
    for ulps in (1, 2):
        noisy(ulps)
        got = run_problem(prob, oracle_path, device="cpu")
        noisy(0)
        worst[ulps] = {k: (float(_ratios(got, want, k).max()), int((_ratios(got, want, k) > 1.0).sum()), got[k].size)
                       for k in ("flux", "traj")}
    print("pow noise -> worst |err| / tol, elements outside, size:", worst)
    # the ensemble-mean fluxes stay well inside under either perturbation ...
    assert worst[1]["flux"][0] < 0.5 and worst[2]["flux"][0] < 0.5
    # ... the storages do not: a 1-ulp pow uses a third to all of the tolerance, a 2-ulp pow exceeds it somewhere
    assert 0.2 < worst[1]["traj"][0] < 1.5
    assert worst[2]["traj"][0] > 0.8, "the trajectory no longer moves under a 2-ulp pow: revisit OUTLIER_*"
    # and what moves out is isolated and near: inside the allowance assert_close grants
    n_out, size = worst[2]["traj"][1], worst[2]["traj"][2]
    assert n_out <= int(OUTLIER_FRAC * size) and worst[2]["traj"][0] < OUTLIER_FACTOR


def test_noise_switch_is_off_by_default_and_deterministic(noisy, oracle_path):
    prob = make_problem(seed=3, model="Hbv", T=60, B=4, M=4, dyn=("parBETA",), betaet=True)
    a = run_problem(prob, oracle_path, device="cpu")
    noisy(2)
    b1 = run_problem(prob, oracle_path, device="cpu")
    b2 = run_problem(prob, oracle_path, device="cpu")
    noisy(0)
    c = run_problem(prob, oracle_path, device="cpu")
    assert np.array_equal(a["flux"], c["flux"]) and np.array_equal(a["g_params"], c["g_params"])
    assert np.array_equal(b1["flux"], b2["flux"]) and not np.array_equal(a["flux"], b1["flux"])
