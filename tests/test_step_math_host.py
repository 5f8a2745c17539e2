"""CPU tier: the product's device math (hydrodl2_amd/csrc/hbv_step.h) compiled for the host
and driven through the ABI, against the oracle, on identical descriptors.  Catches slips in the
hand-written adjoint before any GPU time is spent."""
import os
import subprocess

import pytest

from .abi_util import assert_close, make_problem, run_problem

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hosttest", "hbvx_host.cpp")
LIB = os.path.join(HERE, "hosttest", "libhbvx_steptest.so")
STEP_H = os.path.join(os.path.dirname(HERE), "hydrodl2_amd", "csrc", "hbv_step.h")


@pytest.fixture(scope="module")
def steptest_lib():
    newest = max(os.path.getmtime(SRC), os.path.getmtime(STEP_H))
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17",
                               "-ffp-contract=off", "-o", LIB, SRC])
    return LIB


CASES = [
    dict(model="Hbv", M=4, dyn=()),
    dict(model="Hbv", M=16, dyn=("parBETA", "parBETAET"), drop_frac=0.4),
    dict(model="Hbv", M=3, dyn=("parK0", "parTT"), muwts=True, cold=True, raw_scale=2.0),
    dict(model="Hbv_1_1p", M=2, dyn=("parBETA", "parFC", "parC", "parBETAET"), cold=True),
    dict(model="Hbv_2", M=4, dyn=("parBETA", "parK0", "parRT", "parAC")),
]


@pytest.mark.parametrize("kw", CASES, ids=lambda k: f"{k['model']}-M{k['M']}-{len(k['dyn'])}dyn")
def test_step_math_matches_oracle(kw, steptest_lib, oracle_path):
    prob = make_problem(T=50, B=6, seed=5, routing=False, **kw)
    a = run_problem(prob, steptest_lib, x_grad=True)
    b = run_problem(prob, oracle_path, x_grad=True)
    assert_close("flux", a["flux"], b["flux"], 1e-5, 1e-6)
    assert_close("state_out", a["state_out"], b["state_out"], 1e-5, 1e-6)
    assert_close("traj", a["traj"], b["traj"], 1e-5, 1e-6)
    for k in ("g_params", "g_x") + (("g_muwts",) if "g_muwts" in b else ()):
        assert_close(k, a[k], b[k], 2e-4, 2e-6)
