"""CPU tier: the product's device math (hydrodl2_amd/csrc/hbv_step.h) compiled for the host
and driven through the ABI, against the oracle, on identical descriptors.  Catches slips in the
hand-written adjoint before any GPU time is spent."""
import os
import subprocess

import pytest

from .abi_util import assert_close, assert_grad_close, column_groups, make_problem, run_problem

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hosttest", "hbvx_host.cpp")
LIB = os.path.join(HERE, "hosttest", "libhbvx_steptest.so")
STEP_H = os.path.join(os.path.dirname(HERE), "hydrodl2_amd", "csrc", "hbv_step.h")
ADJ_H = os.path.join(os.path.dirname(HERE), "hydrodl2_amd", "csrc", "hbv_adj_step.h")


@pytest.fixture(scope="module")
def steptest_lib():
    deps = [SRC, STEP_H, ADJ_H, os.path.join(os.path.dirname(HERE), "include", "hbvx.h"),
            os.path.join(os.path.dirname(HERE), "hydrodl2_amd", "csrc", "hbv_step_hourly.h")]
    newest = max(os.path.getmtime(f) for f in deps)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17",
                               "-ffp-contract=off", "-o", LIB, SRC, "-ldl"])
    return LIB


CASES = [
    dict(model="Hbv", M=4, dyn=()),
    dict(model="Hbv", M=16, dyn=("parBETA", "parBETAET"), drop_frac=0.4),
    dict(model="Hbv", M=3, dyn=("parK0", "parTT"), muwts=True, cold=True, raw_scale=2.0),
    dict(model="Hbv_1_1p", M=2, dyn=("parBETA", "parFC", "parC", "parBETAET"), cold=True),
    dict(model="Hbv_2", M=4, dyn=("parBETA", "parK0", "parRT", "parAC")),
    dict(model="Hbv_2_hourly", M=4, dyn=("parBETA", "parF0", "parALPHA", "parBETAET"), drop_frac=0.3),
    dict(model="Hbv_2_hourly", M=2, dyn=(), cold=True, raw_scale=2.0),
]


@pytest.mark.parametrize("kw", CASES, ids=lambda k: f"{k['model']}-M{k['M']}-{len(k['dyn'])}dyn")
def test_step_math_matches_oracle(kw, steptest_lib, oracle_path):
    prob = make_problem(T=50, B=6, seed=5, routing=False, **kw)
    a = run_problem(prob, steptest_lib, x_grad=True)
    b = run_problem(prob, oracle_path, x_grad=True)
    assert_close("flux", a["flux"], b["flux"], 1e-5, 1e-6)          # (same libm on both sides: 10 x tighter than stated)
    assert_close("state_out", a["state_out"], b["state_out"], 1e-5, 1e-6)
    assert_close("traj", a["traj"], b["traj"], 1e-5, 1e-6)
    for k in ("g_params", "g_x") + (("g_muwts",) if "g_muwts" in b else ()):
        w = b[k].shape[-1]
        groups = column_groups(w, prob["M"]) if k == "g_params" else (list(range(w)) if k == "g_x" else None)
        assert_grad_close(k, a[k], b[k], groups, rtol=2e-4, atol_rel=1e-6)


def _pow_inputs():
    import numpy as np
    from . import synth
    n = 400000
    u = synth.uniform((n,), 77, 1).astype(np.float64)
    v = synth.uniform((n,), 77, 2).astype(np.float64)
    # bases spanning SM/FC in [1e-8, 1] and the evap ratio up to ~5; exponents in [0.3, 6]
    x = np.where(u < 0.7, 10.0 ** (-8.0 * synth.uniform((n,), 77, 3).astype(np.float64)),
                 5.0 * synth.uniform((n,), 77, 4).astype(np.float64) + 1e-3)
    y = 0.3 + 5.7 * v
    x[:8] = [1.0, 0.5, 2.0, 1e-38, 3.0e38, 1.0000001, 0.9999999, 1e-45]
    y[:8] = [3.3, 2.0, 0.5, 0.3, 0.3, 6.0, 6.0, 1.0]
    return x.astype(np.float32), y.astype(np.float32)


def pow_error_ulps(got, x, y):
    import numpy as np
    exact = np.power(x.astype(np.float64), y.astype(np.float64))
    ref32 = exact.astype(np.float32)
    ok = np.isfinite(ref32) & (ref32 > 1e-37)      # normal range
    ulp = np.spacing(np.abs(ref32[ok]))
    return np.abs(got[ok].astype(np.float64) - exact[ok]) / ulp


POW_HW_MAX_ULP = 4.0   # stated bound of hbvx::pow_hw_ (hardware log2 / exp2 on the reduced mantissa)


def test_pow_accuracy(steptest_lib):
    """hbvx::pow_pos_ (hardware-transcendental arrangement; libm log2f / exp2f stand in on the
    host) stays within POW_HW_MAX_ULP of the exact x**y; the fp64-polynomial variant within 0.75."""
    import ctypes as C
    import numpy as np
    lib = C.CDLL(steptest_lib)
    x, y = _pow_inputs()
    for fn, bound in ((lib.hbvx_test_pow, POW_HW_MAX_ULP), (lib.hbvx_test_pow_f64, 0.75)):
        out = np.empty_like(x)
        fn(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
           out.ctypes.data_as(C.c_void_p), C.c_int(x.size))
        err = pow_error_ulps(out, x, y)
        assert err.max() <= bound, f"max error {err.max():.3f} ulp"
        # exact ends: 1**y == 1 (the clamp ties at SM == FC rely on it), subnormal base
        assert out[0] == 1.0
        want = np.power(np.float64(np.float32(1e-38)), np.float64(np.float32(0.3)))
        assert abs(float(out[3]) - want) <= bound * np.spacing(np.float32(want))


def test_sparse_transposed_jacobian_matches_bwd(steptest_lib):
    """Step::jt_coef / jt_unit (block-triangular J^T of HBV 1.0, used by the time-parallel adjoint's
    chunk maps) reproduces Step::bwd with zero flux adjoints on random days, including tie cases
    (empty snow pack, SM at FC, SUZ below PERC)."""
    import ctypes as C
    import numpy as np
    from . import synth
    from .abi_util import BOUNDS
    from .golden_cases import PHY_NAMES
    lib = C.CDLL(steptest_lib)
    lib.hbvx_test_jt.restype = C.c_float
    n = 20000
    u = lambda k: synth.uniform((n,), 55, k).astype(np.float64)
    st = np.stack([np.where(u(1) < 0.3, 0.0, 80 * u(2)), np.where(u(3) < 0.3, 0.0, 10 * u(4)),
                   1e-5 + 900 * u(5) ** 2, 60 * u(6), 200 * u(7)], 1).astype(np.float32)
    f = np.stack([np.where(u(8) < 0.6, 0.0, 40 * u(9)), 30 * u(10) - 12, 6 * u(11)], 1).astype(np.float32)
    names = PHY_NAMES["Hbv"] + ["parBETAET"]
    p = np.zeros((n, 19), np.float32)   # NPARAM_MAX slots
    for i, nm in enumerate(names):
        lo, hi = BOUNDS[nm]
        p[:, i] = (lo + (hi - lo) * u(20 + i)).astype(np.float32)
    st[: n // 10, 2] = p[: n // 10, 1]          # SM == FC exactly
    for be in (0, 1):
        worst = lib.hbvx_test_jt(st.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p),
                                 p.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_float(1e-5), C.c_int(be))
        assert worst < 2e-6, f"betaet={be}: worst relative difference {worst:.3g}"
    # capillary models: the snow rows alone (HBV 2.0 instantiation, both sides of the 2000 m switch)
    lib.hbvx_test_jt_snow.restype = C.c_float
    p2 = np.zeros((n, 19), np.float32)
    for i, nm in enumerate(PHY_NAMES["Hbv_2"]):
        lo, hi = BOUNDS[nm]
        p2[:, i] = (lo + (hi - lo) * u(60 + i)).astype(np.float32)
    worst = lib.hbvx_test_jt_snow(st.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p),
                                  p2.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_float(1e-5))
    assert worst < 2e-6, f"snow rows: worst difference {worst:.3g}"
    # ... and the three rows the capillary exchange couples (Step::jt_coef_cap / jt_cap), HBV 1.1p and 2.0,
    # with lower zones on both sides of the capillary demand (SLZ = 0 .. 200) and SM at FC
    lib.hbvx_test_jt_cap.restype = C.c_float
    for model, fam in ((1, "Hbv_1_1p"), (2, "Hbv_2")):
        pc = np.zeros((n, 19), np.float32)
        for i, nm in enumerate(PHY_NAMES[fam]):
            lo, hi = BOUNDS[nm]
            pc[:, i] = (lo + (hi - lo) * u(80 + 20 * model + i)).astype(np.float32)
        stc = st.copy()
        stc[: n // 10, 2] = pc[: n // 10, 1]
        stc[n // 10: n // 5, 4] = 1e-5          # lower zone at its floor: capillary rise limited by SLZ
        worst = lib.hbvx_test_jt_cap(stc.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p),
                                     pc.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_float(1e-5), C.c_int(model))
        assert worst < 4e-6, f"{fam}: worst relative difference {worst:.3g}"


def test_staged_implicit_solve_zeroes_the_residual(steptest_lib):
    """AdjStaged (csrc/hbv_adj_step.h): the closed forms of the snow, upper-zone and lower-zone blocks solve their
    equations of hbv_adj.py:425-429 exactly (residual at rounding level on 200 000 random days, including empty
    stores, T == TT, stores at the kinks), and the scalar Newton leaves |G2| <= gtol within max_iter + 1 updates."""
    import ctypes as C
    import numpy as np
    from . import synth
    from .abi_util import BOUNDS
    from .golden_cases import PHY_NAMES
    lib = C.CDLL(steptest_lib)
    n = 200000
    u = lambda k: synth.uniform((n,), 56, k).astype(np.float64)
    st = np.stack([np.where(u(1) < 0.3, 0.0, 300 * u(2) ** 2), np.where(u(3) < 0.3, 0.0, 30 * u(4) ** 2),
                   np.where(u(12) < 0.05, 0.0, 900 * u(5) ** 2), np.where(u(13) < 0.1, 0.0, 120 * u(6) ** 2),
                   200 * u(7)], 1).astype(np.float32)
    f = np.stack([np.where(u(8) < 0.6, 0.0, 60 * u(9)), 35 * u(10) - 14, 6 * u(11)], 1).astype(np.float32)
    names = PHY_NAMES["Hbv"] + ["parBETAET"]
    p = np.zeros((n, 19), np.float32)
    for i, nm in enumerate(names):
        lo, hi = BOUNDS[nm]
        v = lo + (hi - lo) * u(20 + i)
        v = np.where(u(40 + i) < 0.03, lo, np.where(u(40 + i) > 0.97, hi, v))   # saturated parameters
        p[:, i] = v.astype(np.float32)
    k = n // 20
    f[:k, 1] = p[:k, names.index("parTT")]                      # T == TT: neither melt nor refreezing
    st[k:2 * k, 2] = p[k:2 * k, names.index("parFC")]           # SM == FC
    st[2 * k:3 * k, 3] = p[2 * k:3 * k, names.index("parPERC")]  # SUZ == PERC
    st[3 * k:4 * k, 3] = p[3 * k:4 * k, names.index("parUZL")]   # SUZ == UZL
    args = (st.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), C.c_int(n))
    # the reference's policy: gtol 1e-3, at most 4 updates (hbv_adj.py:518-519)
    out = np.zeros(6, np.float32)
    g2 = np.zeros((n, 2), np.float32)
    lib.hbvx_test_staged_residual(*args, C.c_float(1e-3), C.c_int(3), out.ctypes.data_as(C.c_void_p),
                                  g2.ctypes.data_as(C.c_void_p))
    assert out[5] <= 4, f"soil-moisture updates (or Q mismatch): {out[5]}"
    # (G3 sees Peff from the soil block's reciprocal-based quotients against the checker's divisions: a few 1e-7
    # of a flux of hundreds of mm)
    for kk, bound in ((0, 2e-6), (1, 2e-6), (3, 6e-6), (4, 2e-6)):
        assert out[kk] < bound, f"block residual G{kk}: {out[kk]:.3g} (relative)"
    # even a melt pulse of hundreds of mm into an empty 50 mm store is solved to 1e-3 within the four updates: the
    # bracket (root below / above FC known from G2(FC), which needs no power) plus Halley's correction.  (Plain
    # Newton left ~100 of these 200 000 days unsolved, cycling across the kink at FC with |G2| up to 76.)
    late = np.abs(g2[:, 0]) > 1e-3
    assert late.sum() == 0, (int(late.sum()), float(np.abs(g2[:, 0]).max()))
    # given the updates, every day converges
    lib.hbvx_test_staged_residual(*args, C.c_float(1e-3), C.c_int(12), out.ctypes.data_as(C.c_void_p),
                                  g2.ctypes.data_as(C.c_void_p))
    assert out[5] <= 4 and np.abs(g2[:, 0]).max() <= 1e-3, (out[5], float(np.abs(g2[:, 0]).max()))
