"""Implicit HBV ("HBV adjoint").  Parity is UNPINNED by the reference (its file is not
importable and its own tests skip it); these tests pin the product to the independent float64
autograd oracle (oracle/hbv_adj_oracle.py) and the oracle to finite differences.

CPU tier: the product's math header compiled for the host (tests/hosttest) driven through the
package's HbvAdj module.  GPU tier (-m gpu): the HIP kernels."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from . import synth
from .test_step_math_host import steptest_lib  # noqa: F401  (fixture)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("hbv_adj_oracle",
                                               os.path.join(ROOT, "oracle", "hbv_adj_oracle.py"))
adj_oracle = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(adj_oracle)


def _inputs(T, B, M, seed, betaet, cold=False, scale=1.0):
    n = 13 if betaet else 12
    x = synth.forcing(T, B, seed, cold=cold)
    p = synth.raw_parameters(T, B, n * M + 2, seed, scale)
    w = synth.loss_weights((T, B, 1), seed, 70)
    return torch.from_numpy(x), torch.from_numpy(p), torch.from_numpy(w)


def _product(device, x, p, w, cfg):
    import hydrodl2_amd
    H = hydrodl2_amd.load_model("hbv_adj", "HbvAdj")
    m = H(cfg, torch.device(device))
    pp = p.to(device).clone().requires_grad_(True)
    out = m({"x_phy": x.to(device)}, pp)["flow_sim"]
    (out * w[-out.shape[0]:].to(device)).sum().backward()
    return out.detach().cpu().numpy(), pp.grad.cpu().numpy()


def _oracle(x, p, w, cfg, **kw):
    pp = p.clone().double().requires_grad_(True)
    out, its = adj_oracle.hbv_adj_forward(
        x, pp, nmul=cfg["nmul"], warm_up=cfg.get("warm_up", 0),
        dynamic_params=cfg["dynamic_params"]["HbvAdj"], dy_drop=cfg.get("dy_drop", 0.0),
        gtol=cfg.get("newton_gtol", 1e-3), max_iter=cfg.get("newton_max_iter", 3), **kw)
    (out * w[-out.shape[0]:].double()).sum().backward()
    return out.detach().numpy(), pp.grad.numpy(), its


def _close(name, a, b, rtol, atol_rel):
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b)
    tol = atol_rel * scale + rtol * np.abs(b)
    assert (err <= tol).all(), f"{name}: max err {err.max():.3e} (scale {scale:.3e}), " \
                               f"{int((err > tol).sum())}/{err.size} outside"


CASES = [
    dict(T=40, B=5, M=4, seed=81, cfg=dict(nmul=4, dynamic_params={"HbvAdj": []})),
    dict(T=50, B=4, M=16, seed=82,
         cfg=dict(nmul=16, warm_up=10, dynamic_params={"HbvAdj": ["parBETA", "parBETAET"]})),
    dict(T=36, B=6, M=3, seed=83, cold=True, scale=2.0,
         cfg=dict(nmul=3, dy_drop=0.5, dynamic_params={"HbvAdj": ["parK0", "parFC", "parBETAET"]})),
    # a 40-day warm-up: long enough for the pipelined kernel, which then runs WITHOUT a flux output and is
    # differentiated through (hbv_adj.py:257-274), followed by 50 main days with a dynamic snow parameter
    dict(T=90, B=5, M=16, seed=84, cfg=dict(nmul=16, warm_up=40, dynamic_params={"HbvAdj": ["parCFMAX"]})),
    # five dynamic parameters with drops: above ADJ_FEW, the generic instances of the time-parallel adjoint kernels
    # (hbv_adj_kernels.h; up to three take the slot-list ones)
    dict(T=44, B=4, M=8, seed=85,
         cfg=dict(nmul=8, dy_drop=0.3,
                  dynamic_params={"HbvAdj": ["parBETA", "parK1", "parLP", "parTT", "parBETAET"]})),
]


SOLVERS = ["staged", "joint"]   # csrc/hbv_adj_step.h: AdjStaged (the default) and the reference's joint iteration


def _run_case(device, case, tight, solver="staged"):
    cfg = dict(case["cfg"], newton_solver=solver)
    if tight:
        cfg.update(newton_gtol=1e-6, newton_max_iter=12)
    betaet = "parBETAET" in cfg["dynamic_params"]["HbvAdj"]
    x, p, w = _inputs(case["T"], case["B"], case["M"], case["seed"], betaet,
                      case.get("cold", False), case.get("scale", 1.0))
    torch.manual_seed(5)
    got, ggot = _product(device, x, p, w, cfg)
    torch.manual_seed(5)
    want, gwant, its = _oracle(x, p, w, cfg)     # always the reference's joint iteration, float64
    return got, ggot, want, gwant, its


@pytest.fixture()
def host_math_backend(steptest_lib, oracle_path, monkeypatch):  # noqa: F811
    from tests import seam
    monkeypatch.setenv("HBVX_ORACLE_LIB", oracle_path)
    seam.use_library(steptest_lib)
    yield
    seam.use_library(None)


def test_oracle_gradient_is_the_implicit_function_derivative():
    """Central finite differences of the oracle's own (tightly converged) forward."""
    T, B, M = 12, 2, 2
    x, p, w = _inputs(T, B, M, 80, True)
    cfg = dict(nmul=M, newton_gtol=1e-11, newton_max_iter=40,
               dynamic_params={"HbvAdj": ["parBETA", "parBETAET"]})
    _, g, _ = _oracle(x, p, w, cfg)

    def loss(pv):
        with torch.no_grad():
            out, _ = adj_oracle.hbv_adj_forward(x, pv, nmul=M, dynamic_params=["parBETA", "parBETAET"],
                                                gtol=1e-11, max_iter=40)
        return float((out * w.double()).sum())

    rng = np.random.default_rng(0)
    pd = p.double()
    worst = 0.0
    for _ in range(12):
        idx = (int(rng.integers(T)), int(rng.integers(B)), int(rng.integers(p.shape[2])))
        if abs(g[idx]) < 1e-8:
            continue
        e = torch.zeros_like(pd)
        e[idx] = 1e-5
        fd = (loss(pd + e) - loss(pd - e)) / 2e-5
        worst = max(worst, abs(fd - g[idx]) / max(abs(g[idx]), 1e-6))
    assert worst < 2e-3, worst


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"M{c['M']}-T{c['T']}")
def test_host_math_matches_oracle_converged(case, solver, host_math_backend):
    """Tightly converged Newton: the scheme itself (values and implicit-function gradients).  The staged
    solve (closed-form snow / upper / lower zone blocks + scalar Newton on soil moisture) must land on the
    same root as the float64 joint iteration."""
    got, ggot, want, gwant, _ = _run_case("cpu", case, tight=True, solver=solver)
    _close("flow_sim", got, want, 2e-4, 2e-5)
    _close("grad", ggot, gwant, 2e-3, 2e-4)


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("case", CASES[:2], ids=lambda c: f"M{c['M']}-T{c['T']}")
def test_host_math_matches_oracle_reference_policy(case, solver, host_math_backend):
    """The reference's Newton policy (gtol 1e-3, <= 4 updates), per-lane stopping."""
    got, ggot, want, gwant, its = _run_case("cpu", case, tight=False, solver=solver)
    assert float(its.max()) <= 4
    _close("flow_sim", got, want, 5e-3, 5e-4)
    _close("grad", ggot, gwant, 5e-2, 5e-3)


def test_the_solver_is_never_inferred_from_the_presence_of_a_key():
    """{'newton_stop': 'lane'} -- the default written out -- must select what omitting it selects."""
    import hydrodl2_amd
    C = hydrodl2_amd.load_model("hbv_adj", "HbvAdj")
    base = {"nmul": 2, "dynamic_params": {"HbvAdj": ["parBETAET"]}}
    dev = torch.device("cpu")
    assert C(base, dev).newton_solver == "staged"
    assert C(dict(base, newton_stop="lane"), dev).newton_solver == "staged"
    assert C(dict(base, newton_stop="global"), dev).newton_solver == "joint"      # only the joint iteration has that rule
    assert C(dict(base, newton_solver="joint"), dev).newton_solver == "joint"
    with pytest.raises(ValueError):
        C(dict(base, newton_solver="staged", newton_stop="global"), dev)


def test_global_stopping_rule_differs_only_within_gtol():
    """What the per-lane rule changes w.r.t. the reference's batch-global maximum."""
    case = CASES[0]
    x, p, w = _inputs(case["T"], case["B"], case["M"], case["seed"], False)
    a, _, _ = _oracle(x, p, w, case["cfg"], stop="lane")
    b, _, _ = _oracle(x, p, w, case["cfg"], stop="global")
    assert np.abs(a - b).max() <= 2e-3 * max(1.0, np.abs(b).max())


@pytest.mark.gpu
@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"M{c['M']}-T{c['T']}")
def test_hip_matches_oracle_converged(case, solver, hip_backend):
    got, ggot, want, gwant, _ = _run_case("cuda:0", case, tight=True, solver=solver)
    _close("flow_sim", got, want, 2e-4, 2e-5)
    _close("grad", ggot, gwant, 2e-3, 2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"M{c['M']}-T{c['T']}")
def test_hip_pipelined_staged_forward_equals_the_one_wave_stepper(case, hip_backend, monkeypatch):
    """The staged solve runs as a three-wave pipeline (hbv_pipe.h, blocks of consecutive days on different
    waves); HBVX_FWD=tiled pins the single stepper wave that solves the three blocks in sequence.  Same
    block code, same ensemble add order: bit-identical values and gradients."""
    a, ga, _, _, _ = _run_case("cuda:0", case, tight=False)
    monkeypatch.setenv("HBVX_FWD", "tiled")
    b, gb, _, _, _ = _run_case("cuda:0", case, tight=False)
    assert np.array_equal(a, b) and np.array_equal(ga, gb)


@pytest.mark.gpu
def test_hip_reference_policy_and_size(hip_backend):
    got, ggot, want, gwant, its = _run_case("cuda:0", CASES[1], tight=False)
    _close("flow_sim", got, want, 5e-3, 5e-4)
    _close("grad", ggot, gwant, 5e-2, 5e-3)
    # BASELINE config 4 shape: 671 basins x 16 members (365 days here), finite and basin-independent
    import hydrodl2_amd
    dev = torch.device("cuda:0")
    H = hydrodl2_amd.load_model("hbv_adj", "HbvAdj")
    m = H({"nmul": 16, "dynamic_params": {"HbvAdj": ["parBETAET"]}}, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    T, B = 365, 671
    x = torch.from_numpy(synth.forcing(T, B, 90)).to(dev)
    p = torch.randn((T, B, 13 * 16 + 2), generator=g, device=dev).requires_grad_(True)
    out = m({"x_phy": x}, p)["flow_sim"]
    out.sum().backward()
    assert torch.isfinite(out).all() and torch.isfinite(p.grad).all()
    sel = torch.tensor([0, 5, 670], device=dev)
    p2 = p.detach()[:, sel].contiguous().requires_grad_(True)
    out2 = m({"x_phy": x[:, sel].contiguous()}, p2)["flow_sim"]
    assert torch.equal(out[:, sel], out2)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"M{c['M']}-T{c['T']}")
def test_hip_time_parallel_adjoint_matches_oracle(case, hip_backend, monkeypatch):
    """Chunks of 8 days so that these short cases take the time-parallel adjoint (ragged last chunk)."""
    monkeypatch.setenv("HBVX_CHUNK", "8")
    got, ggot, want, gwant, _ = _run_case("cuda:0", case, tight=True)
    _close("flow_sim", got, want, 2e-4, 2e-5)
    _close("grad", ggot, gwant, 2e-3, 2e-4)


@pytest.mark.gpu
def test_hip_one_wave_kernels_still_match(hip_backend, monkeypatch):
    """HBVX_KERNEL=simple: the first-cut one-wave forward (kept as a cross-check)."""
    monkeypatch.setenv("HBVX_KERNEL", "simple")
    got, ggot, want, gwant, _ = _run_case("cuda:0", CASES[1], tight=True)
    _close("flow_sim", got, want, 2e-4, 2e-5)
    _close("grad", ggot, gwant, 2e-3, 2e-4)


@pytest.mark.gpu
def test_hip_global_newton_rule_is_the_oracles_global_rule(hip_backend):
    """newton_stop='global' (hbvx_desc.adj_stop = 1): all 64 lanes of a wavefront take as many Newton
    updates as the slowest of them -- for a batch of exactly one wavefront (4 basins x 16 members) that
    is the reference's batch-global rule (hbv_adj.py:544,546), which the oracle implements as
    stop='global'.  The HIP result with the global rule must match the oracle's global run, and must be
    closer to it than the per-lane HIP result is (so the switch is shown to act)."""
    T, B, M = 60, 4, 16
    x, p, w = _inputs(T, B, M, 91, True)
    cfg = dict(nmul=M, dynamic_params={"HbvAdj": ["parBETA", "parBETAET"]})
    got_g, gg_g = _product("cuda:0", x, p, w, dict(cfg, newton_solver="joint", newton_stop="global"))
    got_l, gg_l = _product("cuda:0", x, p, w, dict(cfg, newton_solver="joint", newton_stop="lane"))
    want_g, gw_g, its_g = _oracle(x, p, w, cfg, stop="global")
    want_l, _, its_l = _oracle(x, p, w, cfg, stop="lane")
    assert float(its_g.max()) <= 4
    assert np.abs(want_g - want_l).max() > 0, "the two rules coincide on this case: not a test of the switch"
    # un-converged Newton (gtol 1e-3): float32 against float64 at the tolerances of the reference-policy tests
    _close("flow_sim (global)", got_g, want_g, 1e-3, 5e-4)
    # gradients of an un-converged iteration are sensitive element by element: compare in the L2 norm
    rel = np.linalg.norm(gg_g - gw_g) / np.linalg.norm(gw_g)
    assert rel < 2e-2, rel
    e_g = np.abs(got_g - want_g).max()
    e_l = np.abs(got_l - want_g).max()
    assert e_g < e_l, (e_g, e_l)


@pytest.mark.gpu
def test_hip_cfg4_full_length_and_oracle_spot_check(hip_backend, monkeypatch):
    """BASELINE config 4 at its full size -- 671 basins x 16 members x 7300 days, pipelined staged forward +
    time-parallel adjoint: finite, basin-independent, bit-identical to the one-wave stepper over the whole
    record; and a float64-oracle spot check (values and gradients) of three basins over 1460 days with the
    reference's Newton policy."""
    import hydrodl2_amd
    dev = torch.device("cuda:0")
    H = hydrodl2_amd.load_model("hbv_adj", "HbvAdj")
    cfg = {"nmul": 16, "dynamic_params": {"HbvAdj": ["parBETAET"]}}
    m = H(cfg, dev)
    T, B, M = 7300, 671, 16
    x = torch.from_numpy(synth.forcing(T, B, 92)).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    p = torch.randn((T, B, 13 * M + 2), generator=g, device=dev).requires_grad_(True)
    out = m({"x_phy": x}, p)["flow_sim"]
    out.sum().backward()
    assert out.shape == (T, B, 1) and torch.isfinite(out).all() and torch.isfinite(p.grad).all()
    sel = [0, 335, 670]
    pick = torch.tensor(sel, device=dev)
    p2 = p.detach()[:, pick].contiguous().requires_grad_(True)
    out2 = m({"x_phy": x[:, pick].contiguous()}, p2)["flow_sim"]
    assert torch.equal(out[:, pick], out2)
    # the three-wave pipeline against the single stepper wave solving the three blocks in sequence: 913 tiles of
    # 8 days, dynamic parBETAET staged by the filler waves, 168 workgroups -- same numbers to the last bit
    monkeypatch.setenv("HBVX_FWD", "tiled")
    with torch.no_grad():
        out_t = m({"x_phy": x}, p.detach())["flow_sim"]
    monkeypatch.delenv("HBVX_FWD")
    assert torch.equal(out, out_t)
    # oracle: the first 1460 days of those three basins (a day-by-day float64 autograd loop)
    Ts = 1460
    xs, ps = x[:Ts, pick].cpu(), p.detach()[:Ts, pick].cpu()
    ws = torch.from_numpy(synth.loss_weights((Ts, 3, 1), 92, 71))
    got, ggot = _product("cuda:0", xs, ps, ws, cfg)
    want, gwant, its = _oracle(xs, ps, ws, cfg)
    assert float(its.max()) <= 4
    _close("flow_sim", got, want, 5e-3, 5e-4)
    _close("grad", ggot, gwant, 5e-2, 5e-3)
