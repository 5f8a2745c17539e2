"""GPU tier: BASELINE.json's full sizes, checked through size-independent properties
(the oracle cannot run 7.8e7 lane-days in seconds, so it checks a basin subset):

 * basins are independent: a subset of basins run alone gives bit-identical outputs/gradients;
 * oracle spot-check: a few basins at the full 7300 days against the CPU oracle;
 * time continuation: two half-length calls with carried states == one call (un-routed fluxes);
 * gradient structure: static parameters only receive gradient in the last row.
"""
import numpy as np
import pytest
import torch

from .abi_util import assert_close, assert_grad_close, column_groups, compare_runs, make_problem, run_problem

pytestmark = pytest.mark.gpu


def _gen(T, B, ny, seed, dev):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    day = torch.arange(T, device=dev, dtype=torch.float32)[:, None]
    season = torch.sin(2 * torch.pi * day / 365.0)
    P = torch.clamp((torch.rand((T, B), generator=g, device=dev) - 0.7) * 60.0, min=0.0)
    Tm = 10.0 * season + 5.0 * torch.randn((T, B), generator=g, device=dev) \
        + (torch.rand((1, B), generator=g, device=dev) * 25.0 - 10.0)
    PET = torch.clamp(3.0 + 2.5 * season, min=0.0).expand(T, B)
    x = torch.stack([P, Tm, PET], dim=-1).contiguous()
    p = torch.randn((T, B, ny), generator=g, device=dev)
    w = torch.randn((T, B, 1), generator=g, device=dev)
    return x, p, w


def _fwd_bwd(model, x, p, w, keys=("streamflow",)):
    p = p.detach().clone().requires_grad_(True)
    out = model({"x_phy": x}, p)
    loss = sum((out[k] * w[-out[k].shape[0]:]).sum() for k in keys)
    loss.backward()
    return out, p.grad


def test_cfg2_full_size_properties(hip_backend, oracle_path):
    """hbv, 671 basins x 16 members x 7300 days, static parameters, fwd+bwd."""
    import hydrodl2_amd
    dev = torch.device("cuda:0")
    T, B, M = 7300, 671, 16
    Hbv = hydrodl2_amd.load_model("hbv", "Hbv")
    model = Hbv({"nmul": M, "dynamic_params": {"Hbv": []}}, dev)
    ny = model.learnable_param_count
    x, p, w = _gen(T, B, ny, 5, dev)
    out, grad = _fwd_bwd(model, x, p, w, keys=("streamflow", "AET_hydro", "SWE"))
    assert all(torch.isfinite(v).all() for v in out.values())
    assert torch.isfinite(grad).all()
    # static parameters: gradient only in the last row
    assert float(grad[:-1].abs().max()) == 0.0 and float(grad[-1].abs().max()) > 0.0

    # (1) basin independence, including a block that straddles wave boundaries and the tail
    sel = torch.tensor([0, 1, 2, 3, 4, 333, 334, 335, 668, 669, 670], device=dev)
    out_s, grad_s = _fwd_bwd(model, x[:, sel].contiguous(), p[:, sel].contiguous(),
                             w[:, sel].contiguous(), keys=("streamflow", "AET_hydro", "SWE"))
    for k in out:
        if k == "BFI":  # torch.sum glue: its reduction order depends on the tensor shape
            torch.testing.assert_close(out[k][sel], out_s[k], rtol=1e-5, atol=1e-5)
            continue
        assert torch.equal(out[k][:, sel], out_s[k]), k
    assert torch.equal(grad[-1, sel], grad_s[-1])

    # (2) oracle spot-check of 3 basins over the full 7300 days (through the ABI)
    pick = [0, 335, 670]
    prob = make_problem(model="Hbv", T=T, B=len(pick), M=M, dyn=(), seed=1)
    prob["x"] = x[:, pick].cpu().numpy()
    prob["params"] = p[:, pick].cpu().numpy()
    prob["gflux"] = np.zeros((11, T, len(pick)), np.float32)
    prob["grouted"] = np.zeros((4, T, len(pick)), np.float32)
    prob["grouted"][0] = w[:, pick, 0].cpu().numpy()
    want = run_problem(prob, oracle_path, device="cpu")
    got = run_problem(prob, None, device="cuda:0")
    compare_runs(prob, got, want, label="cfg2")
    assert_close("cfg2 module streamflow", out["streamflow"][:, pick, 0].detach().cpu().numpy(), want["routed"][0])

    # (3) time continuation with cache_states (un-routed fluxes are bit-identical)
    m2 = Hbv({"nmul": M, "dynamic_params": {"Hbv": []}, "cache_states": True}, dev)
    h = T // 2
    with torch.no_grad():
        p1 = torch.cat([p[:h - 1], p[-1:]], 0)
        o1 = m2({"x_phy": x[:h]}, p1)
        o2 = m2({"x_phy": x[h:]}, p[h:])
        for k in ("streamflow_no_rout", "SWE", "AET_hydro", "percolation"):
            assert torch.equal(torch.cat([o1[k], o2[k]], 0), out[k].detach()), k


def test_cfg3_full_size_dynamic_parameters(hip_backend, oracle_path):
    """hbv_1_1p, all 14 parameters dynamic, 671 x 16 x 7300: 4.4 GB of streamed parameters."""
    import hydrodl2_amd
    dev = torch.device("cuda:0")
    T, B, M = 7300, 671, 16
    H = hydrodl2_amd.load_model("hbv_1_1p", "Hbv_1_1p")
    names = list(H(None, dev).parameter_bounds)
    model = H({"nmul": M, "dynamic_params": {"Hbv_1_1p": names}}, dev)
    ny = model.learnable_param_count
    x, p, w = _gen(T, B, ny, 6, dev)
    out, grad = _fwd_bwd(model, x, p, w)
    assert all(torch.isfinite(v).all() for v in out.values()) and torch.isfinite(grad).all()
    assert float(grad[:, :, :14 * M].abs().sum(dim=(1, 2)).min()) > 0.0   # every day gets gradient

    sel = torch.tensor([5, 6, 7, 400, 670], device=dev)
    out_s, grad_s = _fwd_bwd(model, x[:, sel].contiguous(), p[:, sel].contiguous(),
                             w[:, sel].contiguous())
    assert torch.equal(out["streamflow"][:, sel], out_s["streamflow"])
    assert torch.equal(grad[:, sel], grad_s)

    pick = [5, 670]
    prob = make_problem(model="Hbv_1_1p", T=T, B=len(pick), M=M, dyn=tuple(names), seed=1)
    prob["x"] = x[:, pick].cpu().numpy()
    prob["params"] = p[:, pick].cpu().numpy()
    prob["gflux"] = np.zeros((12, T, len(pick)), np.float32)
    prob["grouted"] = np.zeros((4, T, len(pick)), np.float32)
    prob["grouted"][0] = w[:, pick, 0].cpu().numpy()
    want = run_problem(prob, oracle_path, device="cpu")
    assert_close("cfg3 streamflow", out["streamflow"][:, pick, 0].detach().cpu().numpy(), want["routed"][0])
    assert_grad_close("cfg3 g_params", grad[:, pick].cpu().numpy(), want["g_params"], column_groups(ny, M))


def test_hbv2_many_basins(hip_backend):
    """hbv_2, 40k basins x 16 members x 365 days (one GPU's share of a 100k+-basin run):
    basin independence across 10 000 workgroups and the state series."""
    import hydrodl2_amd
    dev = torch.device("cuda:0")
    T, B, M = 365, 40000, 16
    H2 = hydrodl2_amd.load_model("hbv_2", "Hbv_2")
    dyn = ["parBETA", "parK0", "parBETAET"]
    model = H2({"nmul": M, "dynamic_params": {"Hbv_2": dyn}}, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    x, _, w = _gen(T, B, 1, 9, dev)
    pd = torch.rand((T, B, 3 * M), generator=g, device=dev).requires_grad_(True)
    ps = torch.rand((B, 13 * M), generator=g, device=dev).requires_grad_(True)
    xd = {"x_phy": x, "ac_all": torch.rand(B, generator=g, device=dev) * 5000,
          "elev_all": torch.rand(B, generator=g, device=dev) * 3000}
    out = model(xd, (pd, ps))
    (out["streamflow"] * w).sum().backward()
    st = model.get_states()
    assert len(st) == 5 and st[0].shape == (T, B, M)
    assert torch.isfinite(out["streamflow"]).all() and torch.isfinite(pd.grad).all()
    sel = torch.tensor([0, 3, 4, 19999, 39996, 39999], device=dev)
    pd2 = pd.detach()[:, sel].contiguous().requires_grad_(True)
    ps2 = ps.detach()[sel].contiguous().requires_grad_(True)
    xd2 = {"x_phy": x[:, sel].contiguous(), "ac_all": xd["ac_all"][sel], "elev_all": xd["elev_all"][sel]}
    m2 = H2({"nmul": M, "dynamic_params": {"Hbv_2": dyn}}, dev)
    out2 = m2(xd2, (pd2, ps2))
    (out2["streamflow"] * w[:, sel]).sum().backward()
    assert torch.equal(out["streamflow"][:, sel], out2["streamflow"])
    assert torch.equal(st[2][:, sel], m2.get_states()[2])
    # gradients: the full run takes the single-pass streaming adjoint, the 6-basin run the
    # time-parallel one; they agree to rounding (composed chunk maps vs the serial sweep)
    torch.testing.assert_close(pd.grad[:, sel], pd2.grad, rtol=1e-4, atol=1e-6 * float(pd2.grad.abs().max()))
    torch.testing.assert_close(ps.grad[sel], ps2.grad, rtol=1e-4, atol=1e-6 * float(ps2.grad.abs().max()))


def test_cfg5_full_size_on_one_gpu(hip_backend, oracle_backend_path):
    """BASELINE config 5 at its stated size on ONE GPU: hbv_2, 100 000 basins x 16 members x 730 days, three
    dynamic parameters (1.17e9 lane-days; ~60 GB of inputs, trajectory and gradients of the 288 GB), forward +
    backward through the module, twice: a loss on streamflow (the bench's step: four-wave streaming adjoint) and a
    loss on EVERY flux key (the twelve-series adjoint).  Finite; then 34 basins -- both ends, and both sides of every
    boundary between the XCDs' runs of basin groups in the streaming kernels' wave -> basin maps (hbv_stream2.h:
    s2_lane for the adjoint, 3 125 waves per XCD = 12 500 basins; s2_lane_mw for the eight-wave forward, 391
    workgroups per XCD = 12 512 basins) -- reproduce, bit for bit in the outputs and the state series and to rounding in
    the gradients, what they give as a 34-basin problem (other kernels: pipelined forward + time-parallel adjoint), and
    agree with the CPU oracle in every flux key, one state series and both gradients."""
    import hydrodl2_amd
    from tests import seam
    dev = torch.device("cuda:0")
    T, B, M = 730, 100000, 16
    H2 = hydrodl2_amd.load_model("hbv_2", "Hbv_2")
    dyn = ["parBETA", "parK0", "parBETAET"]
    cfgd = {"nmul": M, "dynamic_params": {"Hbv_2": dyn}}
    model = H2(cfgd, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(15)
    x, _, w = _gen(T, B, 1, 15, dev)
    pd = torch.rand((T, B, 3 * M), generator=g, device=dev).requires_grad_(True)
    ps = torch.rand((B, 13 * M), generator=g, device=dev).requires_grad_(True)
    xd = {"x_phy": x, "ac_all": torch.rand(B, generator=g, device=dev) * 5000,
          "elev_all": torch.rand(B, generator=g, device=dev) * 3000}
    picks = [0, 3, 99996, 99999]
    for c in range(1, 8):
        picks += [12500 * c - 1, 12500 * c, 12512 * c - 1, 12512 * c]
    picks = sorted(set(picks))
    assert len(picks) >= 32
    sel = torch.tensor(picks, device=dev)
    keys = [k for k in model.flux_names if k not in ("BFI", "PET_hydro")]
    wk = {k: 0.25 + 0.5 * torch.rand((T, B, 1), generator=g, device=dev) for k in keys[:3]}   # (three dense weights: 0.9 GB)

    def loss_of(out, which, cols=None):
        if which == "streamflow":
            return (out["streamflow"] * (w if cols is None else w[:, cols])).sum()
        tot = 0.0
        for i, k in enumerate(keys):
            wt = wk[keys[i % 3]]
            tot = tot + (out[k] * (wt if cols is None else wt[:, cols])).sum() * (1.0 + 0.1 * i)
        return tot

    full = {}
    for which in ("streamflow", "all"):
        pd.grad = ps.grad = None
        out = model(xd, (pd, ps))
        loss_of(out, which).backward()
        assert out["streamflow"].shape == (T, B, 1)
        assert torch.isfinite(out["streamflow"]).all() and torch.isfinite(pd.grad).all() and torch.isfinite(ps.grad).all()
        full[which] = ({k: out[k][:, sel].detach().clone() for k in keys}, pd.grad[:, sel].clone(), ps.grad[sel].clone(),
                       model.get_states()[2][:, sel].clone())
        del out
    pd.grad = ps.grad = None
    torch.cuda.empty_cache()

    def small(device, lib, which):
        seam.use_library(lib)
        try:
            pd2 = pd.detach()[:, sel].to(device).contiguous().requires_grad_(True)
            ps2 = ps.detach()[sel].to(device).contiguous().requires_grad_(True)
            xd2 = {"x_phy": x[:, sel].to(device).contiguous(), "ac_all": xd["ac_all"][sel].to(device),
                   "elev_all": xd["elev_all"][sel].to(device)}
            m2 = H2(cfgd, torch.device(device))
            o2 = m2(xd2, (pd2, ps2))
            if which == "streamflow":
                l2 = (o2["streamflow"] * w[:, sel].to(device)).sum()
            else:
                l2 = 0.0
                for i, k in enumerate(keys):
                    l2 = l2 + (o2[k] * wk[keys[i % 3]][:, sel].to(device)).sum() * (1.0 + 0.1 * i)
            l2.backward()
            return {k: o2[k].detach() for k in keys}, pd2.grad, ps2.grad, m2.get_states()[2].detach()
        finally:
            seam.use_library(None)

    for which in ("streamflow", "all"):
        fo, fgd, fgs, fsm = full[which]
        o2, gd2, gs2, sm2 = small("cuda:0", None, which)
        for k in keys:
            assert torch.equal(fo[k], o2[k]), (which, k)
        assert torch.equal(fsm, sm2), which                       # the soil-moisture series (hbv_2.py:571-575)
        torch.testing.assert_close(fgd, gd2, rtol=1e-4, atol=1e-6 * float(gd2.abs().max()))
        torch.testing.assert_close(fgs, gs2, rtol=1e-4, atol=1e-6 * float(gs2.abs().max()))
        oo, gdo, gso, smo = small("cpu", oracle_backend_path, which)
        for k in keys:
            assert_close(f"cfg5 {which} {k}", o2[k].cpu().numpy(), oo[k].numpy())
        assert_close(f"cfg5 {which} SM series", sm2.cpu().numpy(), smo.numpy())
        assert_grad_close(f"cfg5 {which} g_dyn", gd2.cpu().numpy(), gdo.numpy(), column_groups(gdo.shape[-1], M))
        assert_grad_close(f"cfg5 {which} g_sta", gs2.cpu().numpy(), gso.numpy(), column_groups(gso.shape[-1], M))


def _slice_problem(prob, pick):
    """The sub-problem of basins `pick` (every per-basin array of tests/abi_util.make_problem)."""
    sub = dict(prob)
    sub["B"] = len(pick)
    sub["x"] = np.ascontiguousarray(prob["x"][:, pick])
    sub["params"] = np.ascontiguousarray(prob["params"][:, pick])
    sub["gflux"] = np.ascontiguousarray(prob["gflux"][:, :, pick])
    sub["grouted"] = np.ascontiguousarray(prob["grouted"][:, :, pick])
    for k in ("ac", "elev"):
        if k in prob:
            sub[k] = np.ascontiguousarray(prob[k][pick])
    if "drop" in prob:
        sub["drop"] = np.ascontiguousarray(prob["drop"][:, pick])
    return sub


@pytest.mark.parametrize("model,M,B,T,dyn", [
    ("Hbv_2", 16, 8200, 300, ("parBETA", "parK0", "parBETAET")),
    ("Hbv_2_hourly", 4, 32800, 256, ("parBETA", "parK0", "parBETAET")),
    ("Hbv", 16, 8200, 260, ("parBETA", "parBETAET")),
    ("Hbv_1_1p", 16, 8200, 256, ()),
])
def test_streaming_kernels_oracle_spot_check(model, M, B, T, dyn, hip_backend, oracle_path):
    """>= 2048 wavefronts of state, so BOTH directions take the streaming kernels (packed trajectory,
    hbv_stream2.h): three basins pulled out of the big run -- first, middle, last (the last wave is
    partly filled) -- against the oracle run on those basins alone.  Loss on every flux series."""
    prob = make_problem(model=model, T=T, B=B, M=M, dyn=dyn, drop_frac=0.25 if dyn else 0.0, seed=5)
    got = run_problem(prob, None, device="cuda:0", x_grad=True, keep_traj=False)   # (the trajectory is up to 5.5 GB)
    pick = [0, B // 2 + 1, B - 1]
    want = run_problem(_slice_problem(prob, pick), oracle_path, device="cpu", x_grad=True)
    compare_runs(prob, got, want, label=f"stream {model}", basins=pick)


@pytest.mark.parametrize("fam,cls,B,T,dyn", [
    ("hbv", "Hbv", 8200, 640, ["parBETA", "parBETAET"]),             # raw [T,B,ny] parameters: 4.4 GB
    ("hbv_2", "Hbv_2", 24600, 700, ["parBETA", "parK0", "parBETAET"]),  # packed trajectory records: 4.4 GB
])
def test_tensors_beyond_4gib(fam, cls, B, T, dyn, hip_backend, oracle_path):
    """Beyond a buffer descriptor's 4 GiB: the streaming kernels rebase their descriptors on the day's
    rows.  The module on the GPU at full size against the same module on the CPU oracle for three basins
    (first, middle, last) -- inputs generated on the device, so the test costs seconds, not minutes."""
    import hydrodl2_amd
    from tests import seam
    dev = torch.device("cuda:0")
    M = 16
    C = hydrodl2_amd.load_model(fam, cls)
    conf = {"nmul": M, "dynamic_params": {cls: dyn}}
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    x, _, w = _gen(T, B, 1, 11, dev)
    xd = {"x_phy": x}
    if cls == "Hbv_2":
        params = (torch.rand((T, B, len(dyn) * M), generator=g, device=dev).requires_grad_(True),
                  torch.rand((B, (16 - len(dyn)) * M), generator=g, device=dev).requires_grad_(True))
        xd["ac_all"] = torch.rand(B, generator=g, device=dev) * 5000
        xd["elev_all"] = torch.rand(B, generator=g, device=dev) * 3000
        leaves = list(params)
        assert (T + 1) * B * M * 16 > 1 << 32
    else:
        model0 = C(conf, dev)
        params = torch.randn((T, B, model0.learnable_param_count), generator=g, device=dev).requires_grad_(True)
        leaves = [params]
        assert params.numel() * 4 > 1 << 32
    model = C(conf, dev)
    out = model(xd, params)
    (out["streamflow"] * w).sum().backward()
    assert torch.isfinite(out["streamflow"]).all()

    pick = [0, B // 2 + 1, B - 1]
    sel = torch.tensor(pick, device=dev)

    def cut(t, axis):
        return t.detach().index_select(axis, sel).cpu().contiguous()
    xd_c = {k: cut(v, 1 if k == "x_phy" else 0) for k, v in xd.items()}
    if cls == "Hbv_2":
        params_c = (cut(params[0], 1).requires_grad_(True), cut(params[1], 0).requires_grad_(True))
        leaves_c = list(params_c)
    else:
        params_c = cut(params, 1).requires_grad_(True)
        leaves_c = [params_c]
    seam.use_library(oracle_path)
    try:
        ref = C(conf, torch.device("cpu"))
        out_c = ref(xd_c, params_c)
        (out_c["streamflow"] * cut(w, 1)).sum().backward()
    finally:
        seam.use_library(None)
    assert_close(f"4gib {cls} streamflow", out["streamflow"][:, sel].detach().cpu().numpy(), out_c["streamflow"].detach().numpy())
    for a, b in zip(leaves, leaves_c):
        axis = 1 if a.dim() == 3 else 0
        assert_grad_close(f"4gib {cls} grad", a.grad.index_select(axis, sel).cpu().numpy(), b.grad.numpy(),
                          column_groups(b.shape[-1], M))


@pytest.mark.parametrize("mode", ["overlap", "early"])
def test_gradient_fill_schedules_are_equivalent(mode, hip_backend, monkeypatch):
    """The [T,B,ny] gradient fill beside the adjoint (static and routing gradients go through a separate last row)
    and the fill requested from the forward call (hbvx_fwd_out.zero_ptr; at 250 workgroups the launch has no idle CUs
    to spare, so a fill kernel runs behind the forward) give bit-identical gradients to the plain fill-then-adjoint
    order (a 310 MB gradient: above the size these paths start at)."""
    import hydrodl2_amd
    from hydrodl2_amd import ops
    dev = torch.device("cuda:0")
    T, B, M = 400, 1000, 16
    model = hydrodl2_amd.load_model("hbv", "Hbv")({"nmul": M, "dynamic_params": {"Hbv": []}}, dev)
    x, p, w = _gen(T, B, model.learnable_param_count, 21, dev)
    assert p.numel() >= ops._EARLY_ZERO_MIN

    def run(overlap, early):
        monkeypatch.setattr(ops, "_FILL_OVERLAP", overlap)
        monkeypatch.setattr(ops, "_EARLY_ZERO", "1" if early else "0")
        out, grad = _fwd_bwd(model, x, p, w, keys=("streamflow", "SWE"))
        return out["streamflow"].detach().clone(), grad.clone()
    s0, g0 = run(False, False)
    s1, g1 = run(True, mode == "early")
    if mode == "early":
        assert not ops.get_library().zero_in_launch()
    assert torch.equal(s0, s1)
    assert torch.equal(g0, g1)
    assert float(g1[:-1].abs().max()) == 0.0 and float(g1[-1].abs().max()) > 0.0


@pytest.mark.parametrize("family,cls,dyn,dy_drop,warm_up", [
    ("hbv", "Hbv", [], 0.0, 0), ("hbv", "Hbv", ["parBETA", "parBETAET"], 0.4, 40), ("hbv_adj", "HbvAdj", ["parBETAET"], 0.0, 0)])
def test_zero_fill_inside_the_forward_launch(family, cls, dyn, dy_drop, warm_up, hip_backend, monkeypatch):
    """hbvx_fwd_out.zero_ptr (ABI 10): with 150 workgroups of the recurrence on 256 CUs, surplus workgroups of the
    pipelined forward's own launch write zeros into the [T,B,ny] gradient while the recurrence runs, and backward
    fills what they left (hbvx_zero_rest).  Outputs and gradients are bit-identical to the whole fill beside the
    adjoint, over three steps with fresh inputs (a piece left unfilled, or dynamic columns stored before their zeros
    landed, would differ), and the library reports that the launch carried the fill.  (Default: the implicit scheme
    only -- ops._EARLY_ZERO.)"""
    import hydrodl2_amd
    from hydrodl2_amd import ops
    dev = torch.device("cuda:0")
    T, B, M = 640, 600, 16
    conf = {"nmul": M, "dy_drop": dy_drop, "warm_up": warm_up, "dynamic_params": {cls: dyn}}
    lib = ops.get_library()

    def run(early):
        monkeypatch.setattr(ops, "_EARLY_ZERO", "1" if early else "0")
        model = hydrodl2_amd.load_model(family, cls)(conf, dev)      # a fresh module: no memo of the other mode
        res = []
        for rnd in range(3):
            x, p, w = _gen(T, B, model.learnable_param_count, 51 + rnd, dev)
            assert p.numel() >= ops._EARLY_ZERO_MIN
            torch.manual_seed(9 + rnd)                                # the dy_drop masks come from the CPU generator
            pl = p.detach().clone().requires_grad_(True)
            out = model({"x_phy": x}, pl)
            key = "streamflow" if "streamflow" in out else next(iter(out))
            took = lib.zero_in_launch()
            (out[key] * w[warm_up:]).sum().backward()
            res.append((out[key].detach().clone(), pl.grad.clone(), took, lib.last_dispatch(0)))
        return res
    plain, early = run(False), run(True)
    for (s0, g0, t0, _), (s1, g1, t1, fam) in zip(plain, early):
        assert fam == "pipe"
        assert not t0 and t1
        assert torch.equal(s0, s1)
        assert torch.equal(g0, g1)
        assert float(g1.abs().max()) > 0.0


@pytest.mark.parametrize("dy_drop", [0.0, 0.4])
def test_gated_fill_with_dynamic_columns_is_equivalent(dy_drop, hip_backend, monkeypatch):
    """A gradient tensor with FEW dynamic columns (BETA + BETAET of 13 parameters: the delta-MG default) is filled
    densely on the second stream beside the reading passes of the adjoint, and the adjoint's storing kernel waits for
    the fill through hbvx_bwd_io.store_gate.  Bit-identical to the plain fill-then-adjoint order -- without the gate
    the two race (that was measured: wrong gradients) -- also with dy_drop masks, whose dropped basins accumulate
    into the separate static row."""
    import hydrodl2_amd
    from hydrodl2_amd import ops
    dev = torch.device("cuda:0")
    T, B, M = 400, 1000, 16
    conf = {"nmul": M, "dy_drop": dy_drop, "dynamic_params": {"Hbv": ["parBETA", "parBETAET"]}}
    model = hydrodl2_amd.load_model("hbv", "Hbv")(conf, dev)
    x, p, w = _gen(T, B, model.learnable_param_count, 23, dev)
    assert p.numel() >= ops._EARLY_ZERO_MIN

    def run(overlap):
        monkeypatch.setattr(ops, "_FILL_OVERLAP", overlap)
        torch.manual_seed(3)        # the dy_drop masks come from the CPU generator
        out, grad = _fwd_bwd(model, x, p, w, keys=("streamflow",))
        return out["streamflow"].detach().clone(), grad.clone()
    s0, g0 = run(False)
    for _ in range(3):              # a race would not show every time
        s1, g1 = run(True)
        assert torch.equal(s0, s1)
        assert torch.equal(g0, g1)
    dyn_cols = torch.zeros(p.shape[-1], dtype=torch.bool, device=dev)
    dyn_cols[0:M] = True
    dyn_cols[12 * M:13 * M] = True
    assert float(g1[:-1][:, :, ~dyn_cols].abs().max()) == 0.0 and float(g1[:-1][:, :, dyn_cols].abs().max()) > 0.0


@pytest.mark.parametrize("dyn,dy_drop,warm_up", [([], 0.0, 0), (["parBETA", "parBETAET"], 0.4, 50)])
def test_persistent_gradient_buffer_equals_fresh(dyn, dy_drop, warm_up, hip_backend):
    """Module key grad_buffer='persistent': the [T,B,ny] gradient is written into a buffer the module keeps (zero-filled
    once; every step rewrites the dynamic columns and the last row only).  Bit-identical to the default over three steps
    with new inputs, new dy_drop masks and two loss patterns, out of the SAME storage from the second step on (the
    first step's tensor may be a copy autograd made).  With dy_drop masks the step configuration is a per-call copy:
    the buffer lives in the memo it shares with the cached original (round 5: it used to be rebuilt -- and refilled --
    every step, which this assertion caught as [A, B, A])."""
    import hydrodl2_amd
    from hydrodl2_amd import ops
    dev = torch.device("cuda:0")
    T, B, M = 400, 1000, 16
    conf = {"nmul": M, "dy_drop": dy_drop, "warm_up": warm_up, "dynamic_params": {"Hbv": dyn}}
    C = hydrodl2_amd.load_model("hbv", "Hbv")
    fresh, keep = C(dict(conf), dev), C(dict(conf, grad_buffer="persistent"), dev)
    ptrs = []
    for rnd in range(3):
        x, p, w = _gen(T, B, fresh.learnable_param_count, 31 + rnd, dev)
        assert p.numel() >= ops._EARLY_ZERO_MIN
        keys = ("streamflow",) if rnd != 1 else ("streamflow", "SWE", "BFI")
        res = []
        for m in (fresh, keep):
            torch.manual_seed(5 + rnd)        # the dy_drop masks come from the CPU generator
            pl = p.detach().clone().requires_grad_(True)
            out = m({"x_phy": x}, pl)
            loss = sum((out[k] * (w[-out[k].shape[0]:] if out[k].dim() == 3 else 1.0)).sum() for k in keys)
            loss.backward()
            res.append(pl.grad.clone())
            if m is keep:
                ptrs.append(pl.grad.untyped_storage().data_ptr())
            del pl, out, loss
        assert torch.equal(res[0], res[1]), f"round {rnd}"
    assert len(set(ptrs[1:])) == 1, ptrs
    with pytest.raises(ValueError, match="grad_buffer"):
        C(dict(conf, grad_buffer="sometimes"), dev)({"x_phy": x}, p)


def test_persistent_gradient_buffer_of_the_implicit_scheme(hip_backend):
    """`HbvAdj` with grad_buffer='persistent' (its step configurations are cached per shape, so the buffer survives
    the call): gradients bit-identical to the default over three steps, one storage."""
    import hydrodl2_amd
    from hydrodl2_amd import ops
    dev = torch.device("cuda:0")
    T, B, M = 400, 1000, 16
    conf = {"nmul": M, "dynamic_params": {"HbvAdj": ["parBETAET"]}}
    C = hydrodl2_amd.load_model("hbv_adj", "HbvAdj")
    fresh, keep = C(dict(conf), dev), C(dict(conf, grad_buffer="persistent"), dev)
    ptrs = []
    for rnd in range(3):
        x, p, w = _gen(T, B, fresh.learnable_param_count, 41 + rnd, dev)
        assert p.numel() >= ops._EARLY_ZERO_MIN
        res = []
        for m in (fresh, keep):
            torch.manual_seed(9 + rnd)
            pl = p.detach().clone().requires_grad_(True)
            out = m({"x_phy": x}, pl)
            (out["flow_sim"] * w).sum().backward()
            res.append(pl.grad.clone())
            if m is keep:
                ptrs.append(pl.grad.untyped_storage().data_ptr())
            del pl, out
        assert torch.equal(res[0], res[1]), f"round {rnd}"
    assert len(set(ptrs[1:])) == 1, ptrs      # (the first step's tensor may be a copy autograd made; from then on: one storage)
