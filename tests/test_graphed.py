"""GPU tier: a module built with `graph=True` (hydrodl2_amd/graphed.py) replays its forward and backward as HIP
graphs.  Same kernels in the same order on static buffers: outputs and gradients must equal the eager module's bit
for bit, call after call, for fresh inputs, for a second gradient pattern, and with the host generator advanced
exactly as the eager path advances it (hbv.py:240)."""
import numpy as np
import pytest
import torch

import hydrodl2_amd
from tests.synth import forcing, raw_parameters

pytestmark = pytest.mark.gpu


def _models(cls_file, cls, cfg):
    C = hydrodl2_amd.load_model(cls_file, cls)
    dev = torch.device("cuda:0")
    return C(dict(cfg), dev), C(dict(cfg, graph=True), dev)


def _inputs(T, B, ny, seed):
    x = torch.from_numpy(forcing(T, B, seed)).cuda()
    p = torch.from_numpy(raw_parameters(T, B, ny, seed + 1)).cuda()
    return x, p


@pytest.mark.parametrize("cls_file,cls,dyn", [("hbv", "Hbv", ["parBETA", "parBETAET"]), ("hbv", "Hbv", []),
                                              ("hbv_1_1p", "Hbv_1_1p", ["parBETA", "parK0", "parBETAET"])])
def test_graph_replay_equals_eager(hip_backend, cls_file, cls, dyn):
    T, B, M = 96, 9, 16
    cfg = {"nmul": M, "warm_up": 32, "dynamic_params": {cls: dyn}}
    eager, graphed = _models(cls_file, cls, cfg)
    ny = eager.learnable_param_count
    torch.manual_seed(5)
    for rnd in range(3):                      # call 0 captures, calls 1-2 replay with fresh inputs
        x, p = _inputs(T, B, ny, 10 + rnd)
        w = torch.randn(T - 32, B, 1, device="cuda", generator=torch.Generator("cuda").manual_seed(rnd))
        res = []
        for m in (eager, graphed):
            state = torch.get_rng_state()
            pl = p.clone().requires_grad_(True)
            out = m({"x_phy": x}, pl)
            key = "streamflow" if rnd < 2 else "AET_hydro"          # round 2: another gradient pattern
            loss = (out[key] * w).sum() + (out["BFI"].sum() if rnd == 1 else 0.0)
            loss.backward()
            res.append(({k: v.detach().cpu().numpy().copy() for k, v in out.items()}, pl.grad.cpu().numpy().copy(),
                        [s.cpu().numpy().copy() for s in m.get_states()], torch.get_rng_state()))
            if m is eager:
                torch.set_rng_state(state)
        (oe, ge, se, re_), (og, gg, sg, rg) = res
        assert set(oe) == set(og)
        for k in oe:
            np.testing.assert_array_equal(oe[k], og[k], err_msg=f"round {rnd} {k}")
        np.testing.assert_array_equal(ge, gg, err_msg=f"round {rnd} gradient")
        for a, b in zip(se, sg):
            np.testing.assert_array_equal(a, b)
        assert torch.equal(re_, rg), "the graphed call must advance the host generator like the eager call"


def test_graph_inference_and_refusals(hip_backend):
    T, B, M = 64, 5, 4
    cfg = {"nmul": M, "dynamic_params": {"Hbv": ["parBETA"]}}
    eager, graphed = _models("hbv", "Hbv", cfg)
    x, p = _inputs(T, B, eager.learnable_param_count, 3)
    with torch.no_grad():
        a, b = eager({"x_phy": x}, p), graphed({"x_phy": x}, p)
        b2 = graphed({"x_phy": x}, p)         # same objects, same versions: replay without copies
    for k in a:
        np.testing.assert_array_equal(a[k].cpu().numpy(), b[k].cpu().numpy(), err_msg=k)
        np.testing.assert_array_equal(a[k].cpu().numpy(), b2[k].cpu().numpy(), err_msg=k)
    p.add_(0.25)                              # in place: the version counter moves, the copy must happen
    with torch.no_grad():
        a, b = eager({"x_phy": x}, p), graphed({"x_phy": x}, p)
    np.testing.assert_array_equal(a["streamflow"].cpu().numpy(), b["streamflow"].cpu().numpy())
    C = hydrodl2_amd.load_model("hbv", "Hbv")
    drop = C(dict(cfg, graph=True, dy_drop=0.5), torch.device("cuda:0"))
    with pytest.raises(ValueError, match="dy_drop"):
        drop({"x_phy": x}, p)
    with pytest.raises(ValueError, match="muwts"):
        graphed({"x_phy": x, "muwts": torch.rand(B, M, device="cuda")}, p)


def _family_case(cls, seed):
    """(config, x_dict, parameter tensors, loss key) of one class of the family on cuda:0 (T = 96, B = 9)."""
    from tests import synth
    T, B, M = 96, 9, 4
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    x = synth.forcing(T, B, seed)
    if cls == "Hbv_2":
        dyn = ["parBETA", "parK0", "parBETAET"]
        cfg = {"nmul": M, "dynamic_params": {cls: dyn}, "routing": True}
        xd = {"x_phy": t(x), "ac_all": t(synth.uniform((B,), seed, 7) * np.float32(5000)),
              "elev_all": t(synth.uniform((B,), seed, 8) * np.float32(3000))}
        ps = [t(synth.unit_parameters((T, B, 3 * M), seed, 4)), t(synth.unit_parameters((B, 13 * M + 2), seed, 6))]
        return ("hbv_2", cfg, xd, ps, "streamflow", T)
    if cls == "Hbv_2_hourly":
        G = 3
        dyn = ["parBETA", "parF0"]
        cfg = {"nmul": M, "dynamic_params": {cls: dyn}}
        topo = (synth.uniform((G, B), seed, 13) < np.float32(0.45)).astype(np.float32)
        topo[np.arange(B) % G, np.arange(B)] = 1.0
        xd = {"x_phy": t(x * np.array([1 / 8.0, 1.0, 1 / 24.0], np.float32)),
              "ac_all": t(synth.uniform((B,), seed, 7) * np.float32(5000)),
              "elev_all": t(synth.uniform((B,), seed, 8) * np.float32(3000)), "outlet_topo": t(topo),
              "areas": t(synth.uniform((B,), seed, 14) * np.float32(90) + np.float32(5))}
        ps = [t(synth.unit_parameters((T, B, 2 * M), seed, 4)), t(synth.unit_parameters((B, 17 * M), seed, 6)),
              t(synth.unit_parameters((int(topo.sum()), 3), seed, 15))]
        return ("hbv_2_hourly", cfg, xd, ps, "streamflow", T)
    assert cls == "HbvAdj"
    cfg = {"nmul": M, "warm_up": 16, "dynamic_params": {cls: ["parBETAET"]}}
    ps = [t(synth.raw_parameters(T, B, 13 * M + 2, seed + 1))]
    return ("hbv_adj", cfg, {"x_phy": t(x)}, ps, "flow_sim", T - 16)


@pytest.mark.parametrize("cls", ["Hbv_2", "Hbv_2_hourly", "HbvAdj"])
def test_graph_replay_equals_eager_for_the_rest_of_the_family(hip_backend, cls):
    """`graph=True` on the tuple-parameter classes and the implicit scheme (hbv_2.py:324-390, hbv_2_hourly.py:376-449,
    hbv_adj.py:227-330): outputs, every parameter tensor's gradient and the host generator equal the eager module's
    bit for bit over three calls (capture, two replays with fresh inputs), and gradients ACCUMULATE across steps
    into a leaf's .grad (the static gradient buffer is never what autograd adopts)."""
    torch.manual_seed(11)
    res = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(11)
        rounds = []
        model = None
        acc = None
        for rnd in range(3):
            fam, cfg, xd, ps, key, T_out = _family_case(cls, 40 + rnd)
            if model is None:
                C = hydrodl2_amd.load_model(fam, cls)
                model = C(dict(cfg, graph=(mode == "graph")), torch.device("cuda:0"))
                if cls == "Hbv_2_hourly":
                    topo_keep, areas_keep = xd["outlet_topo"], xd["areas"]
            if cls == "Hbv_2_hourly":      # the same gage topology objects every round: one capture
                fam, cfg, xd, ps, key, T_out = _family_case(cls, 40)
                xd["outlet_topo"], xd["areas"] = topo_keep, areas_keep
                xd["x_phy"] = xd["x_phy"] * (1.0 + 0.1 * rnd)
            if acc is None:
                acc = [p.clone().requires_grad_(True) for p in ps]        # leaves that live across the rounds
            else:
                with torch.no_grad():
                    for a, p in zip(acc, ps):
                        a.copy_(p)
            params = tuple(acc) if len(acc) > 1 else acc[0]
            out = model(xd, params)
            w = torch.randn(out[key].shape, device="cuda", generator=torch.Generator("cuda").manual_seed(rnd))
            (out[key] * w).sum().backward()                               # .grad accumulates over the three rounds
            rounds.append(({k: v.detach().cpu().numpy().copy() for k, v in out.items()},
                           [a.grad.detach().cpu().numpy().copy() for a in acc], torch.get_rng_state()))
        res[mode] = rounds
    for rnd, ((oe, ge, re_), (og, gg, rg)) in enumerate(zip(res["eager"], res["graph"])):
        assert set(oe) == set(og)
        for k in oe:
            np.testing.assert_array_equal(oe[k], og[k], err_msg=f"{cls} round {rnd} {k}")
        for i, (a, b) in enumerate(zip(ge, gg)):
            np.testing.assert_array_equal(a, b, err_msg=f"{cls} round {rnd} accumulated gradient of tensor {i}")
        assert torch.equal(re_, rg), "the graphed call must advance the host generator like the eager call"
    assert np.abs(res["graph"][2][1][0]).max() > 0


def test_graph_key_tells_dtype_and_structure(hip_backend):
    """A float64 forcing tensor of the same shape is another capture (copy_ would convert silently), and a new
    `outlet_topo` object re-captures the hourly model (its pair lists are baked into the graph)."""
    fam, cfg, xd, ps, key, _ = _family_case("Hbv_2_hourly", 50)
    model = hydrodl2_amd.load_model(fam, "Hbv_2_hourly")(dict(cfg, graph=True), torch.device("cuda:0"))
    with torch.no_grad():
        a = model(xd, tuple(ps))[key].clone()
        n0 = len(model._graph_cache)
        xd2 = dict(xd)
        topo2 = xd["outlet_topo"].clone()                    # another topology with as many (gage, unit) pairs
        one = (topo2[0] == 1).nonzero()[-1, 0]
        zero = (topo2[0] == 0).nonzero()[0, 0]
        topo2[0, one], topo2[0, zero] = 0.0, 1.0
        xd2["outlet_topo"] = topo2
        b = model(xd2, tuple(ps))[key].clone()
        assert len(model._graph_cache) == n0 + 1 and not torch.equal(a, b)
        c = model(xd, tuple(ps))[key]
    assert torch.equal(a, c)
