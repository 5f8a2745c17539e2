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
