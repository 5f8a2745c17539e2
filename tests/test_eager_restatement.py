"""CPU tier: the pure-torch eager restatement (oracle/hbv_torch_eager.py, the `cpu_baseline.eager`
leg of bench.py) against the fixtures the reference itself produced."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from . import golden_cases as gc
from .helpers import compare, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _eager():
    spec = importlib.util.spec_from_file_location("hbv_torch_eager", os.path.join(ROOT, "oracle", "hbv_torch_eager.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# the fixture cases inside the restatement's scope: Hbv, no warm-up, no dy_drop, no muwts
CASES = ["cfg1_hbv_default", "hbv_static_m16", "hbv_static_m16_sf", "hbv_ties", "hbv_short"]


@pytest.mark.parametrize("name", CASES)
def test_eager_matches_reference(name):
    spec = gc.CASES[name]
    cfg = spec["config"] or {}
    inp = gc.build_inputs(name)
    x = torch.from_numpy(inp["x_phy"])
    p = torch.from_numpy(inp["parameters"]).requires_grad_(True)
    out = _eager().hbv_eager(x, p, cfg.get("nmul", 1), dynamic=cfg.get("dynamic_params", {}).get("Hbv", ()))
    res = {f"out/{k}": v.detach().numpy() for k, v in out.items()}
    keys = gc.loss_keys(name)
    if keys:
        loss = sum((torch.from_numpy(gc.loss_weight(name, k, out[k].shape)) * out[k]).sum() for k in keys)
        loss.backward()
        res["grad/parameters"] = p.grad.numpy()
    ref = load_golden(name)
    sub = {k: ref[k] for k in ref.files if k in res}

    class View:
        files = list(sub)

        def __getitem__(self, k):
            return sub[k]
    assert len(sub) >= 17
    compare(name, res, View())


def test_eager_dynamic_matches_oracle(oracle_backend):
    """Dynamic parBETA + parBETAET (not covered by a warm-up-free fixture): against the package's
    host logic driving the C oracle."""
    import hydrodl2_amd
    T, B, M = 40, 4, 3
    g = torch.Generator().manual_seed(5)
    x = torch.stack([torch.clamp((torch.rand(T, B, generator=g) - 0.6) * 40, min=0),
                     torch.randn(T, B, generator=g) * 6 + 1, torch.rand(T, B, generator=g) * 5], -1)
    dyn = ["parBETA", "parBETAET"]
    p1 = torch.randn(T, B, 13 * M + 2, generator=g).requires_grad_(True)
    p2 = p1.detach().clone().requires_grad_(True)
    w = torch.randn(T, B, 1, generator=g)
    a = _eager().hbv_eager(x, p1, M, dynamic=dyn)
    model = hydrodl2_amd.load_model("hbv", "Hbv")({"nmul": M, "dynamic_params": {"Hbv": dyn}}, torch.device("cpu"))
    b = model({"x_phy": x}, p2)
    (a["streamflow"] * w).sum().backward()
    (b["streamflow"] * w).sum().backward()
    for k in a:
        np.testing.assert_allclose(a[k].detach().numpy(), b[k].detach().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
    scale = p2.grad.abs().max().item()
    np.testing.assert_allclose(p1.grad.numpy(), p2.grad.numpy(), rtol=1e-3, atol=1e-5 * scale)
