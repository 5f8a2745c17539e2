"""Shared helpers: run a golden case through hydrodl2_amd and compare."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import golden_cases as gc
from .abi_util import GROUP_FLOOR, REPORT, ROUTE_ATOL_REL, column_groups

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Stated tolerances (fp32; pow / sum-order / sigmoid differ by ulps between
# implementations).  BASELINE.md §2 proposal: fluxes rtol 1e-4 / atol 1e-5,
# gradients rtol 1e-3 / atol 1e-6 relative to the gradient's scale.
from .abi_util import FLUX_ATOL, FLUX_RTOL, GRAD_ATOL_REL, GRAD_RTOL  # noqa: E402  (one statement of the tolerances)


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))


def run_case(name: str, device: str):
    """Forward (+ backward) of golden case `name` through the package on `device`."""
    import hydrodl2_amd
    spec = gc.CASES[name]
    dev = torch.device(device)
    cls = hydrodl2_amd.load_model(spec["model"].lower(), spec["model"])
    cfg = spec["config"]
    model = cls(None if cfg is None else dict(cfg), dev)
    inp = gc.build_inputs(name)
    res = {}

    x = torch.from_numpy(inp["x_phy"]).to(dev)
    if spec.get("x_grad"):
        x.requires_grad_(True)
    x_dict = {"x_phy": x}
    if "muwts" in inp:
        x_dict["muwts"] = torch.from_numpy(inp["muwts"]).to(dev)
    if spec["model"] in ("Hbv_2", "Hbv_2_hourly"):
        pd = torch.from_numpy(inp["p_dyn"]).to(dev).requires_grad_(True)
        ps = torch.from_numpy(inp["p_sta"]).to(dev).requires_grad_(True)
        x_dict["ac_all"] = torch.from_numpy(inp["ac_all"]).to(dev)
        x_dict["elev_all"] = torch.from_numpy(inp["elev_all"]).to(dev)
        params = (pd, ps)
        leaves = [("p_dyn", pd), ("p_sta", ps)]
        if spec["model"] == "Hbv_2_hourly":
            pr = torch.from_numpy(inp["p_distr"]).to(dev).requires_grad_(True)
            x_dict["outlet_topo"] = torch.from_numpy(inp["outlet_topo"]).to(dev)
            x_dict["areas"] = torch.from_numpy(inp["areas"]).to(dev)
            params = (pd, ps, pr)
            leaves.append(("p_distr", pr))
    else:
        p = torch.from_numpy(inp["parameters"]).to(dev).requires_grad_(True)
        params = p
        leaves = [("parameters", p)]
    if spec.get("x_grad"):
        leaves.append(("x_phy", x))
    if "torch_seed" in spec:
        torch.manual_seed(spec["torch_seed"])

    if spec.get("two_call"):
        T = spec["T"]
        h = T // 2
        with torch.no_grad():
            p1 = torch.cat([p[:h - 1], p[-1:]], 0)
            out1 = model({"x_phy": x[:h]}, p1)
            st1 = torch.stack(list(model.get_states())).cpu().numpy()
            out2 = model({"x_phy": x[h:]}, p[h:])
            st2 = torch.stack(list(model.get_states())).cpu().numpy()
        for k, v in out1.items():
            res[f"out1/{k}"] = v.cpu().numpy()
        for k, v in out2.items():
            res[f"out2/{k}"] = v.cpu().numpy()
        res["states1"], res["states2"] = st1, st2
        return res

    keys = gc.loss_keys(name)
    if keys:
        out = model(x_dict, params)
    else:
        with torch.no_grad():
            out = model(x_dict, params)
    for k, v in out.items():
        res[f"out/{k}"] = v.detach().cpu().numpy()
    res["states"] = torch.stack([s.detach() for s in model.get_states()]).cpu().numpy()
    if keys:
        loss = 0.0
        for k in keys:
            w = torch.from_numpy(gc.loss_weight(name, k, out[k].shape)).to(dev)
            loss = loss + (w * out[k]).sum()
        loss.backward()
        res["loss"] = np.array(float(loss.detach()))
        for lname, leaf in leaves:
            g = leaf.grad if leaf.grad is not None else torch.zeros_like(leaf)
            res[f"grad/{lname}"] = g.cpu().numpy()
    return res


def compare(name: str, res: dict, ref, report=None, flux_rtol=FLUX_RTOL, flux_atol=FLUX_ATOL,
            grad_rtol=GRAD_RTOL, grad_atol_rel=GRAD_ATOL_REL, grad_outlier_frac=0.0, grad_outlier_atol_rel=0.0):
    """Assert `res` matches the golden record `ref` within the stated tolerances.  `grad_outlier_*` (0 for every
    committed fixture): at most that fraction of a gradient tensor's elements (or four) may miss the per-group tolerance, by no
    more than that multiple of the tensor's largest entry -- for randomly drawn cases (tests/test_live_reference.py),
    where an element can be a sum whose terms cancel four orders of magnitude deep."""
    bad = []
    nmul = (gc.CASES[name]["config"] or {}).get("nmul", 1) if name in gc.CASES else 1
    if name in gc.CASES:
        grad_atol_rel = gc.CASES[name].get("grad_atol_rel", grad_atol_rel)      # a case's own, with its reason beside it
    for key in ref.files:
        if key in ("torch_version", "loss"):
            continue
        a, b = np.asarray(res[key], dtype=np.float64), np.asarray(ref[key], dtype=np.float64)
        assert a.shape == b.shape, f"{name}:{key}: shape {a.shape} vs {b.shape}"
        if a.size == 0:
            continue
        if key.startswith("grad/"):
            # 1e-6 x the largest gradient of the element's own parameter group (the nmul members of one physical
            # parameter; a routing column, a forcing channel, a gage-routing parameter are groups of their own)
            w = b.shape[-1]
            labels = np.arange(w) if key in ("grad/x_phy", "grad/p_distr") else column_groups(w, nmul)
            colmax = np.abs(b).reshape(-1, w).max(0)
            # (absolute floor: a gradient tensor whose largest entry is below 1e-2 -- a record in which no water moves -- is
            #  compared at 2e-11; every committed fixture's tensors are larger and unaffected)
            floor = max(GROUP_FLOOR * colmax.max(), 1e-5)
            scale = np.array([max(colmax[labels == g].max(), floor) * (ROUTE_ATOL_REL / GRAD_ATOL_REL if g < 0 else 1.0)
                              for g in labels])
            tol = grad_atol_rel * scale + grad_rtol * np.abs(b)
        else:
            tol = flux_atol + flux_rtol * np.abs(b)
        err = np.abs(a - b)
        nbad = int((err > tol).sum())
        if nbad and key.startswith("grad/") and grad_outlier_frac > 0.0:
            out = err > tol
            if out.sum() <= max(grad_outlier_frac * a.size, 4) and err[out].max() <= grad_outlier_atol_rel * np.abs(b).max():
                nbad = 0
        REPORT.append((f"{name}:{key}", float(err.max()), float((err / tol).max()), nbad, a.size))
        if report is not None:
            report.append((key, float(err.max()), float((err / (np.abs(b) + 1e-12)).max()), nbad,
                           a.size))
        if nbad:
            i = np.unravel_index(np.argmax(err - tol), err.shape)
            bad.append(f"{key}: {nbad}/{a.size} outside tol; worst at {i}: got {a[i]!r} "
                       f"want {b[i]!r}")
    assert not bad, f"{name}: " + "; ".join(bad)
