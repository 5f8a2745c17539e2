#!/usr/bin/env python3
"""Generate the golden fixtures by running the REFERENCE (mhpi/hydrodl2).

Runs only in the authoring container, where the reference checkout is mounted
read-only at /root/reference.  Nothing of the reference travels: this script
imports it, feeds it the bit-reproducible synthetic inputs of tests/synth.py
and stores *outputs and gradients only* in tests/golden/<case>.npz.

    python tests/golden/make_golden.py            # all cases
    python tests/golden/make_golden.py hbv_ties   # one case

Import recipe: SURVEY.md §9.6 (the VCS-generated hydrodl2._version module is
absent from the tree and is pre-seeded; the licence prompt skips itself because
stdin is not a tty).
"""
from __future__ import annotations

import os
import sys
import types
import warnings

import numpy as np
import torch

REF = "/root/reference/src"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests import golden_cases as gc  # noqa: E402


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference not present: fixtures can only be generated in the "
                         "authoring container")
    sys.path.insert(0, REF)
    v = types.ModuleType("hydrodl2._version")
    v.__version__ = "1.2.0"
    sys.modules["hydrodl2._version"] = v
    import hydrodl2  # noqa: F401
    return hydrodl2


def _run_case(hydrodl2, name: str) -> dict:
    spec = gc.CASES[name]
    cls = hydrodl2.load_model(spec["model"].lower(), spec["model"])
    dev = torch.device("cpu")
    cfg = spec["config"]
    model = cls(None if cfg is None else dict(cfg), dev)
    inp = gc.build_inputs(name)
    rec = {"torch_version": np.array(torch.__version__)}

    x = torch.from_numpy(inp["x_phy"]).clone()
    if spec.get("x_grad"):
        x.requires_grad_(True)
    x_dict = {"x_phy": x}
    if "muwts" in inp:
        x_dict["muwts"] = torch.from_numpy(inp["muwts"])
    leaves = []
    if spec["model"] in ("Hbv_2", "Hbv_2_hourly"):
        pd = torch.from_numpy(inp["p_dyn"]).clone().requires_grad_(True)
        ps = torch.from_numpy(inp["p_sta"]).clone().requires_grad_(True)
        x_dict["ac_all"] = torch.from_numpy(inp["ac_all"])
        x_dict["elev_all"] = torch.from_numpy(inp["elev_all"])
        params = (pd, ps)
        leaves = [("p_dyn", pd), ("p_sta", ps)]
        if spec["model"] == "Hbv_2_hourly":
            pr = torch.from_numpy(inp["p_distr"]).clone().requires_grad_(True)
            x_dict["outlet_topo"] = torch.from_numpy(inp["outlet_topo"])
            x_dict["areas"] = torch.from_numpy(inp["areas"])
            params = (pd, ps, pr)
            leaves.append(("p_distr", pr))
    else:
        p = torch.from_numpy(inp["parameters"]).clone().requires_grad_(True)
        params = p
        leaves = [("parameters", p)]
    if spec.get("x_grad"):
        leaves.append(("x_phy", x))

    if "torch_seed" in spec:
        torch.manual_seed(spec["torch_seed"])

    if spec.get("two_call"):
        # cache_states continuation: first half, then second half, same static row.
        T = spec["T"]
        h = T // 2
        with torch.no_grad():
            # the static parameters are read from the LAST row of what is passed
            # in, so pass the same last row to both calls.
            p1 = torch.cat([p[:h - 1], p[-1:]], 0)
            out1 = model({"x_phy": x[:h]}, p1)
            st1 = [s.clone() for s in model.get_states()]
            out2 = model({"x_phy": x[h:]}, p[h:])
            st2 = [s.clone() for s in model.get_states()]
        for k, v in out1.items():
            rec[f"out1/{k}"] = v.numpy()
        for k, v in out2.items():
            rec[f"out2/{k}"] = v.numpy()
        rec["states1"] = torch.stack(st1).numpy()
        rec["states2"] = torch.stack(st2).numpy()
        return rec

    keys = gc.loss_keys(name)
    if keys:
        out = model(x_dict, params)
    else:
        with torch.no_grad():
            out = model(x_dict, params)
    for k, v in out.items():
        rec[f"out/{k}"] = v.detach().numpy().copy()
    st = model.get_states()
    if spec["model"] == "Hbv_2_hourly":
        st = model._state_cache  # get_states() reads a differently spelled attribute
    if spec["model"] in ("Hbv_2", "Hbv_2_hourly"):
        # full state series [5,T,B,M]
        rec["states"] = torch.stack([s.detach() for s in st]).numpy()
    else:
        rec["states"] = torch.stack(list(st)).numpy()
    if keys:
        loss = 0.0
        for k in keys:
            w = torch.from_numpy(gc.loss_weight(name, k, out[k].shape))
            loss = loss + (w * out[k]).sum()
        loss.backward()
        rec["loss"] = np.array(float(loss))
        for lname, leaf in leaves:
            g = leaf.grad if leaf.grad is not None else torch.zeros_like(leaf)
            rec[f"grad/{lname}"] = g.numpy().copy()
    return rec


def _uh_fixture(hydrodl2) -> dict:
    """uh_gamma / uh_conv alone, including T' < 15 (SURVEY.md §8c case 8)."""
    from hydrodl2.core.calc import uh_conv, uh_gamma
    from tests import synth
    rec = {}
    for tag, T, B in (("long", 50, 7), ("short", 8, 5)):
        a = torch.from_numpy(synth.uniform((B,), 31, 1) * np.float32(2.9)).requires_grad_(True)
        b = torch.from_numpy(synth.uniform((B,), 31, 2) * np.float32(6.5)).requires_grad_(True)
        xq = torch.from_numpy(synth.uniform((T, B), 31, 3) * np.float32(5.0)).requires_grad_(True)
        UH = uh_gamma(a.repeat(T, 1).unsqueeze(-1), b.repeat(T, 1).unsqueeze(-1), lenF=15)
        rf = xq.unsqueeze(-1).permute([1, 2, 0])
        y = uh_conv(rf, UH.permute([1, 2, 0])).permute([2, 0, 1])  # [T,B,1]
        w = torch.from_numpy(synth.loss_weights((T, B, 1), 31, 4))
        (w * y).sum().backward()
        rec[f"{tag}/UH"] = UH.detach().numpy()[:, :, 0]
        rec[f"{tag}/y"] = y.detach().numpy()[:, :, 0]
        rec[f"{tag}/ga"] = a.grad.numpy().copy()
        rec[f"{tag}/gb"] = b.grad.numpy().copy()
        rec[f"{tag}/gx"] = xq.grad.numpy().copy()
    return rec


def main(argv):
    warnings.filterwarnings("ignore")
    torch.set_num_threads(4)
    hydrodl2 = _import_reference()
    outdir = os.path.dirname(os.path.abspath(__file__))
    names = argv or (list(gc.CASES) + ["uh_routing"])
    for name in names:
        rec = _uh_fixture(hydrodl2) if name == "uh_routing" else _run_case(hydrodl2, name)
        path = os.path.join(outdir, f"{name}.npz")
        np.savez_compressed(path, **rec)
        print(f"{name:24s} {os.path.getsize(path) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main(sys.argv[1:])
