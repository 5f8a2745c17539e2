#!/usr/bin/env python3
"""Secondary measurements (not the headline): BASELINE configs 3, 4 and 5-per-GPU-share, kernel
times from per-launch HIP events and achieved algorithmic HBM bandwidth.

    python tools/bench_configs.py [cfg2|cfg2dyn|cfg3|cfg4|cfg5|dmg|hourly] ...
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import hydrodl2_amd  # noqa: E402
from hydrodl2_amd import ops  # noqa: E402


def gen(T, B, dev, seed=0):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    day = torch.arange(T, device=dev, dtype=torch.float32)[:, None]
    season = torch.sin(2 * torch.pi * day / 365.0)
    P = torch.clamp((torch.rand((T, B), generator=g, device=dev) - 0.7) * 60.0, min=0.0)
    Tm = 10 * season + 5 * torch.randn((T, B), generator=g, device=dev) \
        + torch.rand((1, B), generator=g, device=dev) * 25 - 10
    PET = torch.clamp(3 + 2.5 * season, min=0).expand(T, B)
    return torch.stack([P, Tm, PET], -1).contiguous(), g


def run(name, steps=5, warmup=2):
    dev = torch.device("cuda:0")
    if name in ("cfg2", "cfg2dyn", "cfg2dynK", "dmg", "dmggraph"):
        T, B, M = (730, 100, 16) if name.startswith("dmg") else (7300, 671, 16)
        dyn = [] if name == "cfg2" else (["parK0", "parK1"] if name == "cfg2dynK" else ["parBETA", "parBETAET"])
        cfgd = {"nmul": M, "dynamic_params": {"Hbv": dyn}}
        if name.startswith("dmg"):
            cfgd["warm_up"] = 365
        if name == "dmggraph":
            cfgd["graph"] = True
        model = hydrodl2_amd.load_model("hbv", "Hbv")(cfgd, dev)
        n_dyn, nf = len(dyn), 11
    elif name == "cfg3":
        T, B, M = 7300, 671, 16
        H = hydrodl2_amd.load_model("hbv_1_1p", "Hbv_1_1p")
        names = list(H(None, dev).parameter_bounds)
        model = H({"nmul": M, "dynamic_params": {"Hbv_1_1p": names}}, dev)
        n_dyn, nf = 14, 12
    elif name == "cfg5":
        T, B, M = 730, 12500, 16   # one GPU's share of 100k basins over 8 GPUs
        dyn = ["parBETA", "parK0", "parBETAET"]
        model = hydrodl2_amd.load_model("hbv_2", "Hbv_2")({"nmul": M, "dynamic_params": {"Hbv_2": dyn}}, dev)
        n_dyn, nf = 3, 12
    elif name == "cfg4":
        T, B, M = 7300, 671, 16
        extra = {}
        if os.environ.get("HBVX_CFG4_MAXITER"):      # timing floors only (0: exactly one update per day)
            extra["newton_max_iter"] = int(os.environ["HBVX_CFG4_MAXITER"])
        model = hydrodl2_amd.load_model("hbv_adj", "HbvAdj")(
            {"nmul": M, "dynamic_params": {"HbvAdj": ["parBETAET"]}, **extra}, dev)
        n_dyn, nf = 1, 1
    elif name.startswith("grid:"):
        # grid:<hbv|hbv_2>:<B>:<T>[:<M>]  -- M members (16), 3 dynamic parameters for hbv_2, 2 for hbv
        _, fam, Bs, Ts, *rest = name.split(":")
        T, B, M = int(Ts), int(Bs), int(rest[0]) if rest else 16
        if fam == "hbv":
            dyn = ["parBETA", "parBETAET"]
            model = hydrodl2_amd.load_model("hbv", "Hbv")({"nmul": M, "dynamic_params": {"Hbv": dyn}}, dev)
            n_dyn, nf = 2, 11
        else:
            dyn = ["parBETA", "parK0", "parBETAET"]
            model = hydrodl2_amd.load_model("hbv_2", "Hbv_2")({"nmul": M, "dynamic_params": {"Hbv_2": dyn}}, dev)
            n_dyn, nf = 3, 12
    elif name == "hourly":
        T, B, M = 2160, 4000, 4    # 90 days of hours, 4000 units draining to 100 gages
        dyn = ["parBETA", "parK0", "parBETAET"]
        model = hydrodl2_amd.load_model("hbv_2_hourly", "Hbv_2_hourly")(
            {"nmul": M, "dynamic_params": {"Hbv_2_hourly": dyn}}, dev)
        n_dyn, nf = 3, 12
    else:
        raise SystemExit(name)
    x, g = gen(T, B, dev)
    xd = {"x_phy": x}
    if name == "hourly":
        G = 100
        x = x * torch.tensor([1 / 8.0, 1.0, 1 / 24.0], device=dev)
        xd["x_phy"] = x
        topo = (torch.rand((G, B), generator=g, device=dev) < 0.02).float()
        topo[torch.arange(B, device=dev) % G, torch.arange(B, device=dev)] = 1.0
        pd = torch.rand((T, B, 3 * M), generator=g, device=dev).requires_grad_(True)
        ps = torch.rand((B, 16 * M), generator=g, device=dev).requires_grad_(True)
        pr = torch.rand((int(topo.sum()), 3), generator=g, device=dev).requires_grad_(True)
        params = (pd, ps, pr)
        xd["ac_all"] = torch.rand(B, generator=g, device=dev) * 5000
        xd["elev_all"] = torch.rand(B, generator=g, device=dev) * 3000
        xd["outlet_topo"] = topo
        xd["areas"] = torch.rand(B, generator=g, device=dev) * 90 + 5
        leaves = [pd, ps, pr]
    elif name == "cfg5" or name.startswith("grid:hbv_2"):
        pd = torch.rand((T, B, 3 * M), generator=g, device=dev).requires_grad_(True)
        ps = torch.rand((B, 13 * M), generator=g, device=dev).requires_grad_(True)
        params = (pd, ps)
        xd["ac_all"] = torch.rand(B, generator=g, device=dev) * 5000
        xd["elev_all"] = torch.rand(B, generator=g, device=dev) * 3000
        leaves = [pd, ps]
    else:
        p = torch.randn((T, B, model.learnable_param_count), generator=g, device=dev).requires_grad_(True)
        params, leaves = p, [p]
    Tp = T - (model.warm_up if getattr(model, "warm_up_states", True) else 0)
    w = torch.randn((Tp, 100 if name == "hourly" else B, 1), generator=g, device=dev)

    def step():
        for l in leaves:
            l.grad = None
        out = model(xd, params)
        key = "flow_sim" if name == "cfg4" else "streamflow"
        (out[key] * w).sum().backward()

    for _ in range(warmup):
        step()
    ops.KERNEL_EVENTS = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    kt = {}
    for nm, e0, e1 in ev:
        kt.setdefault(nm, []).append(e0.elapsed_time(e1))
    # per step there can be two forward launches (warm-up + main): sum per step
    kms = {k: sum(v) / steps for k, v in kt.items()}
    ls = B * M * T
    routed = 4 if model.routing else 0
    b_f = ls * (12 / M + 4 * n_dyn + 4 * nf / M + 20)
    b_b = ls * (12 / M + 8 * n_dyn + 20 + 4 * max(routed, 1) / M)
    res = {"config": name, "T": T, "B": B, "M": M, "n_dyn": n_dyn, "ms_per_step": round(dt * 1e3, 3),
           "lane_steps_per_s": ls / dt, "kernel_ms": {k: round(v, 4) for k, v in kms.items()}}
    if "hbvx_forward" in kms:
        res["fwd_GBps"] = round(b_f / (kms["hbvx_forward"] * 1e-3) / 1e9, 1)
        res["bwd_GBps"] = round(b_b / (kms["hbvx_backward"] * 1e-3) / 1e9, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["cfg2", "cfg2dyn", "cfg3", "cfg4", "cfg5", "dmg"]):
        run(n)
