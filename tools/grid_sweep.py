#!/usr/bin/env python3
"""Which kernel family wins at which grid size?  For 16-member ensembles and T = 730, wavefronts of
state W = 256 .. 4096 (B = 4 W basins), `hbv` with the delta-MG dynamic set and `hbv_2` with its three:
forward / adjoint ms with (a) the library's own dispatch, (b) the streaming kernels forced in both
directions (packed trajectory), (c) the streaming kernels switched off (pipelined / tiled forward,
time-parallel adjoint).  One JSON line per (model, W, T); profiles/r0x_grid_sweep.jsonl is this output.

    python tools/grid_sweep.py > gpurun_out/grid_sweep.jsonl
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import hydrodl2_amd  # noqa: E402
from hydrodl2_amd import ops  # noqa: E402

MODES = {"auto": {}, "stream": {"HBVX_STREAM_MIN": "1"}, "nostream": {"HBVX_STREAM": "0"}}


def one(fam, B, T=730, M=16, steps=4):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    day = torch.arange(T, device=dev, dtype=torch.float32)[:, None]
    season = torch.sin(2 * torch.pi * day / 365.0)
    x = torch.stack([torch.clamp((torch.rand((T, B), generator=g, device=dev) - 0.7) * 60.0, min=0.0),
                     10 * season + 5 * torch.randn((T, B), generator=g, device=dev),
                     torch.clamp(3 + 2.5 * season, min=0).expand(T, B)], -1).contiguous()
    xd = {"x_phy": x}
    if fam == "hbv":
        model = hydrodl2_amd.load_model("hbv", "Hbv")({"nmul": M, "dynamic_params": {"Hbv": ["parBETA", "parBETAET"]}}, dev)
        p = torch.randn((T, B, model.learnable_param_count), generator=g, device=dev).requires_grad_(True)
        params, leaves = p, [p]
    else:
        model = hydrodl2_amd.load_model("hbv_2", "Hbv_2")({"nmul": M, "dynamic_params": {"Hbv_2": ["parBETA", "parK0", "parBETAET"]}}, dev)
        pd = torch.rand((T, B, 3 * M), generator=g, device=dev).requires_grad_(True)
        ps = torch.rand((B, 13 * M), generator=g, device=dev).requires_grad_(True)
        params, leaves = (pd, ps), [pd, ps]
        xd["ac_all"] = torch.rand(B, generator=g, device=dev) * 5000
        xd["elev_all"] = torch.rand(B, generator=g, device=dev) * 3000
    w = torch.randn((T, B, 1), generator=g, device=dev)
    res = {"model": fam, "B": B, "W": (B + 3) // 4, "T": T}
    for mode, env in MODES.items():
        for k in ("HBVX_STREAM_MIN", "HBVX_STREAM"):
            os.environ.pop(k, None)
        os.environ.update(env)

        def step():
            for l in leaves:
                l.grad = None
            (model(xd, params)["streamflow"] * w).sum().backward()
        step()
        ops.KERNEL_EVENTS = []
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
        kt = {}
        for nm, e0, e1 in ev:
            kt[nm] = kt.get(nm, 0.0) + e0.elapsed_time(e1) / steps
        res[mode] = {"fwd": round(kt.get("hbvx_forward", 0), 4), "bwd": round(kt.get("hbvx_backward", 0), 4)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    # python tools/grid_sweep.py [T ...]   (default 730; at T = 7300 the raw [T,B,ny] tensor and its gradient bound
    # the sweep: W <= 1536, 35 GB each)
    for T in ([int(a) for a in sys.argv[1:]] or [730]):
        Ws = (128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144) if T <= 1000 else (128, 256, 512, 768, 1024, 1536)
        for fam in ("hbv", "hbv_2"):
            for W in Ws:
                one(fam, 4 * W, T=T, steps=4 if T <= 1000 else 2)
                torch.cuda.empty_cache()
