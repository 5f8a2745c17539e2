"""Multi-timescale golden cases (Hbv_2_mts): inputs, configs and the runner shared by
tests/golden/make_golden_mts.py (reference side) and tests/test_mts.py (this package)."""
from __future__ import annotations

import numpy as np
import torch

from . import synth

DYN = ["parBETA", "parK0", "parBETAET"]
M = 4
N_LOW, N_HIGH = 16, 19

CASES = {
    # all units fit one block, training mode: only the unit runoff 'Qs' is produced
    # (hbv_2_mts.py:193-195 switches the gage routing off).
    "mts_train": dict(T_low=40, T_high=96, B=7, G=3, seed=41, simulate=False,
                      chunks=dict(train_spatial_chunk_size=16, simulate_spatial_chunk_size=3,
                                  simulate_temporal_chunk_size=40, train_warmup=24)),
    # simulate mode: 3 spatial blocks of <= 3 units, routing in temporal chunks of 40 h with a
    # 24 h overlap.  On the reference this path needs the alias described in make_golden_mts.py.
    "mts_chunked": dict(T_low=30, T_high=120, B=7, G=3, seed=42, simulate=True,
                        chunks=dict(train_spatial_chunk_size=16, simulate_spatial_chunk_size=3,
                                    simulate_temporal_chunk_size=40, train_warmup=24)),
}


def configs(name: str):
    spec = CASES[name]
    low = {"nmul": M, "dynamic_params": {"Hbv_2": list(DYN)}, "cache_states": True}
    high = {"nmul": M, "dynamic_params": {"Hbv_2_hourly": list(DYN)}}
    high.update(spec["chunks"])
    return low, high


def build_inputs(name: str) -> dict:
    spec = CASES[name]
    Tl, Th, B, G, seed = spec["T_low"], spec["T_high"], spec["B"], spec["G"], spec["seed"]
    out = {}
    out["x_low"] = synth.forcing(Tl, B, seed)
    out["x_high"] = (synth.forcing(Th, B, seed + 1000)
                     * np.array([1.0 / 8.0, 1.0, 1.0 / 24.0], np.float32))
    out["low_dyn"] = synth.unit_parameters((Tl, B, len(DYN) * M), seed, 4)
    out["low_sta"] = synth.unit_parameters((B, (N_LOW - len(DYN)) * M), seed, 6)
    out["high_dyn"] = synth.unit_parameters((Th, B, len(DYN) * M), seed, 16)
    out["high_sta"] = synth.unit_parameters((B, (N_HIGH - len(DYN)) * M), seed, 17)
    out["ac_all"] = (synth.uniform((B,), seed, 7) * np.float32(5000.0)).astype(np.float32)
    out["elev_all"] = (synth.uniform((B,), seed, 8) * np.float32(3000.0)).astype(np.float32)
    topo = (synth.uniform((G, B), seed, 13) < np.float32(0.45)).astype(np.float32)
    topo[np.arange(B) % G, np.arange(B)] = 1.0
    out["outlet_topo"] = topo
    out["areas"] = (synth.uniform((B,), seed, 14) * np.float32(90.0) + np.float32(5.0)).astype(np.float32)
    out["p_distr"] = synth.unit_parameters((int(topo.sum()), 3), seed, 15)
    return out


LEAVES = ["low_dyn", "low_sta", "high_dyn", "high_sta", "p_distr"]


def run(model, name: str, dev) -> dict:
    """Forward + backward of case `name` through `model` (reference or this package)."""
    spec = CASES[name]
    inp = build_inputs(name)
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    for k in LEAVES:
        t[k].requires_grad_(True)
    model.set_mode(spec["simulate"])
    x_dict = {"x_phy_low_freq": t["x_low"], "x_phy_high_freq": t["x_high"], "ac_all": t["ac_all"],
              "elev_all": t["elev_all"], "outlet_topo": t["outlet_topo"], "areas": t["areas"]}
    params = ([t["low_dyn"], t["low_sta"]], [t["high_dyn"], t["high_sta"], t["p_distr"]])
    out = model(x_dict, params)
    res = {f"out/{k}": v.detach().cpu().numpy().copy() for k, v in out.items()}
    loss = 0.0
    for i, (k, v) in enumerate(sorted(out.items())):
        w = torch.from_numpy(synth.loss_weights(tuple(v.shape), spec["seed"], 50 + i)).to(dev)
        loss = loss + (w * v).sum()
    loss.backward()
    for k in LEAVES:
        g = t[k].grad if t[k].grad is not None else torch.zeros_like(t[k])
        res[f"grad/{k}"] = g.detach().cpu().numpy().copy()
    return res
