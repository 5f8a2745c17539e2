"""PyTorch CPU *eager* restatement of the reference's HBV 1.0 hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/ and bench.py's `cpu_baseline` leg may import this file; the product (`hydrodl2_amd/`)
never does.  It exists for one purpose the C oracle cannot serve: to time, on the GPU box's host
cores, the *kind* of computation the reference actually runs -- one small ATen kernel per operator
per day plus the autograd tape -- since the reference itself cannot travel to that box.

What it follows (file:line in /root/reference/src/hydrodl2):
  * parameter prep  models/hbv/hbv.py:182-256 (sigmoid, [T,B,n,M] view, static = last row repeated
    over T, dynamic = per-day rows, `p*(hi-lo)+lo` core/calc/utils.py:24) -- the materialised
    [T,B,M] tensors and the per-day row selects are kept on purpose: their `select_backward`
    zero-fills are what makes the reference's backward O(T^2);
  * daily step      models/hbv/hbv.py:428-505 in the association order of SURVEY.md §9.3;
  * ensemble mean   models/hbv/hbv.py:507-511;  routing  core/calc/uh_routing.py:5-57;
  * BFI / outputs   models/hbv/hbv.py:555-596.
Pinned against the reference-generated fixtures in tests/test_eager_restatement.py.
No warm-up / dy_drop / muwts / state caching: the timed configuration (static or dynamic
parameters, warm_up = 0) does not use them and the C oracle covers them.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BOUNDS = [("parBETA", 1.0, 6.0), ("parFC", 50, 1000), ("parK0", 0.05, 0.9), ("parK1", 0.01, 0.5),
          ("parK2", 0.001, 0.2), ("parLP", 0.2, 1), ("parPERC", 0, 10), ("parUZL", 0, 100),
          ("parTT", -2.5, 2.5), ("parCFMAX", 0.5, 10), ("parCFR", 0, 0.1), ("parCWH", 0, 0.2)]
BETAET = ("parBETAET", 0.3, 5)
ROUTE = [("route_a", 0, 2.9), ("route_b", 0, 6.5)]
SERIES = ["Qsim", "Q0", "Q1", "Q2", "AET", "SWE", "recharge", "excs", "evapfactor", "tosoil", "PERC"]


def _rescale(u, lo, hi):
    return u * (hi - lo) + lo


def gamma_uh(a, b, length):
    """Normalised gamma-pdf unit hydrograph [length, B] from route_a / route_b [B]."""
    shape = torch.relu(a) + 0.1
    scale = torch.relu(b) + 0.5
    tk = torch.arange(0.5, length * 1.0, dtype=a.dtype).unsqueeze(-1)          # [L,1]
    dens = tk ** (shape - 1) * torch.exp(-tk / scale) / (shape.lgamma().exp() * scale ** shape)
    return dens / dens.sum(0)


def route(q, uh):
    """Causal per-basin FIR, zero history: q [T,B] with uh [L,B] -> [T,B] (grouped conv1d)."""
    T, B = q.shape
    L = uh.shape[0]
    kern = torch.flip(uh.t().unsqueeze(1), [2])                               # [B,1,L]
    y = F.conv1d(q.t().unsqueeze(0), kern, groups=B, padding=L - 1)
    return y[0, :, :T].t()


def hbv_eager(x_phy, parameters, nmul, dynamic=(), nearzero=1e-5, routing=True):
    """x_phy [T,B,3] (prcp, tmean, pet), parameters raw [T,B,ny] -> dict of [T,B,1] series + BFI."""
    T, B, _ = x_phy.shape
    M = nmul
    table = list(BOUNDS) + ([BETAET] if "parBETAET" in dynamic else [])
    n = len(table)
    unit = torch.sigmoid(parameters[:, :, : n * M]).view(T, B, n, M)
    par = {}
    for i, (name, lo, hi) in enumerate(table):
        if name in dynamic:
            par[name] = _rescale(unit[:, :, i, :], lo, hi)
        else:
            par[name] = _rescale(unit[-1, :, i, :].unsqueeze(0).repeat([T, 1, 1]), lo, hi)
    P = x_phy[:, :, 0].unsqueeze(-1).repeat(1, 1, M)
    Tm = x_phy[:, :, 1].unsqueeze(-1).repeat(1, 1, M)
    PET = x_phy[:, :, 2].unsqueeze(-1).repeat(1, 1, M)

    SP = torch.full((B, M), 0.001)
    MW, SM, SUZ, SLZ = SP.clone(), SP.clone(), SP.clone(), SP.clone()
    out = {k: torch.zeros(T, B, M) for k in SERIES}
    for t in range(T):
        p = {k: v[t] for k, v in par.items()}
        # snow
        rain = P[t] * (Tm[t] >= p["parTT"]).float()
        snow = P[t] * (Tm[t] < p["parTT"]).float()
        SP = SP + snow
        melt = torch.min(torch.clamp(p["parCFMAX"] * (Tm[t] - p["parTT"]), min=0.0), SP)
        MW = MW + melt
        SP = SP - melt
        refr = torch.min(torch.clamp(p["parCFR"] * p["parCFMAX"] * (p["parTT"] - Tm[t]), min=0.0), MW)
        SP = SP + refr
        MW = MW - refr
        tosoil = torch.clamp(MW - p["parCWH"] * SP, min=0.0)
        MW = MW - tosoil
        # soil
        wet = torch.clamp((SM / p["parFC"]) ** p["parBETA"], min=0.0, max=1.0)
        rech = (rain + tosoil) * wet
        SM = SM + rain + tosoil - rech
        exc = torch.clamp(SM - p["parFC"], min=0.0)
        SM = SM - exc
        ef = SM / (p["parLP"] * p["parFC"])
        if "parBETAET" in p:
            ef = ef ** p["parBETAET"]
        ef = torch.clamp(ef, min=0.0, max=1.0)
        et = torch.min(SM, PET[t] * ef)
        SM = torch.clamp(SM - et, min=nearzero)
        # groundwater
        SUZ = SUZ + rech + exc
        perc = torch.min(SUZ, p["parPERC"])
        SUZ = SUZ - perc
        q0 = p["parK0"] * torch.clamp(SUZ - p["parUZL"], min=0.0)
        SUZ = SUZ - q0
        q1 = p["parK1"] * SUZ
        SUZ = SUZ - q1
        SLZ = SLZ + perc
        q2 = p["parK2"] * SLZ
        SLZ = SLZ - q2
        for k, v in zip(SERIES, (q0 + q1 + q2, q0, q1, q2, et, SP, rech, exc, ef, tosoil, perc)):
            out[k][t] = v
    mean = {k: v.mean(-1) for k, v in out.items()}
    if routing:
        ra = _rescale(torch.sigmoid(parameters[-1, :, n * M]), ROUTE[0][1], ROUTE[0][2])
        rb = _rescale(torch.sigmoid(parameters[-1, :, n * M + 1]), ROUTE[1][1], ROUTE[1][2])
        uh = gamma_uh(ra, rb, min(T, 15))
        routed = [route(mean[k], uh) for k in ("Qsim", "Q0", "Q1", "Q2")]
    else:
        routed = [mean[k] for k in ("Qsim", "Q0", "Q1", "Q2")]
    col = lambda v: v.unsqueeze(-1)  # noqa: E731
    res = {"streamflow": col(routed[0]), "srflow": col(routed[1]), "ssflow": col(routed[2]),
           "gwflow": col(routed[3]), "AET_hydro": col(mean["AET"]), "PET_hydro": col(PET.mean(-1)),
           "SWE": col(mean["SWE"]), "streamflow_no_rout": col(mean["Qsim"]),
           "srflow_no_rout": col(mean["Q0"]), "ssflow_no_rout": col(mean["Q1"]),
           "gwflow_no_rout": col(mean["Q2"]), "recharge": col(mean["recharge"]), "excs": col(mean["excs"]),
           "evapfactor": col(mean["evapfactor"]), "tosoil": col(mean["tosoil"]),
           "percolation": col(mean["PERC"])}
    res["BFI"] = 100 * routed[3].sum(0) / (routed[0].sum(0) + nearzero)
    return res
