#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

    python bench.py --gpus N --steps K --warmup W

A "step" is one forward + backward pass of the HBV hot path over one batch of
synthetic CAMELS-shaped input (configs[1]: hbv, 671 basins x 16 members x 7300
days, static parameters) through the drop-in module (`hydrodl2_amd.load_model
('hbv')`): raw NN output [T,B,ny] -> flux dictionary -> a linear loss on
`streamflow` (a fixed N(0,1) tensor stands in for the NSE-loss gradient) ->
gradient w.r.t. the raw NN output.  Inputs are resident in HBM before the timed
region.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank owns its
own 671-basin shard (weak scaling; basins are independent); the
only collective is an RCCL all-reduce of the loss and of the basin-summed
gradient row that a shared parameterisation network would receive.

Rank 0 prints ONE JSON line.  `value` = whole-job basin-ensemble-timesteps/s.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synth_inputs(T, B, ny, dev, seed):
    """CAMELS-shaped forcings and raw N(0,1) parameters, generated on the GPU."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    day = torch.arange(T, device=dev, dtype=torch.float32)[:, None]
    season = torch.sin(2 * torch.pi * day / 365.0)
    u = torch.rand((T, B), generator=g, device=dev)
    P = torch.clamp((u - 0.7) * 60.0, min=0.0)
    boff = torch.rand((1, B), generator=g, device=dev) * 25.0 - 10.0
    Tm = 10.0 * season + 5.0 * torch.randn((T, B), generator=g, device=dev) + boff
    PET = torch.clamp(3.0 + 2.5 * season + 0.3 * torch.randn((T, B), generator=g, device=dev),
                      min=0.0)
    x = torch.stack([P, Tm, PET], dim=-1).contiguous()
    params = torch.randn((T, B, ny), generator=g, device=dev)
    w = torch.randn((T, B, 1), generator=g, device=dev)
    return x, params, w


def cpu_baseline(B, M, T_sample, seed=0):
    """The CPU oracle (oracle/, OpenMP over basins) on a bounded sample of the same workload,
    called through the same C ABI; reported next to the GPU number, never the target."""
    import ctypes as C
    import numpy as np
    import __graft_entry__ as ge
    from hydrodl2_amd import _abi
    if not os.path.exists(ge.ORACLE_LIB):
        ge.build_oracle()
    lib = _abi.Library(ge.ORACLE_LIB)
    lib.dll.hbvo_num_threads.restype = C.c_int
    threads = int(lib.dll.hbvo_num_threads())
    rng = np.random.default_rng(seed)
    T, n = T_sample, 12
    ny = n * M + 2
    day = np.arange(T, dtype=np.float32)[:, None]
    season = np.sin(2 * np.pi * day / 365.0).astype(np.float32)
    x = np.stack([np.maximum((rng.random((T, B), dtype=np.float32) - 0.7) * 60.0, 0),
                  10 * season + 5 * rng.standard_normal((T, B), dtype=np.float32),
                  np.maximum(3 + 2.5 * season + 0 * rng.random((T, B), dtype=np.float32), 0)],
                 -1).astype(np.float32)
    row = rng.standard_normal((1, B, ny), dtype=np.float32)  # static: one live row
    N = B * M
    flux = np.empty((11, T, B), np.float32)
    st = np.empty((5, B, M), np.float32)
    traj = np.empty((5, T + 1, N), np.float32)
    aux = np.empty((2, T, N), np.float32)
    gflux = np.zeros((11, T, B), np.float32)
    gflux[0] = rng.standard_normal((T, B), dtype=np.float32)
    grow = np.zeros_like(row)

    bounds = [[1, 6], [50, 1000], [.05, .9], [.01, .5], [.001, .2], [.2, 1], [0, 10], [0, 100],
              [-2.5, 2.5], [.5, 10], [0, .1], [0, .2]]
    d = _abi.Desc()
    d.abi_version, d.model, d.T, d.B, d.M, d.n_param = _abi.ABI_VERSION, _abi.MODEL_HBV10, T, B, M, n
    d.raw_sigmoid, d.ch_prcp, d.ch_tmean, d.ch_pet, d.nearzero = 1, 0, 1, 2, 1e-5
    d.x, d.x_t_stride, d.x_b_stride = x.ctypes.data, B * 3, 3
    io = _abi.BwdIO()
    for i in range(n):
        d.p[i].sta = row.ctypes.data + 4 * i * M
        d.p[i].sta_b_stride = ny
        d.p[i].lo, d.p[i].hi = bounds[i]
        io.g[i].sta = grow.ctypes.data + 4 * i * M
        io.g[i].sta_b_stride = ny
    out = _abi.FwdOut()
    out.flux, out.state_out, out.traj, out.aux, out.n_flux = (
        flux.ctypes.data, st.ctypes.data, traj.ctypes.data, aux.ctypes.data, 11)
    io.traj, io.aux, io.grad_flux, io.n_flux = traj.ctypes.data, aux.ctypes.data, gflux.ctypes.data, 11
    lib.forward(d, out, 0)  # warm-up (page faults)
    reps, dt = 0, 0.0
    while reps < 3 or (dt < 2.0 and reps < 20):   # a few passes: one pass is well under a second on a big host
        t0 = time.perf_counter()
        lib.forward(d, out, 0)
        lib.backward(d, io, 0)
        dt += time.perf_counter() - t0
        reps += 1
    return {"value": reps * B * M * T / dt, "unit": "basin-ensemble-timesteps/s", "cores": threads,
            "kind": "port",
            "sample": f"oracle (C, OpenMP over basins) fwd+bwd recurrence, {reps} passes over {B}x{M}x{T_sample} "
                      f"lane-steps, {dt:.2f} s wall on {threads} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--basins", type=int, default=671)
    ap.add_argument("--nmul", type=int, default=16)
    ap.add_argument("--days", type=int, default=7300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-days", type=int, default=1825)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one rank per GPU)")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import hydrodl2_amd
    from hydrodl2_amd import ops, sharding
    B, M, T = args.basins, args.nmul, args.days
    Hbv = hydrodl2_amd.load_model("hbv", "Hbv")
    model = Hbv({"nmul": M, "dynamic_params": {"Hbv": []}}, dev)
    ny = model.learnable_param_count
    x, params, w = synth_inputs(T, B, ny, dev, seed=1000 + rank)
    params.requires_grad_(True)
    bucket = torch.zeros(ny + 1, device=dev)

    def step():
        params.grad = None
        out = model({"x_phy": x}, params)
        loss = (out["streamflow"] * w).sum()
        loss.backward()
        # what a shared parameterisation network would receive: basin-summed last-row gradient
        bucket[:ny] = params.grad[-1].sum(0)
        bucket[ny] = loss.detach()
        sharding.all_reduce_sum_([bucket])   # the path's only collective (RCCL over xGMI)
        return loss

    for _ in range(args.warmup):
        step()
    ops.KERNEL_EVENTS = []          # per-launch HIP events on the launch stream
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    if world > 1:
        tmax = torch.tensor([dt], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    lane_steps = B * M * T
    value = world * lane_steps * args.steps / dt

    # per-kernel durations (ms) from the events recorded around each ABI call
    kt = {}
    for name, e0, e1 in events:
        kt.setdefault(name, []).append(e0.elapsed_time(e1))
    kavg = {k: sum(v) / len(v) for k, v in kt.items()}

    # Algorithmic HBM bytes per launch ("bytes per lane-step", fp32):
    #   forward : forcings 12/M + 11 mean series 4*11/M + saved trajectory/aux 28   = 31.5  B
    #   adjoint : time-parallel, two passes over (forcings 12/M + trajectory/aux 28 + routed-Q
    #             gradients 4*4/M = 29.75 B) + per-chunk maps/partials (35+12 floats per lane and
    #             64-day chunk, written and read: 5.9 B)                               = 65.4  B
    n_flux = 11
    bytes_fwd = lane_steps * (12.0 / M + 4.0 * n_flux / M + 28.0)
    bytes_bwd = lane_steps * (2.0 * (12.0 / M + 28.0 + 4.0 * 4 / M) + 2.0 * 4.0 * (35 + 12) / 64.0)
    # Dominant kernel: k_fwd_pipe, the one kernel behind hbvx_forward (rocprofv3 --stats,
    # profiles/r01_kernel_stats.csv: the largest single kernel; hbvx_backward is four kernels, the two
    # big ones ~0.65 ms each).  The adjoint call is reported beside it in `calls`.
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    pmc = {}
    if os.path.exists(tj):
        try:
            pmc = json.load(open(tj))
        except Exception:
            pmc = {}
    calls = {}
    bytes_zero = 4.0 * T * B * ny      # the dense [T,B,ny] gradient the autograd contract returns (hbvx_zero)
    for call, nbytes in (("hbvx_forward", bytes_fwd), ("hbvx_backward", bytes_bwd), ("hbvx_zero", bytes_zero)):
        if call in kavg:
            calls[call] = {"avg_ms": round(kavg[call], 4), "algorithmic_bytes": nbytes,
                           "achieved_GBps": round(nbytes / (kavg[call] * 1e-3) / 1e9, 2),
                           "traffic": pmc.get(call, {}).get("hbm_bytes_raw")}
    dom = "hbvx_forward"
    achieved = bytes_fwd / (kavg[dom] * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": "k_fwd_pipe (hbv_pipe.h), the kernel of hbvx_forward",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": calls[dom]["traffic"],
                "avg_ms": round(kavg[dom], 4), "calls": calls,
                "kernel_ms": {k: round(v, 4) for k, v in kavg.items()}}

    if rank == 0:
        res = {
            "metric": "basin-ensemble-timesteps/sec fwd+bwd", "value": value,
            "unit": "basin-ensemble-timesteps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"hbv (HBV 1.0) {B} basins x {M} members x {T} days, static "
                                   "parameters, fwd+bwd, raw parameters [T,B,ny] in HBM",
                       "basins_per_gpu": B, "nmul": M, "days": T, "parallelism": f"basin-shard x{world}"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(B, M, min(args.cpu_sample_days, T))
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
