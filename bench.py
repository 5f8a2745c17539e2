#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's configs.

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg5]

A "step" is one forward + backward pass of the HBV hot path over one batch of synthetic
CAMELS-shaped input through the drop-in module (`hydrodl2_amd.load_model`): raw NN output ->
flux dictionary -> a linear loss on `streamflow` (a fixed N(0,1) tensor stands in for the NSE-loss
gradient) -> gradient w.r.t. the raw NN output.  Inputs are resident in HBM before the timed region.

--config cfg2 (default, the headline; BASELINE.json configs[1]): `hbv`, 671 basins x 16 members x
    7300 days, static parameters.  N > 1: WEAK scaling, every rank owns its own 671-basin shard.
--config cfg5 (configs[4]): `hbv_2`, 100 000 basins x 16 members x 730 days, three dynamic
    parameters.  N > 1: STRONG scaling, rank r owns basins [r*ceil(B/N), ...).
Basins are independent, so the data path has no collective; the only exchange is one RCCL
all-reduce per step of the loss and of the basin-summed static-parameter gradient a shared
parameterisation network would receive.

N > 1 is one process per GPU.  Either the driver starts the ranks (torch.distributed.run: RANK /
LOCAL_RANK / WORLD_SIZE set) or `python bench.py --gpus N` does it itself: the parent starts N
children BEFORE touching the GPU and relays rank 0's JSON line.

Rank 0 prints ONE JSON line.  `value` = whole-job basin-ensemble-timesteps/s.  At N = 1 the line also
carries `secondary` (configs 2-dyn, 3, 4 -- staged and under the reference's joint Newton policy --, one GPU's share of 5,
config 5 at full size, the deltaMG minibatch shape, Hbv_2_hourly with gage routing, the sequence LSTM and one
examples/train_dpl.py step; driver-timed in the same run, each with what limits it: `limited_by`) and
`cpu_baseline` (the pure-torch eager restatement of the reference's path on the host cores -- `value` -- with the
C/OpenMP oracle port beside it); every timed entry carries its per-step samples (`step_samples`: n, median, min, max).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CKPT_K = 64             # checkpoint interval assumed by SURVEY.md §8d's algorithmic byte count
SEC_STEPS = 10          # timed steps of every secondary configuration (median of >= 10)


# --------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a launcher
# --------------------------------------------------------------------------------------------
def self_launch(n: int, argv: list[str]) -> int:
    """Start n ranks of this script (one per GPU) and relay rank 0's stdout; stderr of every rank is
    relayed with a rank prefix.  Runs before the parent has made any GPU call (it never makes one).

    All children are polled: the first one that exits non-zero ends the job -- the others (which would
    otherwise sit in the rendezvous or in a collective until its time-out) are terminated and that
    rank's exit code is returned within seconds."""
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, pumps, out0 = [], [], []

    def pump(stream, sink):
        for line in iter(stream.readline, b""):
            sink(line)
        stream.close()

    def relay_err(rank):
        def sink(line):
            sys.stderr.write(f"[rank {rank}] " + line.decode(errors="replace"))
            sys.stderr.flush()
        return sink

    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                             stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE)
        procs.append(p)
        pumps.append(threading.Thread(target=pump, args=(p.stderr, relay_err(r)), daemon=True))
        if r == 0:
            pumps.append(threading.Thread(target=pump, args=(p.stdout, out0.append), daemon=True))
    for t in pumps:
        t.start()
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = abs(code) or 1
                sys.stderr.write(f"[bench] rank {r} exited with {code}: stopping the other ranks\n")
                for q in sorted(live):
                    procs[q].terminate()
                deadline = time.time() + 10.0
                for q in sorted(live):
                    try:
                        procs[q].wait(max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        procs[q].kill()
                        procs[q].wait()
                live.clear()
                break
        if live:
            time.sleep(0.05)
    for t in pumps:
        t.join(timeout=5.0)
    sys.stdout.write(b"".join(out0).decode())
    sys.stdout.flush()
    return rc


# --------------------------------------------------------------------------------------------
# synthetic workloads
# --------------------------------------------------------------------------------------------
def synth_forcing(T, B, dev, g):
    """CAMELS-shaped forcings, generated on the compute device."""
    import torch
    day = torch.arange(T, device=dev, dtype=torch.float32)[:, None]
    season = torch.sin(2 * torch.pi * day / 365.0)
    u = torch.rand((T, B), generator=g, device=dev)
    P = torch.clamp((u - 0.7) * 60.0, min=0.0)
    boff = torch.rand((1, B), generator=g, device=dev) * 25.0 - 10.0
    Tm = 10.0 * season + 5.0 * torch.randn((T, B), generator=g, device=dev) + boff
    PET = torch.clamp(3.0 + 2.5 * season + 0.3 * torch.randn((T, B), generator=g, device=dev), min=0.0)
    return torch.stack([P, Tm, PET], dim=-1).contiguous()


WORKLOADS = {
    # name: (model file, class, T, B, M, dynamic parameters or "all", routed series with gradient)
    "cfg2": ("hbv", "Hbv", 7300, 671, 16, []),
    "cfg2dyn": ("hbv", "Hbv", 7300, 671, 16, ["parBETA", "parBETAET"]),
    # config 2 with learned ensemble weights (x_dict['muwts'], hbv.py:508-511): Qsim is the weighted sum of the members
    "cfg2mu": ("hbv", "Hbv", 7300, 671, 16, [], {"muwts": True}),
    # the headline with module key grad_buffer='persistent' (opt-in): the dense zero fill of the [T,B,ny] gradient -- what
    # the autograd contract costs -- happens once instead of every step
    "cfg2persist": ("hbv", "Hbv", 7300, 671, 16, [], {"grad_buffer": "persistent"}),
    "cfg2dynpersist": ("hbv", "Hbv", 7300, 671, 16, ["parBETA", "parBETAET"], {"grad_buffer": "persistent"}),
    "cfg3": ("hbv_1_1p", "Hbv_1_1p", 7300, 671, 16, "all"),
    "cfg4": ("hbv_adj", "HbvAdj", 7300, 671, 16, ["parBETAET"]),
    "cfg4persist": ("hbv_adj", "HbvAdj", 7300, 671, 16, ["parBETAET"], {"grad_buffer": "persistent"}),
    "cfg5": ("hbv_2", "Hbv_2", 730, 100000, 16, ["parBETA", "parK0", "parBETAET"]),
    "cfg5share": ("hbv_2", "Hbv_2", 730, 12500, 16, ["parBETA", "parK0", "parBETAET"]),
    # configs[4] at its stated size on ONE GPU (~70 GB of its 288): what `--config cfg5 --gpus 1` runs
    "cfg5full": ("hbv_2", "Hbv_2", 730, 100000, 16, ["parBETA", "parK0", "parBETAET"]),
    # the same with north_star's checkpoint-and-rematerialise adjoint (module key `adjoint_checkpoint` = K days): the
    # forward keeps 20 / K bytes per lane-day instead of 20, the streaming adjoint recomputes its K-day segment in LDS
    # (csrc/hbv_stream2_ckpt.h); Hbv_2.get_states() then returns the final storages, not the series
    "cfg5share_ck4": ("hbv_2", "Hbv_2", 730, 12500, 16, ["parBETA", "parK0", "parBETAET"], {"adjoint_checkpoint": 4}),
    "cfg5full_ck4": ("hbv_2", "Hbv_2", 730, 100000, 16, ["parBETA", "parK0", "parBETAET"], {"adjoint_checkpoint": 4}),
    "cfg5full_ck8": ("hbv_2", "Hbv_2", 730, 100000, 16, ["parBETA", "parK0", "parBETAET"], {"adjoint_checkpoint": 8}),
    # the deltaMG minibatch shape of SURVEY §8d: 100 basins, 365 warm-up + 365 days
    "dmg": ("hbv", "Hbv", 730, 100, 16, ["parBETA", "parBETAET"], {"warm_up": 365}),
    # the same step through captured HIP graphs (module key `graph`: hydrodl2_amd/graphed.py) -- opt-in, for steps whose
    # kernels are shorter than their enqueue time
    "dmggraph": ("hbv", "Hbv", 730, 100, 16, ["parBETA", "parBETAET"], {"warm_up": 365, "graph": True}),
    # configs[3] under the REFERENCE's Newton policy: joint modified Newton on the five storages, max_iter 3,
    # gtol 1e-3 (hbv_adj.py:544-581); `cfg4` above times this package's default, the staged solve
    "cfg4joint": ("hbv_adj", "HbvAdj", 7300, 671, 16, ["parBETAET"], {"newton_solver": "joint"}),
    # Hbv_2_hourly + gage routing (hbv_2_hourly.py:527-675,800-897): 4 000 units x 4 members x
    # 2 160 hours draining to 100 gages (~12 000 gage-unit pairs)
    "hourly": ("hbv_2_hourly", "Hbv_2_hourly", 2160, 4000, 4, ["parBETA", "parK0", "parBETAET"], {"gages": 100}),
}

# What actually limits each configuration on this chip, with the evidence on file (DESIGN.md §4 "what binds").  The
# `roofline` objects price every kernel against HBM (the contract's bounding roofline for an element-wise
# recurrence: "bound": "hbm"); `limited_by` says what the counters / probes show instead of pretending the HBM
# fraction is the lever: hbm | valu-issue | latency | host.
LIMITED_BY = {
    # name: (what limits it, evidence, vector instructions per SIMD and cycle of its kernels -- SQ_INSTS_VALU / SIMD-cycles
    # from the committed counter passes; a pure fma loop at four waves per SIMD reaches 0.70 -- , evidence file)
    "cfg2": ("latency", "forward: 168 workgroups x 7 300 serial days, the stepper waves' own ~205 cycles per day + the tile "
                        "barrier + contention for the workgroup's LDS pipeline (profiles/r04_pipe_helpers_probe.txt); adjoint "
                        "kernels: 4-5 waves per SIMD, each issuing one instruction per ~6-8 cycles, beside the 3.8 GB gradient "
                        "fill (5.4 TB/s in that window)",
             {"k_bwd_chunk_phi": 0.32, "k_bwd_chunk_sweep": 0.32, "k_fwd_pipe": 0.11}, "profiles/r04_sq_counters_cfg2.txt"),
    "cfg2dyn": ("latency", "as cfg2; the soil wave carries two dynamic powers", {}, "profiles/r04_sq_counters_cfg2.txt"),
    "cfg2persist": ("latency", "as cfg2 without the per-step 3.8 GB gradient fill (and without its share of HBM beside the "
                               "adjoint)", {}, "profiles/r05_persist_ab.jsonl"),
    "cfg2mu": ("latency", "as cfg2; the ensemble weights are one staged row of the pipelined forward (round 5), the adjoint "
                          "takes the generic time-parallel instances", {}, "profiles/r05_slotlist_ab.jsonl"),
    "cfg3": ("hbm", "14 dynamic rows streamed twice by the two-pass adjoint: 17.3 GB in 3.6 ms = 4.8 TB/s (76 % of the "
                    "6.3 TB/s this chip copies at); forward bound by its filler waves",
             {"k_bwd_chunk_phi": 0.20, "k_bwd_chunk_sweep": 0.17, "k_fwd_pipe": 0.12}, "profiles/r04_sq_counters_cfg3.txt"),
    "cfg4": ("latency", "the soil-moisture wave's scalar Newton / Halley iteration: a lone wave on its SIMD, ~8 cycles per "
                        "instruction; 97.1 % of the lane-days are solved by one free update but the slowest of a wave's 64 "
                        "lanes asks for a second on 54 % of the wave-days (profiles/r04_ab_soil.txt)", {},
             "profiles/r04_ab_soil.txt"),
    "cfg4joint": ("latency", "one wave per 64 lanes iterating the reference's joint 5-variable Newton", {}, "DESIGN.md §4"),
    "cfg5share": ("hbm", "design bytes (20 B trajectory + 12 B dynamic rows per lane-day, both ways) at 4.4-4.5 TB/s measured "
                         "by the counters; the vector pipes are a third busy",
                  {"k_bwd_stream2": 0.20, "k_fwd_stream2": 0.13}, "profiles/r04_sq_counters_cfg5.txt"),
    "cfg5full": ("hbm", "as cfg5share at 25 000 waves: forward 5.4 TB/s, adjoint 4.7 TB/s of design bytes (86 / 75 % of the "
                        "6.3 TB/s this chip copies at)", {"k_bwd_stream2": 0.20, "k_fwd_stream2": 0.13},
                 "profiles/r04_sq_counters_cfg5.txt"),
    "cfg5": ("hbm", "as cfg5share", {"k_bwd_stream2": 0.20, "k_fwd_stream2": 0.13}, "profiles/r04_sq_counters_cfg5.txt"),
    "dmg": ("host", "kernels sum to less than the enqueue time of the step's launches (tools/host_overhead.py)", {},
            "profiles/r04_host_overhead.txt"),
    "dmggraph": ("latency", "two graph launches per step; what is left is the kernels' own time and the gaps between "
                            "the ~20 nodes of the two graphs", {}, "profiles/r04_host_overhead.txt"),
    "hourly": ("latency", "two-stage pipelined forward (1 000 workgroups), gage routing FIR pair-parallel", {}, "DESIGN.md §4"),
    "lstm": ("latency", "per time step one L1-bypassing store -> load hand-off between the workgroups of a row tile "
                        "(~1.2 us) + H/4 MFMAs; 4 % of HBM peak, 12 % of the fp32 MFMA peak", {}, "DESIGN.md §4"),
    "dplgraph": ("latency", "as dpl with no host in the step: one graph launch; what is left is the kernels' own time "
                            "(LSTM recurrence 4.0 ms, library GEMMs 1.6 ms, HBV 0.3 ms, element-wise / Adam 0.3 ms)", {},
                 "profiles/r05_dpl_graph.txt"),
    "dpl": ("latency", "LSTM recurrence (two persistent kernels) + four fp32 library GEMMs + the HBV calls", {},
            "profiles/r04_kernel_stats.csv"),
}


def limited_by(name):
    """What binds the configuration.  These are RECORDED findings (the committed counter / probe files named in
    `evidence_file`, taken with the profiler), not measurements of this run: the keys say so."""
    lb, why, busy, src = LIMITED_BY[name]
    out = {"limited_by": lb, "evidence_recorded": why, "evidence_file": src}
    if busy:
        out["valu_per_simd_cycle_recorded"] = busy
    return out


class Workload:
    """One model + resident synthetic inputs + the step closure."""

    def __init__(self, name, dev, seed, B=None, T=None, M=None):
        import torch
        import hydrodl2_amd
        fam, cls, T0, B0, M0, dyn = WORKLOADS[name][:6]
        extra = dict(WORKLOADS[name][6]) if len(WORKLOADS[name]) > 6 else {}
        gages = int(extra.pop("gages", 0))       # (not a module key: the synthetic gage topology of `hourly`)
        with_mu = bool(extra.pop("muwts", False))  # (not a module key either: an input)
        self.name, self.T, self.B, self.M = name, T or T0, B or B0, M or M0
        T, B, M = self.T, self.B, self.M
        C = hydrodl2_amd.load_model(fam, cls)
        if dyn == "all":
            dyn = list(C(None, dev).parameter_bounds)
        self.model = C({"nmul": M, "dynamic_params": {cls: list(dyn)}, **extra}, dev)
        self.T_out = T - int(extra.get("warm_up", 0))     # warm_up_states (default): outputs start after the warm-up
        self.n_dyn = len(dyn)
        self.routed = bool(self.model.routing) and cls != "Hbv_2_hourly"    # (hourly: gage routing, priced in extra_bytes)
        self.n_flux = {"Hbv": 11, "HbvAdj": 1}.get(cls, 12)
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        x = synth_forcing(T, B, dev, g)
        self.xd = {"x_phy": x}
        if with_mu:
            u = torch.rand((T, B, M), generator=g, device=dev) + 0.25
            self.xd["muwts"] = (u / u.sum(-1, keepdim=True)).contiguous()
        self.extra_bytes = (0.0, 0.0)       # per launch, beyond the recurrence: (forward, backward)
        out_cols = B
        if cls == "Hbv_2_hourly":
            # per-step depths at an hourly step; every gage drains ~2 % of the units + one unit of its own
            x = x * torch.tensor([1 / 8.0, 1.0, 1 / 24.0], device=dev)
            self.xd["x_phy"] = x
            topo = (torch.rand((gages, B), generator=g, device=dev) < 0.02).float()
            topo[torch.arange(B, device=dev) % gages, torch.arange(B, device=dev)] = 1.0
            pd = torch.rand((T, B, self.n_dyn * M), generator=g, device=dev).requires_grad_(True)
            ps = torch.rand((B, (19 - self.n_dyn) * M), generator=g, device=dev).requires_grad_(True)
            pr = torch.rand((int(topo.sum()), 3), generator=g, device=dev).requires_grad_(True)
            self.params, self.leaves, self.shared = (pd, ps, pr), [pd, ps, pr], ps
            self.xd["ac_all"] = torch.rand(B, generator=g, device=dev) * 5000
            self.xd["elev_all"] = torch.rand(B, generator=g, device=dev) * 3000
            self.xd["outlet_topo"] = topo
            self.xd["areas"] = torch.rand(B, generator=g, device=dev) * 90 + 5
            self.n_pairs = int(topo.sum())
            out_cols = gages
            # gage routing (include/hbvx.h: hbvx_gage_route_*): 4 (T U + T G) bytes each way + the pair parameters
            gb = 4.0 * (T * B + T * gages) + 12.0 * self.n_pairs
            self.extra_bytes = (gb, gb)
        elif cls == "Hbv_2":
            pd = torch.rand((T, B, self.n_dyn * M), generator=g, device=dev).requires_grad_(True)
            ps = torch.rand((B, (16 - self.n_dyn) * M), generator=g, device=dev).requires_grad_(True)
            self.params, self.leaves, self.shared = (pd, ps), [pd, ps], ps
            self.xd["ac_all"] = torch.rand(B, generator=g, device=dev) * 5000
            self.xd["elev_all"] = torch.rand(B, generator=g, device=dev) * 3000
        else:
            p = torch.randn((T, B, self.model.learnable_param_count), generator=g, device=dev).requires_grad_(True)
            self.params, self.leaves, self.shared = p, [p], p
        self.key = "flow_sim" if cls == "HbvAdj" else "streamflow"
        self.w = torch.randn((self.T_out, out_cols, 1), generator=g, device=dev)
        self.cls = cls

    @property
    def lane_steps(self):
        return self.B * self.M * self.T

    def step(self):
        for leaf in self.leaves:
            leaf.grad = None
        out = self.model(self.xd, self.params)
        loss = (out[self.key] * self.w).sum()
        loss.backward()
        self.last_loss = loss.detach()
        return loss

    def shared_grad(self):
        """What a shared parameterisation network would receive from this shard, basin-summed:
        the last-row gradient of the raw NN output (hbv) or the static-parameter gradient (hbv_2)."""
        g = self.shared.grad
        return g[-1].sum(0) if g.dim() == 3 else g.sum(0)

    def alg_bytes(self):
        """SURVEY.md §8d algorithmic HBM bytes per lane-step (fp32, K-day checkpoints, n_g = the
        series with incoming gradient): (forward kernel, backward kernel, routing fwd, routing bwd)."""
        M, nd = self.M, self.n_dyn
        n_g = 4 if self.routed else 1
        fwd = 12.0 / M + 4.0 * nd + 4.0 * self.n_flux / M + 20.0 / CKPT_K
        bwd = 12.0 / M + 4.0 * nd + 4.0 * nd + 4.0 * n_g / M + 20.0 / CKPT_K
        rt = 32.0 / M if self.routed else 0.0
        return fwd, bwd, rt, rt

    def design_bytes(self):
        """Bytes this build's kernels move per lane-step by construction: the forward
        saves the whole trajectory (20 B: the five storages; the two powers are recomputed since round 4)
        instead of K-day checkpoints; the time-parallel adjoint (small grids) reads its inputs twice, the
        streaming adjoint (large grids) once."""
        M, nd = self.M, self.n_dyn
        n_g = 4 if self.routed else 1
        fwd = 12.0 / M + 4.0 * nd + 4.0 * self.n_flux / M + 20.0
        one = 12.0 / M + 4.0 * nd + 20.0 + 4.0 * n_g / M
        passes = 1 if self.B * self.M >= 768 * 64 else 2      # the cross-over of hbvx.hip's dispatch
        return fwd, passes * one + 4.0 * nd


def _median(v):
    s = sorted(v)
    n = len(s)
    return s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2])


def timed_steps(wl, steps, warmup, dev, world, after_step=None):
    """W untimed + K timed steps; returns (seconds for the K steps, per-call avg ms from per-launch HIP events).

    The K steps are ONE timed region (barrier + synchronize on both sides: the contract's `value`).  Inside it every
    step leaves a HIP event on the launch stream and a host time stamp -- no synchronisation -- so the record also
    carries the per-step samples (median of >= 10): `timed_steps.samples` = {n, median / min / max of
    the device time between consecutive step events, the host's enqueue time per step}.  A step that the whole-
    region mean hides (round 4: 11.9 ms mean against 3 ms of kernels in the driver's run of cfg5share) shows as `max`."""
    timed_steps.device_mallocs = 0
    timed_steps.samples = None
    import gc
    import torch
    import torch.distributed as dist
    from hydrodl2_amd import ops
    for _ in range(warmup):
        wl.step()
        if after_step:
            after_step()
    ops.KERNEL_EVENTS = []          # per-launch HIP events on the launch stream
    # no collector pause inside the timed steps: the previous workload's cycles (autograd graphs, module caches) are
    # reclaimed here, and the cyclic collector stays off until the region ends
    gc.collect()
    gc_was = gc.isenabled()
    gc.disable()
    if world > 1:
        dist.barrier()
    mallocs0 = 0
    cuda = dev.type == "cuda"
    marks, host = [], []
    if cuda:
        torch.cuda.synchronize()
        mallocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(e)
    t0 = time.perf_counter()
    for _ in range(steps):
        h0 = time.perf_counter()
        wl.step()
        if after_step:
            after_step()
        host.append(time.perf_counter() - h0)
        if cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(e)
    if cuda:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    if cuda:
        # a hipMalloc inside the timed region (the caching allocator still growing: tens of GB at cfg5) is a
        # host stall of tens to hundreds of ms that no kernel time shows; reported so that a step time can be trusted
        timed_steps.device_mallocs = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - mallocs0
        dev_ms = [a.elapsed_time(b) for a, b in zip(marks, marks[1:])]
        timed_steps.samples = {"n": len(dev_ms), "ms_median": round(_median(dev_ms), 4), "ms_min": round(min(dev_ms), 4),
                               "ms_max": round(max(dev_ms), 4), "ms_mean_region": round(1e3 * dt / steps, 4),
                               "host_enqueue_ms_median": round(1e3 * _median(host), 4),
                               "host_enqueue_ms_max": round(1e3 * max(host), 4),
                               "what": "device time between consecutive end-of-step HIP events on the launch stream"}
    events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    per = {}
    for name, e0, e1 in events:
        per.setdefault(name, []).append(e0.elapsed_time(e1))
    # a step may launch a call twice (warm-up + main forward): report time per STEP
    return dt, {k: sum(v) / steps for k, v in per.items()}


_PMC = None


def pmc_traffic(name):
    """HBM bytes per ABI call from the committed counter passes (profiles/pmc_traffic.json: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate passes, corrected as DESIGN.md §5 states) for configuration `name`, or None.
    Not re-measured in the run: counters need the profiler."""
    global _PMC
    if _PMC is None:
        try:
            _PMC = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except Exception:
            _PMC = {}
    tab = _PMC.get("configs", {}).get(name) or (_PMC if name == "cfg2" else None)
    if not tab:
        return None
    out = {k: v.get("hbm_bytes") for k, v in tab.items() if isinstance(v, dict) and "hbm_bytes" in v}
    return out or None


def roofline_entry(wl, kms, ms_per_step):
    """HBM roofline figures of one workload from its per-call kernel times."""
    ls = wl.lane_steps
    f8, b8, rf8, rb8 = wl.alg_bytes()
    fd, bd = wl.design_bytes()
    xf, xb = wl.extra_bytes
    fk = "hbvx_adj_forward" if wl.cls == "HbvAdj" else "hbvx_forward"
    bk = "hbvx_adj_backward" if wl.cls == "HbvAdj" else "hbvx_backward"
    out = {"bytes_per_lane_step_8d": {"fwd": round(f8, 3), "bwd": round(b8, 3), "route": round(rf8 + rb8, 3),
                                      "K": CKPT_K},
           "kernel_ms": {k: round(v, 4) for k, v in kms.items()}}
    if xf or xb:
        out["gage_routing_bytes_per_launch"] = {"fwd": xf, "bwd": xb}
    if fk in kms and bk in kms:
        gf = f8 * ls / (kms[fk] * 1e-3) / 1e9
        gb = b8 * ls / (kms[bk] * 1e-3) / 1e9
        whole = ((f8 + b8 + rf8 + rb8) * ls + xf + xb) / (ms_per_step * 1e-3) / 1e9
        out.update({"fwd_GBps_8d": round(gf, 1), "bwd_GBps_8d": round(gb, 1), "step_GBps_8d": round(whole, 1),
                    "frac_fwd": round(gf / HBM_PEAK_GBPS, 4), "frac_bwd": round(gb / HBM_PEAK_GBPS, 4),
                    "frac_step": round(whole / HBM_PEAK_GBPS, 4),
                    "fwd_GBps_design": round(fd * ls / (kms[fk] * 1e-3) / 1e9, 1),
                    "bwd_GBps_design": round(bd * ls / (kms[bk] * 1e-3) / 1e9, 1)})
    if wl.name in LIMITED_BY:
        out.update(limited_by(wl.name))
    tr = pmc_traffic(wl.name)
    if tr:
        out["traffic"] = tr
        out["traffic_source"] = "profiles/pmc_traffic.json (committed rocprofv3 --pmc passes; not re-measured in this run)"
    return out


class LstmWorkload:
    """SURVEY §8f rank 4, the caller side: the hand-written sequence LSTM (include/hbvx_lstm.h, csrc/lstm_seq.h),
    forward + backward at the deltaMG shape (T = 730 days, B = 100 basins, H = 256), torch.nn.LSTM semantics."""
    name, cls = "lstm", "SeqLSTM"

    def __init__(self, dev, seed, T=730, B=100, H=256):
        import torch
        from hydrodl2_amd.lstm import SeqLSTM
        torch.manual_seed(seed)
        self.T, self.B, self.H = T, B, H
        self.net = SeqLSTM(H, H).to(dev)
        self.x = torch.randn(T, B, H, device=dev, requires_grad=True)
        self.gh = torch.randn(T, B, H, device=dev)

    def step(self):
        for p in self.net.parameters():
            p.grad = None
        self.x.grad = None
        (self.net(self.x)[0] * self.gh).sum().backward()

    def entry(self, ms, kms):
        T, B, H = self.T, self.B, self.H
        # algorithmic HBM bytes of the two recurrence kernels (tools/bench_lstm.py): gate pre-activations in, gates /
        # cell / hidden series out; backward: those back in, gate gradients out
        bf, bb = T * B * H * 4.0 * (4 + 4 + 1 + 1), T * B * H * 4.0 * (4 + 4 + 2 + 1)
        tiles = (B + 15) // 16
        flops = 2.0 * T * tiles * 16 * H * 4 * H            # padded row tiles; the same for either direction
        e = {"config": "lstm", "what": f"SeqLSTM fwd+bwd, T={T} B={B} I=H={H} (recurrence kernels + 4 fp32 library GEMMs)",
             "T": T, "B": B, "H": H, "steps": 5, "ms_per_step": round(ms, 4),
             "sequence_steps_per_s": T * B / (ms * 1e-3), "kernel_ms": {k: round(v, 4) for k, v in kms.items()}}
        for d, by in (("forward", bf), ("backward", bb)):
            k = f"hbvx_lstm_{d}"
            if k in kms:
                e[f"{d}_GBps_algorithmic"] = round(by / (kms[k] * 1e-3) / 1e9, 1)
                e[f"{d}_frac_hbm"] = round(by / (kms[k] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                e[f"{d}_mfma_tflops"] = round(flops / (kms[k] * 1e-3) / 1e12, 2)
                e[f"{d}_us_per_time_step"] = round(kms[k] * 1e3 / T, 2)
        e.update(limited_by("lstm"))
        return e


class DplWorkload:
    """One step of examples/train_dpl.py: LSTM parameter network -> Hbv (2 dynamic parameters, 365 + 365 days, 100
    basins x 16) -> 1 - NSE -> backward -> Adam, one rank (the bucketed all-reduce is a no-op at world 1)."""
    name, cls = "dpl", "train_dpl"

    def __init__(self, dev, seed, graph=False):
        import importlib.util
        spec = importlib.util.spec_from_file_location("train_dpl", os.path.join(ROOT, "examples", "train_dpl.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        self.graph = graph
        self.name = "dplgraph" if graph else "dpl"
        # HBVX_BENCH_TUNE_GEMM=0: no TunableOp pass (profiling runs: the tuner's trial launches flood a kernel trace)
        tune = graph and os.environ.get("HBVX_BENCH_TUNE_GEMM", "1") not in ("", "0")
        self.tuned = tune
        self.step, self.info = mod.make_trainer(dev, graph=graph, tune_gemm=tune)

    def entry(self, ms, kms):
        i = self.info
        hbv = sum(v for k, v in kms.items() if "lstm" not in k)
        lstm = sum(v for k, v in kms.items() if "lstm" in k)
        e = {"config": self.name,
             "what": "examples/train_dpl.py step: LSTM-256 -> Hbv -> 1-NSE -> Adam (fused LSTM kernels)"
                     + (("; the whole step replayed as ONE captured HIP graph" + (" with TunableOp's GEMM picks (--graph --tune-gemm)"
                                                                                    if self.tuned else " (--graph)"))
                        if self.graph else ""),
             "T": i["T"], "B": i["B"], "M": i["M"], "steps": 5, "ms_per_step": round(ms, 4),
             "lane_steps_per_s": i["T"] * i["B"] * i["M"] / (ms * 1e-3),
             "hbv_calls_ms": round(hbv, 4), "lstm_kernels_ms": round(lstm, 4),
             "kernel_ms": {k: round(v, 4) for k, v in kms.items()}}
        e.update(limited_by(self.name))
        return e


def tuned_graph_entry_from_child():
    """The `dplgraph` entry (whole training step as one HIP graph, TunableOp's GEMM picks) measured by tools/bench_one.py in a
    child process; if the child fails, the workload runs here without the tuner."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_one.py"), "dplgraph", "--steps", str(SEC_STEPS), "--warmup", "3"]
    env = dict(os.environ, HBVX_BENCH_TUNE_GEMM="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        r = json.loads(line)
        e = {"config": "dplgraph",
             "what": "examples/train_dpl.py step: LSTM-256 -> Hbv -> 1-NSE -> Adam (fused LSTM kernels); the whole step replayed "
                     "as ONE captured HIP graph with TunableOp's GEMM picks (--graph --tune-gemm); measured in a child process",
             "T": 730, "B": 100, "M": 16, "ms_per_step": r["ms_median"], "lane_steps_per_s": 730 * 100 * 16 / (r["ms_median"] * 1e-3),
             "steps": SEC_STEPS, "ms_per_step_is": "median of the per-step samples",
             "step_samples": {"n": SEC_STEPS, "ms_median": r["ms_median"], "ms_min": r["ms_min"], "ms_max": r["ms_max"],
                              "ms_mean_region": r["ms_mean_region"], "host_enqueue_ms_median": r["host_enqueue_ms_median"]},
             "kernel_ms": {}, "kernel_sum_ms": 0, "device_mallocs_in_timed_steps": r.get("device_mallocs")}
        e.update(limited_by("dplgraph"))
        return e
    except Exception as ex:  # noqa: BLE001 -- the child died or printed nothing: same workload, no tuner, in this process
        import torch
        os.environ["HBVX_BENCH_TUNE_GEMM"] = "0"
        dev = torch.device("cuda", torch.cuda.current_device())
        w = DplWorkload(dev, 7, graph=True)
        dt, k = timed_steps(w, SEC_STEPS, 3, dev, 1)
        smp = timed_steps.samples
        e = w.entry(smp["ms_median"] if smp else 1e3 * dt / SEC_STEPS, k)
        e.update({"steps": SEC_STEPS, "step_samples": smp, "tuner_child_failed": repr(ex)[:160]})
        return e


def make_workload(name, dev, seed=7):
    """Any named workload of the bench (tools/bench_one.py)."""
    if name == "lstm":
        return LstmWorkload(dev, seed)
    if name in ("dpl", "dplgraph"):
        return DplWorkload(dev, seed, graph=name == "dplgraph")
    return Workload(name, dev, seed=seed)


# --------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only): bounded samples of the same workload on the host cores
# --------------------------------------------------------------------------------------------
def host_cores() -> int:
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU
    box shows all of the host's cores to os.cpu_count() but grants a share; 128 eager-torch threads on a
    16-core share spin against each other and never finish)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


def host_memory_gb() -> float:
    """Host memory this process may use: MemAvailable capped by the cgroup limit."""
    avail = 0.0
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) / 1e6
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            txt = open(path).read().strip()
            if txt != "max":
                lim = int(txt) / 1e9
                used = 0.0
                try:
                    used = int(open(os.path.join(os.path.dirname(path), "memory.current")).read()) / 1e9
                except Exception:
                    pass
                avail = min(avail, lim - used) if avail else lim - used
            break
        except Exception:
            continue
    return avail


def cpu_baseline_port(B, M, T_sample, seed=0):
    """The CPU oracle (oracle/, C, OpenMP over basins) through the same C ABI."""
    import ctypes as C
    import numpy as np
    import __graft_entry__ as ge
    from hydrodl2_amd import _abi
    if not os.path.exists(ge.ORACLE_LIB):
        ge.build_oracle()
    lib = _abi.Library(ge.ORACLE_LIB)
    try:    # torch has usually started the OpenMP runtime already: set the team size through its API
        C.CDLL("libgomp.so.1").omp_set_num_threads(C.c_int(host_cores()))
    except OSError:
        pass
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
    while reps < 3 or (dt < 2.0 and reps < 20):   # one pass is well under a second on a big host
        t0 = time.perf_counter()
        lib.forward(d, out, 0)
        lib.backward(d, io, 0)
        dt += time.perf_counter() - t0
        reps += 1
    return {"value": reps * B * M * T / dt, "unit": "basin-ensemble-timesteps/s", "cores": threads,
            "kind": "port",
            "sample": f"oracle (C, OpenMP over basins) fwd+bwd recurrence, {reps} passes over {B}x{M}x{T_sample} "
                      f"lane-steps, {dt:.2f} s wall on {threads} threads"}


def _eager_child(B, M, T, grad=True):
    """One eager pass (runs in a child process: bench.py --eager-child B M T [grad|nograd])."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("hbv_torch_eager", os.path.join(ROOT, "oracle", "hbv_torch_eager.py"))
    eager = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(eager)
    cores = host_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    x = synth_forcing(T, B, torch.device("cpu"), g)
    # static parameters: only the last row of the raw tensor is live (hbv.py:242); an expanded view keeps the
    # T = 7300 no-grad pass at 0.5 MB instead of 3.8 GB of host memory
    row = torch.randn((1, B, 12 * M + 2), generator=g)
    if grad:
        p = row.expand(T, B, 12 * M + 2).clone().requires_grad_(True)
    else:
        p = row.expand(T, B, 12 * M + 2)
    w = torch.randn((T, B, 1), generator=g)
    t0 = time.perf_counter()
    if grad:
        out = eager.hbv_eager(x, p, M)
    else:
        with torch.no_grad():
            out = eager.hbv_eager(x, p, M)
    t1 = time.perf_counter()
    if grad:
        (out["streamflow"] * w).sum().backward()
    t2 = time.perf_counter()
    print(json.dumps({"T": T, "fwd_s": t1 - t0, "bwd_s": t2 - t1, "cores": cores, "grad": bool(grad)}))


def cpu_baseline_eager(B, M, budget_s=90.0):
    """The reference's kind of CPU path, SURVEY.md §8d's protocol: PyTorch eager, one ATen call per operator
    per day plus the autograd tape (oracle/hbv_torch_eager.py, pinned to the reference's fixtures in the CPU
    tests; the reference itself cannot travel to this box), all granted host cores.  Passes, each in a child
    process under a hard time limit (many-core hosts can be very slow on these tiny operators):
      fwd+bwd at T = 365 (the dMG window) and T = 730 (the eager backward is O(T^2) for static parameters,
      SURVEY.md §3.3), forward only under no_grad at T = 7300, and T = 7300 fwd+bwd reported as twenty
      365-day windows (dMG practice) -- flagged "windowed".  `value` is the T = 365 fwd+bwd rate (falling
      back to shorter records when that pass does not finish)."""
    spent, passes = 0.0, {}

    def one(T, grad, limit):
        nonlocal spent
        t0 = time.perf_counter()
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--eager-child", str(B), str(M), str(T),
                                  "grad" if grad else "nograd"], capture_output=True, text=True, timeout=limit)
            rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as ex:
            rec = {"T": T, "error": type(ex).__name__}
        spent += time.perf_counter() - t0
        return rec

    head = None
    for T in (30, 120, 365):
        left = budget_s * 0.4 - spent
        if left < 3.0 or (head and (head["fwd_s"] + head["bwd_s"]) * (T / head["T"]) ** 2 > left):
            break
        rec = one(T, True, left)
        if "error" in rec:
            break
        head = rec
    if head is None:
        return {"error": f"no eager pass finished within {budget_s * 0.4:.0f} s", "kind": "restatement"}
    s365 = head["fwd_s"] + head["bwd_s"]
    passes[f"fwd+bwd T={head['T']}"] = {"s": round(s365, 3), "fwd_s": round(head["fwd_s"], 3),
                                        "bwd_s": round(head["bwd_s"], 3), "lane_steps_per_s": B * M * head["T"] / s365}
    if head["T"] == 365:
        left = budget_s * 0.75 - spent
        if left > 4.0 * s365:       # T = 730: forward x2, static-parameter backward up to x4
            rec = one(730, True, left)
            if "error" not in rec:
                s = rec["fwd_s"] + rec["bwd_s"]
                passes["fwd+bwd T=730"] = {"s": round(s, 3), "fwd_s": round(rec["fwd_s"], 3), "bwd_s": round(rec["bwd_s"], 3),
                                           "lane_steps_per_s": B * M * 730 / s}
            else:
                passes["fwd+bwd T=730"] = rec
        left = budget_s - spent
        need_gb = 4.0 * 31 * B * M * 7300 / 1e9     # what the tape-free eager forward allocates (12 parameter, 3 forcing,
        if host_memory_gb() < 2.0 * need_gb:       # 11 + 5 output / state buffers of [T,B,M]); never risk the box
            passes["fwd only (no_grad) T=7300"] = {"skipped": f"needs ~{need_gb:.0f} GB of host memory, "
                                                              f"{host_memory_gb():.0f} GB granted"}
        elif left > 25.0 * head["fwd_s"]:
            rec = one(7300, False, left)
            if "error" not in rec:
                passes["fwd only (no_grad) T=7300"] = {"s": round(rec["fwd_s"], 3), "lane_steps_per_s": B * M * 7300 / rec["fwd_s"]}
            else:
                passes["fwd only (no_grad) T=7300"] = rec
        passes["fwd+bwd T=7300 as 20 x 365-day windows"] = {
            "windowed": True, "s": round(20 * s365, 3), "lane_steps_per_s": B * M * 365 / s365,
            "note": "20 x the measured T=365 pass (dMG trains on 365-day windows); not one 7300-day tape"}
    return {"value": B * M * head["T"] / s365, "unit": "basin-ensemble-timesteps/s", "cores": head["cores"],
            "kind": "restatement",
            "what": "pure-torch eager restatement of hbv.py:363-596 (reference-equivalent CPU path; leaner than the "
                    "reference's own module: no dict assembly, no parameter repeat)",
            "sample": f"fwd+bwd, one pass over {B}x{M}x{head['T']} lane-steps: fwd {head['fwd_s']:.2f} s + bwd "
                      f"{head['bwd_s']:.2f} s on {head['cores']} torch threads; {spent:.0f} s spent on all passes",
            "passes": passes}


def validate_collective(dev, world, rank):
    """First use of the backend, self-validating: all-reduce a ones-vector and require `world` in every element, so
    that a record of an N-rank run says "the collective summed N ranks", not "N processes ran".  Every rank reports
    its device and the RCCL version on stderr (the launcher relays it with a rank prefix)."""
    import torch
    import torch.distributed as dist
    info = {"backend": None, "ranks_summed": 1, "nccl_version": None}
    name = torch.cuda.get_device_name(dev) if dev.type == "cuda" else "cpu"
    if world > 1:
        info["backend"] = dist.get_backend()
        ones = torch.ones(64, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        lo, hi = float(ones.min().item()), float(ones.max().item())
        if lo != float(world) or hi != float(world):
            raise SystemExit(f"rank {rank}: all-reduce of ones over {world} ranks returned [{lo}, {hi}]")
        info["ranks_summed"] = int(lo)
        # and a rank-dependent payload: sum_r (r + 1) = world (world + 1) / 2 catches a collective that talks to itself
        v = torch.full((4,), float(rank + 1), device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        if float(v[0].item()) != world * (world + 1) / 2:
            raise SystemExit(f"rank {rank}: all-reduce of rank ids returned {float(v[0].item())}")
    if dev.type == "cuda":
        try:
            info["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            info["nccl_version"] = None
    print(f"[bench] rank {rank}/{world}: device {name}, backend {info['backend']}, rccl {info['nccl_version']}, "
          f"ones summed over {info['ranks_summed']} rank(s)", file=sys.stderr, flush=True)
    return info


# --------------------------------------------------------------------------------------------
def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--eager-child":
        _eager_child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), grad=(sys.argv[5:6] != ["nograd"]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["cfg2", "cfg5"], default="cfg2")
    ap.add_argument("--basins", type=int, default=None)
    ap.add_argument("--nmul", type=int, default=None)
    ap.add_argument("--days", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--cpu-sample-days", type=int, default=1825)
    ap.add_argument("--device", choices=["cuda", "cpu"], default="cuda",
                    help="cpu exists for the launcher's own test (tests/test_bench_launch.py swaps in a host "
                         "implementation of the ABI); the product library refuses host tensors")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("HBVX_BENCH_FAIL_RANK") == str(rank):   # tests/test_bench_launch.py: a rank dies early
        raise SystemExit(f"rank {rank}: injected failure before the rendezvous")
    if args.device == "cuda":
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from datetime import timedelta
        # a rank that never arrives must not hold the others for the default 10 minutes
        tmo = timedelta(seconds=int(os.environ.get("HBVX_BENCH_RDZV_TIMEOUT", "120")))
        if dev.type == "cuda":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group("gloo", timeout=tmo)
    collective = validate_collective(dev, world, rank)

    from hydrodl2_amd import sharding
    strong = args.config == "cfg5"
    _, _, T0, B0, M0 = WORKLOADS[args.config][:5]
    B_total = args.basins or B0
    if strong:
        b0, b1 = sharding.basin_range(B_total, world, rank)
        B_rank = b1 - b0
    else:
        B_rank = B_total
    wl = Workload(args.config, dev, seed=1000 + rank, B=B_rank, T=args.days or T0, M=args.nmul or M0)

    nshared = wl.shared.shape[-1]
    bucket = torch.zeros(nshared + 1, device=dev)
    ar_events = []

    def exchange():
        # the path's only collective (RCCL over xGMI): loss + shared-network gradient, one bucket
        bucket[:nshared] = wl.shared_grad()
        bucket[nshared] = wl.last_loss
        if world > 1 and dev.type == "cuda":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sharding.all_reduce_sum_([bucket])
            e1.record()
            ar_events.append((e0, e1))
        else:
            sharding.all_reduce_sum_([bucket])

    dt, kavg = timed_steps(wl, args.steps, args.warmup, dev, world, after_step=exchange)
    my_ms = 1e3 * dt / args.steps
    rank_ms = [my_ms]
    if world > 1:
        t = torch.zeros(world, device=dev)
        t[rank] = dt
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rank_ms = [round(1e3 * float(v) / args.steps, 4) for v in t]
        dt = float(t.max().item())
    ar_ms = None
    if ar_events:
        v = [a.elapsed_time(b) for a, b in ar_events[-args.steps:]]
        ar_ms = round(sum(v) / len(v), 4)

    lane_steps_job = (B_total if strong else B_total * world) * wl.M * wl.T
    value = lane_steps_job * args.steps / dt
    ms_per_step = 1e3 * dt / args.steps

    res = None
    if rank == 0:
        fam, cls = WORKLOADS[args.config][:2]
        what = (f"{fam} ({cls}) {B_total} basins x {wl.M} members x {wl.T} days, "
                + ("static parameters" if wl.n_dyn == 0 else f"{wl.n_dyn} dynamic parameters")
                + ", fwd+bwd, raw parameters resident in HBM")
        res = {
            "metric": "basin-ensemble-timesteps/sec fwd+bwd", "value": value,
            "unit": "basin-ensemble-timesteps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": what, "name": args.config, "basins_total": B_total if strong else B_total * world,
                       "basins_per_gpu": B_rank,
                       "basins_per_rank": [(lambda ab: ab[1] - ab[0])(sharding.basin_range(B_total, world, r)) if strong
                                           else B_total for r in range(world)],
                       "nmul": wl.M, "days": wl.T,
                       "parallelism": f"basin-shard x{world}"},
            # the number of ranks the collective itself summed over (validate_collective), not the launcher's word
            "rccl_ranks": collective["ranks_summed"], "collective_check": collective,
            "rank_ms_per_step": rank_ms, "allreduce_ms": ar_ms,
            "device_mallocs_in_timed_steps": timed_steps.device_mallocs,
            # per-step samples inside the one timed region (rank 0): median / min / max; `ms_per_step` above is the
            # contract's figure, the region's wall time / K, max over ranks
            "step_samples": timed_steps.samples,
            "kernel_sum_ms": round(sum(kavg.values()), 4),
        }
        if dev.type == "cuda" and "hbvx_forward" in kavg:
            # Dominant kernel = the one kernel behind hbvx_forward (rocprofv3 --stats: the largest single
            # kernel of the step; hbvx_backward is 4 kernels).  `achieved` follows SURVEY.md §8d: algorithmic
            # bytes per lane-step (K = 64-day checkpoints; routing is a separate kernel and not counted
            # here) x lane-steps per launch / the launch's HIP-event duration.  `frac_design` prices the
            # bytes this build moves by construction (full trajectory instead of checkpoints).
            f8, b8, rf8, rb8 = wl.alg_bytes()
            fd, bd = wl.design_bytes()
            ls = wl.lane_steps
            ach = f8 * ls / (kavg["hbvx_forward"] * 1e-3) / 1e9
            ach_d = fd * ls / (kavg["hbvx_forward"] * 1e-3) / 1e9
            src = os.path.join("profiles", "pmc_traffic.json")
            tr = pmc_traffic("cfg2" if args.config == "cfg2" else "cfg5share" if wl.B == 12500 else "none")
            traffic = tr.get("hbvx_forward") if tr else None
            res["roofline"] = {
                # the roofline `achieved` / `peak` are priced against (contract: hbm | mfma; no contraction on this path);
                # `limited_by` is what the counters and probes say actually binds this configuration
                "bound": "hbm", "kernel": "the kernel of hbvx_forward (k_fwd_pipe at cfg2, k_fwd_stream2 at cfg5)",
                **limited_by(args.config),
                "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBPS, 5), "frac_8d": round(ach / HBM_PEAK_GBPS, 5),
                "frac_design": round(ach_d / HBM_PEAK_GBPS, 5),
                "bytes_per_lane_step": {"8d_fwd": round(f8, 3), "8d_bwd": round(b8, 3), "8d_route": round(rf8 + rb8, 3),
                                        "design_fwd": round(fd, 3), "design_bwd": round(bd, 3), "K": CKPT_K},
                "traffic": traffic, "traffic_source": src + " (committed rocprofv3 --pmc passes of this command, "
                                                            "corrected as DESIGN.md §5 states; not re-measured in this run)",
                "avg_ms": round(kavg["hbvx_forward"], 4),
                "whole_step": roofline_entry(wl, kavg, ms_per_step),
            }
    del wl
    if dev.type == "cuda":
        torch.cuda.empty_cache()

    if rank == 0 and world == 1 and dev.type == "cuda" and not args.no_secondary:
        # the other BASELINE configs under the same clock: 5 timed steps each, same event timing
        sec = []
        print(f"[bench] headline done: {ms_per_step:.3f} ms/step; secondary configs ...", file=sys.stderr, flush=True)
        for name in ("cfg2dyn", "cfg2mu", "cfg2persist", "cfg3", "cfg4", "cfg4joint", "cfg5share", "cfg5full", "dmg", "dmggraph", "hourly", "lstm", "dpl", "dplgraph"):
            if name == args.config:
                continue
            try:
                if name == "dplgraph" and os.environ.get("HBVX_BENCH_TUNE_GEMM", "1") not in ("", "0"):
                    # torch's TunableOp times every library GEMM candidate of nine shapes on this device: third-party
                    # kernels this process has never run.  In a child process, so that a fault in one of them costs this
                    # entry, not the headline; the child times the workload with this file's own protocol
                    sec.append(tuned_graph_entry_from_child())
                    print(f"[bench] {name}: {sec[-1].get('ms_per_step', float('nan')):.3f} ms/step (child process)", file=sys.stderr, flush=True)
                    continue
                w2 = make_workload(name, dev, 7)
                dt2, k2 = timed_steps(w2, SEC_STEPS, 3, dev, 1)
                smp = timed_steps.samples
                # the MEDIAN of >= 10 timed steps (per-step device time, HIP events); the mean over the
                # region and the slowest step stand beside it
                ms2 = smp["ms_median"] if smp else 1e3 * dt2 / SEC_STEPS
                if name in ("lstm", "dpl", "dplgraph"):
                    e = w2.entry(ms2, k2)
                else:
                    e = {"config": name, "T": w2.T, "B": w2.B, "M": w2.M, "n_dyn": w2.n_dyn,
                         "ms_per_step": round(ms2, 4), "lane_steps_per_s": w2.lane_steps / (ms2 * 1e-3)}
                    e.update(roofline_entry(w2, k2, ms2))
                e["steps"] = SEC_STEPS
                e["ms_per_step_is"] = "median of the per-step samples"
                e["step_samples"] = smp
                e["kernel_sum_ms"] = round(sum(k2.values()), 4)
                e["device_mallocs_in_timed_steps"] = timed_steps.device_mallocs
                sec.append(e)
                print(f"[bench] {name}: {ms2:.3f} ms/step", file=sys.stderr, flush=True)
                del w2
            except Exception as ex:  # a secondary config must not take the headline down
                sec.append({"config": name, "error": repr(ex)[:200]})
            torch.cuda.empty_cache()
        res["secondary"] = sec

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        print("[bench] cpu baselines ...", file=sys.stderr, flush=True)
        # `value` = the reference-equivalent CPU path (PyTorch eager, one ATen call per operator per day + the autograd
        # tape: kind "restatement"); the C / OpenMP oracle port -- a far better CPU program than the
        # reference is -- stands beside it under "port".  If no eager pass finishes in its budget the port is the value.
        port = cpu_baseline_port(671, 16, min(args.cpu_sample_days, 7300))
        print(f"[bench] oracle port: {port['value']:.3g} lane-steps/s; eager restatement ...", file=sys.stderr, flush=True)
        try:
            cb = cpu_baseline_eager(671, 16)
        except Exception as ex:
            cb = {"error": repr(ex)[:200], "kind": "restatement"}
        if "value" in cb:
            cb["port"] = port
        else:
            cb = dict(port, eager=cb)
        res["cpu_baseline"] = cb
    if rank == 0:
        print(json.dumps(res))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
