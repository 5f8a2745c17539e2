#!/usr/bin/env python3
"""End-to-end differentiable-parameter-learning step around the HBV plug-in (SURVEY.md §8f rank 4:
the caller side, outside the reference repository -- delta-MG style).

    python examples/train_dpl.py [--basins 100] [--rho 365] [--warm-up 365] [--nmul 16] [--steps 20]
                                 [--lstm fused|torch] [--graph] [--tune-gemm]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/train_dpl.py ...

A small LSTM maps normalised forcings + static attributes to the raw parameter tensor [T,B,ny];
`hydrodl2_amd.load_model('hbv')` turns it into streamflow; the loss is 1 - NSE per basin.  With
several processes every rank owns a contiguous block of basins (no collective inside the physics);
the LSTM gradients and the loss normalisers travel in ONE bucketed all-reduce per step
(`hydrodl2_amd.sharding.all_reduce_sum_`).  Data are synthetic: "observations" are produced by the
same physics with a hidden parameter field, so the loss has a meaningful minimum.
Prints one JSON line with ms/step and the share spent in the HBV calls.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import hydrodl2_amd  # noqa: E402
from hydrodl2_amd import ops, sharding  # noqa: E402
from hydrodl2_amd.lstm import SeqLSTM  # noqa: E402


class ParamNet(torch.nn.Module):
    """LSTM parameterisation network: [T,B,n_in] -> raw parameters [T,B,ny]."""

    def __init__(self, n_in: int, hidden: int, ny: int, fused: bool = False):
        super().__init__()
        self.inp = torch.nn.Linear(n_in, hidden)
        # fused: hydrodl2_amd.lstm.SeqLSTM (include/hbvx_lstm.h, one persistent kernel per direction);
        # otherwise torch's own LSTM (MIOpen on the GPU).  Same parameters either way.
        self.lstm = SeqLSTM(hidden, hidden) if fused else torch.nn.LSTM(hidden, hidden)
        self.out = torch.nn.Linear(hidden, ny)

    def forward(self, z):
        h, _ = self.lstm(torch.relu(self.inp(z)))
        return self.out(h)


def nse_loss(sim, obs, inv_den=None):
    """mean over basins of 1 - NSE; sim/obs [T,B].  Returns (sum over local basins, count).  `inv_den`: the
    reciprocal of the observations' sum of squares about their mean per basin (a constant of the data set:
    nse_inv_den) -- without it it is formed here."""
    num = ((sim - obs) ** 2).sum(0)
    if inv_den is None:
        inv_den = nse_inv_den(obs)
    return (num * inv_den).sum(), sim.shape[1]


def nse_inv_den(obs):
    return 1.0 / (((obs - obs.mean(0, keepdim=True)) ** 2).sum(0) + 1e-6)


def synth(T, B, n_attr, dev, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    day = torch.arange(T, device=dev, dtype=torch.float32)[:, None]
    season = torch.sin(2 * torch.pi * day / 365.0)
    P = torch.clamp((torch.rand((T, B), generator=g, device=dev) - 0.7) * 60.0, min=0.0)
    Tm = 10 * season + 5 * torch.randn((T, B), generator=g, device=dev) \
        + torch.rand((1, B), generator=g, device=dev) * 25 - 10
    PET = torch.clamp(3 + 2.5 * season, min=0).expand(T, B)
    x = torch.stack([P, Tm, PET], -1).contiguous()
    attrs = torch.randn((B, n_attr), generator=g, device=dev)
    return x, attrs


def make_trainer(dev, basins=100, rho=365, warm_up=365, nmul=16, hidden=256, lstm="fused", world=1, rank=0,
                 overlap_wanted=True, graph=False, capturable=False, tune_gemm=False):
    """The training step as a closure (bench.py times it as the `dpl` workload; main() below runs it):
    returns (step, info) with step() -> mean loss of the step.

    graph=True (one process, fused LSTM): the WHOLE step -- network, HBV forward, loss, backward, gradient scaling,
    Adam -- is captured once into one HIP graph and step() is a replay plus the read-back of the loss.  Three eager
    steps run first (lazy initialisation, Adam's state); their losses are info["eager_losses"], so that the sequence
    eager_losses + [step() ...] is the loss curve of the same training run.  Adam runs with capturable=True (its
    step count lives on the device); capturable=True alone selects that optimiser for an eager run to compare with.
    tune_gemm=True with graph=True: torch's TunableOp picks the library solution of every GEMM shape during the three
    eager steps (the output layer's 194-wide GEMMs run 2.4x faster than the default pick); the capture records those
    picks and TunableOp is switched off again -- inside a graph the choice costs the host nothing."""
    T, B, M = warm_up + rho, basins, nmul
    dyn = ["parBETA", "parBETAET"]
    cfg = {"nmul": M, "warm_up": warm_up, "dynamic_params": {"Hbv": dyn}}
    Hbv = hydrodl2_amd.load_model("hbv", "Hbv")
    phy = Hbv(cfg, dev)
    ny = phy.learnable_param_count
    n_attr = 8

    # global synthetic data set, identical on every rank; each rank keeps its block of basins
    x_all, attrs_all = synth(T, B, n_attr, dev, seed=0)
    torch.manual_seed(1)
    truth = ParamNet(3 + n_attr, 32, ny).to(dev)
    b0, b1 = sharding.basin_range(B, world, rank)
    x, attrs = x_all[:, b0:b1].contiguous(), attrs_all[b0:b1]
    mean, std = x_all.mean((0, 1)), x_all.std((0, 1)) + 1e-6
    z = torch.cat([(x - mean) / std, attrs[None].expand(T, -1, -1)], -1)
    with torch.no_grad():
        obs = Hbv(cfg, dev)({"x_phy": x}, truth(z))["streamflow"][:, :, 0]
        inv_den = nse_inv_den(obs)

    torch.manual_seed(2)                       # same initial network on every rank
    net = ParamNet(3 + n_attr, hidden, ny, fused=lstm == "fused").to(dev)
    if graph and (world > 1 or lstm != "fused" or dev.type != "cuda"):
        raise ValueError("graph=True: one process on the GPU with the fused LSTM (the all-reduce is not captured)")
    # on the GPU the fused Adam: one kernel for all parameters instead of ~30 (each node of a captured step costs ~5 us)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=bool(graph or capturable),
                           fused=True if dev.type == "cuda" else None)
    params = [p for p in net.parameters()]

    # Backward produces gradients output-layer-first: net.out right after the HBV adjoint, then the
    # LSTM (the expensive part), then net.inp.  With --overlap (default) the first bucket -- loss
    # normalisers + net.out -- is all-reduced asynchronously from a gradient hook while the LSTM backward
    # runs; the second bucket follows when backward returns.
    overlap = world > 1 and overlap_wanted
    early = list(net.out.parameters())
    late = [p for p in params if all(p is not q for q in early)]
    pending = {}

    def on_grad(_p):
        if "bucket" not in pending and all(q.grad is not None for q in early):
            pending["bucket"] = sharding.AsyncBucket([pending["stats"]] + [q.grad for q in early]).start()

    if overlap:
        for q in early:
            q.register_post_accumulate_grad_hook(on_grad)

    def step():
        opt.zero_grad(set_to_none=True)
        raw = net(z)
        sim = phy({"x_phy": x}, raw)["streamflow"][:, :, 0]
        loss_sum, count = nse_loss(sim, obs, inv_den)
        stats = torch.stack([loss_sum.detach(), torch.tensor(float(count), device=dev)])
        pending.clear()
        pending["stats"] = stats
        loss_sum.backward()
        if overlap:
            second = sharding.AsyncBucket([p.grad for p in late]).start()
            pending["bucket"].finish()
            second.finish()
        else:
            # one bucketed all-reduce: [loss sum, basin count, every network gradient]
            sharding.all_reduce_sum_([stats] + [p.grad for p in params])
        torch._foreach_div_([p.grad for p in params], stats[1])
        opt.step()
        return float(stats[0] / stats[1])

    info = {"T": T, "B": B, "M": M, "ny": ny, "overlap": overlap, "graph": bool(graph)}
    if not graph:
        return step, info

    count_t = torch.tensor(float(b1 - b0), device=dev)       # the loss normaliser: a constant of the capture

    def body():
        raw = net(z)
        sim = phy({"x_phy": x}, raw)["streamflow"][:, :, 0]
        loss_sum, _ = nse_loss(sim, obs, inv_den)
        loss_sum.backward()
        torch._foreach_div_([p.grad for p in params], count_t)
        opt.step()
        return loss_sum.detach() / count_t

    if tune_gemm:
        enable_tunable_gemm(write_file=False)
    cur = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    side.wait_stream(cur)
    eager_losses = []
    with torch.cuda.stream(side):                              # warm-up off the capture, as torch's recipe asks
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            eager_losses.append(float(body()))
    cur.wait_stream(side)
    opt.zero_grad(set_to_none=True)                            # the captured backward creates the .grad tensors in the pool
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = body()
    # the capture pass enqueued nothing (a capture records), so the model is where the three eager steps left it
    if tune_gemm:
        import torch.cuda.tunable as tunable
        tunable.enable(False)
    info["eager_losses"] = eager_losses
    info["graph_handle"] = g

    def step_graphed():
        g.replay()
        return float(static_loss)

    return step_graphed, info


def enable_tunable_gemm(write_file=True):
    """torch's TunableOp: every new GEMM shape is timed against the library's solutions once (<= 1 s per shape)."""
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(1000)       # ms per GEMM shape
    # the results file TunableOp writes at exit: out of the working directory
    tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"),
                                      "tunableop_dpl.csv" if write_file else f"tunableop_dpl_{os.getpid()}.csv"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--basins", type=int, default=100)
    ap.add_argument("--rho", type=int, default=365)
    ap.add_argument("--warm-up", type=int, default=365)
    ap.add_argument("--nmul", type=int, default=16)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--lstm", choices=["fused", "torch"], default="fused",
                    help="fused: the HIP sequence kernels of include/hbvx_lstm.h; torch: torch.nn.LSTM")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one blocking all-reduce after the whole backward pass instead of sending the output "
                         "layer's gradients while the LSTM backward is still running")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step (network, HBV, loss, backward, Adam) into one HIP graph and replay it "
                         "(one process, fused LSTM)")
    ap.add_argument("--tune-gemm", action="store_true",
                    help="let torch's TunableOp pick the hipBLASLt/rocBLAS solution of every GEMM shape during "
                         "the warm-up steps (the LSTM's weight-gradient GEMMs have K = T*B: the default "
                         "heuristic is 2x off there)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=None,
                    help="process-group backend (default: nccl = RCCL on the GPU, gloo on the CPU)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="every rank uses cuda:0 (rehearsal of the multi-rank schedule on a one-GPU box; needs --backend gloo)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    on_gpu = args.device.startswith("cuda")
    if on_gpu:
        torch.cuda.set_device(0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0")))
    dev = torch.device(args.device if not on_gpu else f"cuda:{torch.cuda.current_device()}")
    if world > 1:
        dist.init_process_group(args.backend or ("nccl" if on_gpu else "gloo"))

    if args.tune_gemm and on_gpu and not args.graph:
        enable_tunable_gemm()

    step, info = make_trainer(dev, args.basins, args.rho, args.warm_up, args.nmul, args.hidden, args.lstm, world, rank,
                              not args.no_overlap, graph=args.graph, tune_gemm=args.tune_gemm and args.graph)
    T, B, M, overlap = info["T"], info["B"], info["M"], info["overlap"]
    losses = list(info.get("eager_losses", [])) + [step() for _ in range(3)]        # warm-up
    ops.KERNEL_EVENTS = [] if on_gpu else None
    if on_gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(step())
    if on_gpu:
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    hbv_ms = lstm_ms = None
    if on_gpu:
        ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
        hbv_ms = sum(e0.elapsed_time(e1) for n, e0, e1 in ev if "lstm" not in n) / args.steps
        lstm_ms = sum(e0.elapsed_time(e1) for n, e0, e1 in ev if "lstm" in n) / args.steps
    if rank == 0:
        print(json.dumps({"basins": B, "nmul": M, "days": T, "world": world, "ms_per_step": round(dt * 1e3, 3),
                          "lstm": args.lstm, "hidden": args.hidden, "tuned_gemm": bool(args.tune_gemm),
                          "graph": bool(args.graph),
                          "allreduce": "overlapped (2 buckets)" if overlap else "blocking (1 bucket)",
                          "hbv_calls_ms": None if hbv_ms is None else round(hbv_ms, 3),
                          "lstm_kernels_ms": None if not lstm_ms else round(lstm_ms, 3),
                          "loss_first": round(losses[0], 4), "loss_last": round(losses[-1], 4)}))
    if world > 1:
        dist.destroy_process_group()
    return losses


if __name__ == "__main__":
    main()
