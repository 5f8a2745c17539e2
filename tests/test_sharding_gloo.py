"""CPU tier: the N>1 path (basin sharding + one bucketed all-reduce) with world_size 2 over gloo,
driving the oracle backend; compared with the unsharded run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from . import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(T, B, M, seed):
    x = torch.from_numpy(synth.forcing(T, B, seed))
    p = torch.from_numpy(synth.raw_parameters(T, B, 13 * M + 2, seed))
    w = torch.from_numpy(synth.loss_weights((T, B, 1), seed, 50))
    return x, p, w


def _run(model_cfg, x, p, w):
    import hydrodl2_amd
    Hbv = hydrodl2_amd.load_model("hbv", "Hbv")
    m = Hbv(model_cfg, torch.device("cpu"))
    p = p.clone().requires_grad_(True)
    out = m({"x_phy": x}, p)
    loss = (out["streamflow"] * w).sum()
    loss.backward()
    return out, loss.detach(), p.grad


def _worker(rank, world, port, oracle, T, B, M, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from hydrodl2_amd import sharding
    from tests import seam
    seam.use_library(oracle)
    cfg = {"nmul": M, "dynamic_params": {"Hbv": ["parBETA", "parBETAET"]}, "warm_up": 5}
    x, p, w = _make(T, B, M, seed)
    xs, ps = sharding.shard_inputs({"x_phy": x}, p, world, rank)
    b0, b1 = sharding.basin_range(B, world, rank)
    out, loss, grad = _run(cfg, xs["x_phy"], ps, w[5:, b0:b1])
    # what a shared parameterisation network would receive: basin-summed gradient rows
    shared = grad.sum(1)
    lossv = loss.reshape(1).clone()
    sharding.all_reduce_sum_([shared, lossv])
    full = sharding.gather_flux_dict({k: v.detach() for k, v in out.items()}, B)
    if rank == 0:
        q.put((shared.numpy(), lossv.numpy(), {k: v.numpy() for k, v in full.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [10, 7])  # even and uneven shards
def test_two_rank_shard_matches_single_process(B, oracle_path, oracle_backend):
    T, M, seed = 40, 4, 61
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, oracle_path, T, B, M, seed, q))
             for r in range(2)]
    for p_ in procs:
        p_.start()
    shared, lossv, full = q.get(timeout=120)
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0

    cfg = {"nmul": M, "dynamic_params": {"Hbv": ["parBETA", "parBETAET"]}, "warm_up": 5}
    x, p, w = _make(T, B, M, seed)
    out, loss, grad = _run(cfg, x, p, w[5:])
    for k, v in out.items():
        np.testing.assert_array_equal(full[k], v.detach().numpy(), err_msg=k)  # basins independent
    np.testing.assert_allclose(lossv[0], float(loss), rtol=1e-5)
    ref = grad.sum(1).numpy()
    np.testing.assert_allclose(shared, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())


def test_basin_range_covers_everything():
    from hydrodl2_amd.sharding import basin_range
    for B in (1, 7, 671, 100000):
        for world in (1, 2, 3, 8):
            spans = [basin_range(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
