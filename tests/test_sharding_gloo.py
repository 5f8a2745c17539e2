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


def test_shard_inputs_slices_by_name_and_refuses_what_it_cannot_cut():
    """Named keys, not shapes: a [B, n] tensor with n == B is cut on axis 0 (round 4's `shape[1] == B` test cut it on
    axis 1); the hourly model's three-tensor tuple / outlet_topo and unknown tensor keys raise instead of being
    dropped or passed through whole."""
    from hydrodl2_amd.sharding import shard_inputs
    T, B, M = 6, 8, 8          # nmul == B on purpose
    x = {"x_phy": torch.arange(T * B * 3.0).reshape(T, B, 3), "muwts": torch.arange(B * M * 1.0).reshape(B, M),
         "ac_all": torch.arange(B * 1.0), "elev_all": torch.arange(B * 1.0) + 100, "note": "not a tensor"}
    pd, pst = torch.arange(T * B * 2.0).reshape(T, B, 2), torch.arange(B * B * 1.0).reshape(B, B)
    xs, (pd1, ps1) = shard_inputs(x, (pd, pst), 3, 1)         # ceil split: 3 + 3 + 2
    assert torch.equal(xs["x_phy"], x["x_phy"][:, 3:6]) and torch.equal(xs["muwts"], x["muwts"][3:6])
    assert torch.equal(xs["ac_all"], x["ac_all"][3:6]) and torch.equal(xs["elev_all"], x["elev_all"][3:6])
    assert xs["note"] == "not a tensor"
    assert torch.equal(pd1, pd[:, 3:6]) and torch.equal(ps1, pst[3:6]) and ps1.shape == (3, B)
    xs, p1 = shard_inputs({"x_phy": x["x_phy"], "muwts": torch.ones(1, B, M)}, torch.zeros(T, B, 5), 3, 2)
    assert xs["muwts"].shape == (1, 2, M) and p1.shape == (T, 2, 5)
    with pytest.raises(NotImplementedError, match="3-tensor"):
        shard_inputs(x, (pd, pst, torch.zeros(4, 3)), 2, 0)
    with pytest.raises(NotImplementedError, match="outlet_topo"):
        shard_inputs(dict(x, outlet_topo=torch.ones(2, B)), (pd, pst), 2, 0)
    with pytest.raises(KeyError, match="obs"):
        shard_inputs(dict(x, obs=torch.ones(T, B)), (pd, pst), 2, 0)
    with pytest.raises(ValueError, match="ac_all"):
        shard_inputs(dict(x, ac_all=torch.ones(B + 1)), (pd, pst), 2, 0)
    with pytest.raises(ValueError, match="parameters"):
        shard_inputs(x, torch.zeros(T, B + 1, 4), 2, 0)


def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hydrodl2_amd.sharding import AsyncBucket
    n = 1000
    a, b, c = (torch.full((n,), float(v + rank)) for v in (1, 10, 100))
    A, Bk, Ck = AsyncBucket([a]), AsyncBucket([b]), AsyncBucket([c])
    A.start()
    Bk.start()
    A.finish()
    Ck.start()            # same size as B, B still in flight: must not land in B's staging buffer
    slots = (Bk.slot, Ck.slot)
    Bk.finish()
    Ck.finish()
    if rank == 0:
        q.put((a[0].item(), b[0].item(), c[0].item(), slots, sorted(AsyncBucket._busy)))
    dist.barrier()
    dist.destroy_process_group()


def test_async_buckets_in_flight_never_share_a_staging_buffer():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    a, b, c, slots, busy = q.get(timeout=120)
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    assert (a, b, c) == (1 + 2, 10 + 11, 100 + 101)
    assert slots[0] != slots[1] and busy == []
