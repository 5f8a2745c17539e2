"""Basin sharding across GPUs: one process per GPU, torch.distributed (backend "nccl" = RCCL).

The HBV path has no term that couples basins (every operation in the reference's loop is
element-wise over [basin, member]; routing is per basin: hbv.py:423-553, uh_routing.py:47-50),
so the data path needs NO collective: each rank runs the kernels on its contiguous block of
basins.  The only exchange is one level up -- summing, across ranks, the loss and whatever
gradient a shared parameterisation network receives -- and, optionally, gathering outputs on
the basin axis.  This module provides exactly that and nothing more.
"""
from __future__ import annotations

from typing import Iterable, Optional, Sequence

import torch
import torch.distributed as dist


def basin_range(n_basins: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block [b0, b1) of basins owned by `rank` (ceil split; last blocks may be short)."""
    per = (n_basins + world - 1) // world
    b0 = min(rank * per, n_basins)
    return b0, min(b0 + per, n_basins)


# x_dict entries by NAME and the axis their basins sit on (the reference's layouts: hbv.py:284-300, hbv_2.py:324-345).
# muwts is [T,B,nmul], [1,B,nmul] (axis 1) or [B,nmul] (axis 0).  A key that is not listed is an error, not a guess:
# round 4's `shape[1] == B` heuristic mis-sliced a [B,n] tensor whenever n happened to equal B.
_BASIN_AXIS = {'x_phy': 1, 'ac_all': 0, 'elev_all': 0, 'areas': 0}
# entries that couple basins and cannot be cut by a contiguous basin block (the hourly model's units -> gages
# topology and its per-pair routing parameters, hbv_2_hourly.py:819-850; SURVEY.md §8e: out of scope)
_COUPLED = ('outlet_topo', 'x_phy_high_freq', 'x_phy_low_freq')


def shard_inputs(x_dict: dict, parameters, world: int, rank: int):
    """Slice the basin axis of an `x_dict` / `parameters` pair as the reference lays them out:
    x_phy [T,B,3], muwts [T,B,nmul] | [1,B,nmul] | [B,nmul], ac_all / elev_all / areas [B]; parameters [T,B,ny] or
    the HBV 2.0 tuple ([T,B,*], [B,*]).  Raises on anything it cannot shard by a contiguous block of basins -- an
    unknown tensor key, the hourly model's `outlet_topo` / three-tensor tuple (units feed gages across blocks) --
    instead of passing it through or dropping it."""
    B = x_dict['x_phy'].shape[1]
    b0, b1 = basin_range(B, world, rank)
    xs = {}
    for k, v in x_dict.items():
        if not torch.is_tensor(v):
            xs[k] = v
            continue
        if k in _COUPLED:
            raise NotImplementedError(f"shard_inputs: '{k}' couples basins (gage routing / multi-timescale inputs); "
                                      "shard the units by gage upstream of this helper")
        if k == 'muwts':
            axis = 0 if v.dim() == 2 else 1
        elif k in _BASIN_AXIS:
            axis = _BASIN_AXIS[k]
        else:
            raise KeyError(f"shard_inputs: no basin axis known for x_dict['{k}'] (shape {tuple(v.shape)}); "
                           f"known keys: {sorted(_BASIN_AXIS) + ['muwts']}")
        if v.dim() <= axis or v.shape[axis] != B:
            raise ValueError(f"shard_inputs: x_dict['{k}'] has shape {tuple(v.shape)}, expected {B} basins on axis {axis}")
        xs[k] = v.narrow(axis, b0, b1 - b0).contiguous()
    if isinstance(parameters, (tuple, list)):
        if len(parameters) != 2:
            raise NotImplementedError(f"shard_inputs: a {len(parameters)}-tensor parameter tuple (the hourly model's "
                                      "per-pair routing parameters) cannot be cut by basin blocks")
        pd, pst = parameters
        if pd.dim() != 3 or pd.shape[1] != B or pst.dim() != 2 or pst.shape[0] != B:
            raise ValueError(f"shard_inputs: HBV 2.0 parameters must be ([T,{B},*], [{B},*]), got "
                             f"{tuple(pd.shape)}, {tuple(pst.shape)}")
        ps = (pd[:, b0:b1].contiguous(), pst[b0:b1].contiguous())
    else:
        if parameters.dim() != 3 or parameters.shape[1] != B:
            raise ValueError(f"shard_inputs: parameters must be [T,{B},ny], got {tuple(parameters.shape)}")
        ps = parameters[:, b0:b1].contiguous()
    return xs, ps


_BUCKETS: dict = {}     # (tag, device, dtype, numel) -> persistent flat buffer of a step's collective


def _persistent(tag: str, device, dtype, numel: int) -> torch.Tensor:
    """A flat buffer kept between steps.  The collectives sit on the tail of every training step; a fresh
    allocation there is a caching-allocator call (and, while the pool still grows, a hipMalloc) per step for a
    buffer whose size never changes.  `tag` keeps buffers that are alive at the same time apart (the blocking
    bucket, an AsyncBucket in flight, the all-gather's send and receive sides)."""
    key = (tag, device, dtype, numel)
    buf = _BUCKETS.get(key)
    if buf is None:
        if len(_BUCKETS) > 32:
            _BUCKETS.clear()
        buf = _BUCKETS[key] = torch.empty(numel, dtype=dtype, device=device)
    return buf


def _bucket(tensors: Sequence[torch.Tensor], tag: str = "sync") -> torch.Tensor:
    """Flat staging buffer for `tensors`, kept between steps (see _persistent), filled with their values."""
    t0 = tensors[0]
    n = sum(t.numel() for t in tensors)
    flat = _persistent(tag, t0.device, t0.dtype, n)
    off = 0
    for t in tensors:
        k = t.numel()
        flat[off:off + k].copy_(t.reshape(-1))
        off += k
    return flat


def all_reduce_sum_(tensors: Sequence[torch.Tensor], group=None) -> None:
    """Sum `tensors` over ranks in ONE bucketed all-reduce (flatten -> all_reduce -> scatter back).

    xGMI is point-to-point and a ring all-reduce of a small bucket is latency-bound, so the
    loss scalar(s) and the shared-network gradient travel together in a single collective.  A single tensor is
    reduced in place (no staging at all)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    if len(tensors) == 1 and tensors[0].is_contiguous():
        dist.all_reduce(tensors[0], op=dist.ReduceOp.SUM, group=group)
        return
    flat = _bucket(tensors)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


class AsyncBucket:
    """One bucket of tensors summed over ranks WHILE the caller keeps computing.

    `start()` flattens the tensors and issues `all_reduce(async_op=True)` (RCCL runs it on its own
    stream; on xGMI a few-MB bucket is latency-bound, tens of microseconds, so it hides completely behind
    any kernel that is still running); `finish()` waits and scatters the sums back.  Used to send the
    gradients that the backward pass produces FIRST (the output layer of the parameterisation network,
    right after the HBV adjoint) while the part that is produced last (the LSTM) is still being
    computed -- the overlap SURVEY.md §8f rank 4 asks for."""

    # Staging slots in flight.  A bucket takes the lowest FREE slot in start() and gives it back in finish() (also
    # when the wait raises): with a plain counter, start A, start B, finish A, start C handed C the slot -- and with
    # the same size the very buffer -- of B while B's all-reduce was still in flight.
    _busy: set = set()

    def __init__(self, tensors: Sequence[torch.Tensor], group=None):
        self.tensors, self.group = list(tensors), group
        self.flat = self.work = None
        self.slot = None

    def start(self) -> "AsyncBucket":
        if self.work is not None:
            raise RuntimeError("AsyncBucket.start() called twice without finish()")
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            # the staging buffer persists between steps like the blocking path's (one per bucket in flight: the
            # step's early and late buckets overlap in time)
            slot = 0
            while slot in AsyncBucket._busy:
                slot += 1
            AsyncBucket._busy.add(slot)
            self.slot = slot
            try:
                self.flat = _bucket(self.tensors, tag=f"async{slot}")
                self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            except BaseException:
                self._release()
                raise
        return self

    def _release(self) -> None:
        if self.slot is not None:
            AsyncBucket._busy.discard(self.slot)
        self.slot = self.flat = self.work = None

    def finish(self) -> None:
        if self.work is None:
            return
        try:
            self.work.wait()
            off = 0
            for t in self.tensors:
                n = t.numel()
                t.copy_(self.flat[off:off + n].view_as(t))
                off += n
        finally:
            self._release()


def gather_basins(local: torch.Tensor, n_basins: int, dim: int, group=None) -> torch.Tensor:
    """All-gather a per-shard tensor along its basin axis (shards may be uneven)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    per = (n_basins + world - 1) // world
    pad_shape = list(local.shape)
    pad_shape[dim] = per
    n = 1
    for v in pad_shape:
        n *= v
    # send and receive sides persist between calls (one flat receive buffer for all ranks, all_gather_into_tensor:
    # no per-call list of `world` padded tensors); only the result is a fresh tensor
    send = _persistent("gather_send", local.device, local.dtype, n).view(pad_shape)
    have = local.shape[dim]
    send.narrow(dim, 0, have).copy_(local)
    if have < per:
        send.narrow(dim, have, per - have).zero_()
    recv = _persistent("gather_recv", local.device, local.dtype, n * world)
    dist.all_gather_into_tensor(recv, send.reshape(-1), group=group)
    parts = recv.view([world] + pad_shape)
    out = []
    for r in range(world):
        b0, b1 = basin_range(n_basins, world, r)
        out.append(parts[r].narrow(dim, 0, b1 - b0))
    return torch.cat(out, dim=dim)


def gather_flux_dict(flux: dict, n_basins: int, group=None) -> dict:
    """Gather a flux dictionary ([T,b,1] series, BFI [b]) to full-basin tensors on every rank."""
    return {k: gather_basins(v, n_basins, 0 if v.dim() == 1 else 1, group) for k, v in flux.items()}
