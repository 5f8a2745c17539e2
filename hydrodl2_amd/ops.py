"""torch.autograd shim over the C ABI (include/hbvx.h).

One autograd.Function covers the whole hot path of one `_PBM` call
(reference: hbv.py:363-553): parameter prep + daily recurrence + ensemble mean
(`hbvx_forward`) followed by unit-hydrograph routing of the four runoff series
(`hbvx_route_forward`).  Keeping both stages in one Function means the
gradient w.r.t. the raw NN output is written into ONE buffer (the routing
columns and the physical columns live in the same `parameters` tensor).

PyTorch is used for device memory, the current stream and autograd plumbing
only; every arithmetic step of the path runs in the HIP library.
"""
from __future__ import annotations

import ctypes as _C
from dataclasses import dataclass, field
import functools
import os
import struct
from typing import List, NamedTuple, Optional, Sequence

import torch

from . import _abi
from ._lib import get_library


@dataclass
class ParamSource:
    """Where physical-parameter slot `slot` lives inside the input tensors.

    Offsets/strides are in float32 elements relative to tensor `tensor_idx`'s
    first element; `dyn_off` addresses (t=0 of this call, b=0, j=0).
    """
    slot: int
    lo: float
    hi: float
    tensor_idx: int
    sta_off: int
    sta_bs: int
    dyn_tensor_idx: int = -1
    dyn_off: int = -1          # -1: static parameter
    dyn_ts: int = 0
    dyn_bs: int = 0
    drop: Optional[torch.Tensor] = None  # uint8 [B] on the compute device


@dataclass
class RouteSource:
    tensor_idx: int
    a_off: int
    b_off: int
    stride: int
    a_bounds: Sequence[float]
    b_bounds: Sequence[float]


@dataclass
class StepConfig:
    model: int
    n_param: int
    n_flux: int
    T: int                     # steps of this call
    t0: int                    # first row of x used by this call
    B: int
    M: int
    raw_sigmoid: bool
    channels: Sequence[int]    # (prcp, tmean, pet) channel indices in x
    nearzero: float
    params: List[ParamSource] = field(default_factory=list)
    route: Optional[RouteSource] = None
    want_flux: bool = True
    want_traj: bool = False    # keep the state trajectory even without autograd
    adj_gtol: float = 1e-3     # implicit scheme only (hbv_adj.py:519)
    adj_max_iter: int = 3      # implicit scheme only (hbv_adj.py:518)
    adj_stop: int = 0          # implicit scheme only: 0 per-lane stopping rule, 1 per wavefront (hbv_adj.py:544)
    mu_t0: Optional[int] = None  # first row of muwts used by this call (None: same as t0)
    want_bfi: bool = False     # also return BFI (hbv.py:562-567) from the same autograd node (no second Function)
    ckpt_days: int = 0         # 4 / 8 / 16: keep K-day checkpoints instead of the trajectory (memory-lean adjoint)
    persistent_grad: bool = False   # module key grad_buffer='persistent': see _overlapped_grad_buffers


def _device_guard(fn):
    """Run an autograd.Function method with the CUDA device of its tensors current.  The library takes
    the stream from the tensors' device, but HIP calls it makes besides the launch (function attributes
    for dynamic LDS, device properties) go to the calling thread's CURRENT device: a model living on
    cuda:k in a process that never called torch.cuda.set_device(k) would otherwise mix devices."""
    @functools.wraps(fn)
    def wrapped(ctx, *args):
        dev = next((a.device for a in args if torch.is_tensor(a) and a.is_cuda), None)
        if dev is None or torch.cuda.current_device() == dev.index:   # the common case: nothing to switch
            return fn(ctx, *args)
        with torch.cuda.device(dev):
            return fn(ctx, *args)
    return wrapped


# bench.py sets this to a list to collect (abi_call, start_event, end_event) per launch,
# recorded on the stream the kernels are enqueued on.
KERNEL_EVENTS = None


def _call(lib, name, fn, *args):
    if KERNEL_EVENTS is None or not lib.is_device:
        return fn(*args)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(*args)
    e1.record()
    KERNEL_EVENTS.append((name, e0, e1))


_POISON = os.environ.get("HBVX_DEBUG_POISON", "") not in ("", "0")


def _out(shape, device) -> torch.Tensor:
    """Output buffer the library overwrites completely.  HBVX_DEBUG_POISON=1 (set by the GPU
    test tier) fills it with NaN first, so a kernel that skips an element cannot pass on stale
    memory the caching allocator handed back."""
    if _POISON:
        return torch.full(shape, float("nan"), dtype=torch.float32, device=device)
    return torch.empty(shape, dtype=torch.float32, device=device)


def _zeros_like(lib, t: torch.Tensor) -> torch.Tensor:
    """Dense zero gradient shaped like `t`, filled through hbvx_zero (6.6 TB/s on MI355X against 5.3 for
    torch's fill kernel: the [T,B,ny] gradient of config 2 is 3.8 GB per step)."""
    out = torch.empty_like(t)           # dense like t (same strides when t is dense): numel elements of storage
    if out.numel():
        _call(lib, 'hbvx_zero', lib.zero, out.data_ptr(), out.numel() * out.element_size(), _stream_of(lib, out))
    return out


def _dyn_columns(p: torch.Tensor, cfg: "StepConfig", tensor_idx: int):
    """Which elements of the gradient of parameter tensor `p` the adjoint STORES (never accumulates): the nmul
    columns of every dynamic parameter on the days of this call (hbvx_backward writes each of those elements
    exactly once, 0 for a dy_drop-masked basin).  Returns (keep bit mask over column groups of width nmul,
    first day) for hbvx_zero_except, (0, 0) when the tensor holds no dynamic parameter, or None when its layout
    is not the plain [T,B,width] one (then only a dense fill is safe)."""
    dyn = [ps for ps in cfg.params if ps.dyn_off >= 0 and ps.dyn_tensor_idx == tensor_idx]
    if not dyn:
        return 0, 0
    M, B, T = cfg.M, cfg.B, cfg.T
    if p.dim() != 3 or not p.is_contiguous() or p.shape[1] != B:
        return None
    W = p.shape[2]
    keep, t_first = 0, None
    for ps in dyn:
        tf, rem = divmod(ps.dyn_off, B * W)
        k, j0 = divmod(rem, M)
        if (ps.dyn_ts != B * W or ps.dyn_bs != W or rem >= W or j0 != 0 or (k + 1) * M > W or k >= 32
                or (t_first is not None and tf != t_first) or tf + T > p.shape[0]):
            return None
        t_first = tf
        keep |= 1 << k
    return keep, t_first


def _fill_grad(lib, out: torch.Tensor, cfg: "StepConfig", cols) -> None:
    """Zero `out` (a [T,B,W] gradient buffer) except the columns the adjoint stores (`cols` from _dyn_columns)."""
    keep, t_first = cols
    B, W = out.shape[1], out.shape[2]
    _call(lib, 'hbvx_zero', lib.zero_except, out.data_ptr(), out.shape[0] * B, W, t_first * B, (t_first + cfg.T) * B,
          cfg.M, keep, _stream_of(lib, out))


def _grad_like(lib, p: torch.Tensor, cfg: "StepConfig", tensor_idx: int) -> torch.Tensor:
    """Gradient buffer for parameter tensor `p`: zero wherever the adjoint accumulates (static rows,
    routing columns) or never writes, left alone where it STORES (_dyn_columns).  Falls back to the dense fill
    when the layout is not the plain [T,B,width] one.  Under HBVX_DEBUG_POISON the left-alone part is NaN: an
    element a kernel skipped cannot pass."""
    cols = _dyn_columns(p, cfg, tensor_idx)
    if cols is None or cols[0] == 0:
        return _zeros_like(lib, p)
    out = torch.full_like(p, float("nan")) if _POISON else torch.empty_like(p)
    _fill_grad(lib, out, cfg, cols)
    return out


# Part of the dense zero fill of a [T,B,ny] gradient (3.8 GB at config 2) INSIDE THE FORWARD LAUNCH (hbvx_fwd_out.zero_ptr,
# ABI 10): the pipelined forward is a latency chain on 168 of 256 CUs; surplus workgroups of the same launch, on the CUs
# it leaves idle, write zeros into the buffer front to back for as long as the recurrence runs.  The buffer is allocated
# on the caller's stream in forward and handed to backward through the autograd context (first backward only: a second
# one over a retained graph fills its own); backward fills what is missing (hbvx_zero_rest) on the second stream beside
# the adjoint, as it fills the whole buffer without this.  One stream and one launch in forward: no cross-stream
# ownership (round 5's second-stream fill beside the forward cost a hipMalloc every other step: the caching allocator
# cannot recycle a block another stream still owns -- profiles/r05_earlyzero.txt).
# WHERE IT PAYS (profiles/r05_inlaunch_fill.txt): the implicit scheme, whose 4.2 ms forward (575 ns per day) does not
# notice the fill -- config 4: 6.75-6.84 -> 6.32-6.38 ms, the speed of grad_buffer='persistent' with a fresh gradient
# tensor per step.  The explicit models' recurrence (115 ns per day, a tile of ten days in flight) is sensitive to ANY
# concurrent HBM writes: 22 fill waves (1 TB/s) stretch config 2's forward from 0.90 to 1.08 ms, 88 waves to 1.25 --
# what the adjoint gains (1.33 -> 0.92) the forward loses (2.37-2.55 -> 2.22-2.46 ms over three boxes, config 2 with two
# dynamic parameters +-0).  So: "auto" = the implicit scheme only; HBVX_EARLY_ZERO=1 all models, 0 none.
_EARLY_ZERO = os.environ.get("HBVX_EARLY_ZERO", "auto")
_EARLY_ZERO_MIN = 1 << 26     # elements: below 256 MB the fill is not worth a special path
_SIDE_STREAMS: dict = {}


def _row_split(p: torch.Tensor, cfg: "StepConfig", i: int):
    """Whether parameter tensor i takes the big-buffer + separate-last-row form of its gradient: (ok, base, cols).
    It does when it is large, [T,B,W]-shaped, every static / routing source of it lies in its last row, and at most half
    of its columns belong to dynamic parameters of this call."""
    base = (p.shape[0] - 1) * p[0].numel() if p.dim() == 3 else -1
    offs = [ps.sta_off for ps in cfg.params if ps.tensor_idx == i]
    if cfg.route is not None and cfg.route.tensor_idx == i:
        offs += [cfg.route.a_off, cfg.route.b_off]
    cols = _dyn_columns(p, cfg, i)
    few_dyn = cols is not None and bin(cols[0]).count("1") * cfg.M * 2 <= p.shape[-1]
    ok = (p.dim() == 3 and p.is_contiguous() and p.numel() >= _EARLY_ZERO_MIN and p.shape[0] > 1
          and all(o >= base for o in offs) and cols is not None and (cols[0] == 0 or few_dyn))
    return ok, base, cols


def _early_zero_request(lib, cfg: "StepConfig", ptensors, needs, out) -> Optional[tuple]:
    """Forward side: allocate the gradient buffer of the (one) qualifying parameter tensor and offer it to the forward
    launch (out.zero_ptr / zero_bytes / zero_state).  Returns (tensor index, buffer, state words) for the autograd
    context, or None.  A configuration whose forward carries no fill workgroups (another kernel family, no idle CUs)
    stops offering after its first call: holding the buffer from forward to backward would buy nothing."""
    wanted = _EARLY_ZERO == "1" or (_EARLY_ZERO not in ("", "0") and cfg.model == _abi.MODEL_HBVADJ)
    if not (wanted and lib.is_device and not cfg.persistent_grad and _FILL_OVERLAP):
        return None
    if cfg.__dict__.get("_early_zero_useless"):
        return None
    for i, (p, need) in enumerate(zip(ptensors, needs)):
        if need and _row_split(p, cfg, i)[0]:
            big = torch.empty_like(p)
            state = torch.zeros(2, dtype=torch.int32, device=p.device)
            out.zero_ptr, out.zero_bytes, out.zero_state = big.data_ptr(), big.numel() * big.element_size(), state.data_ptr()
            return i, big, state
    return None


def _early_zero_result(lib, cfg: "StepConfig", early) -> None:
    if early is not None and not lib.zero_in_launch():
        cfg.__dict__["_early_zero_useless"] = True


# Default: the fill runs BESIDE the adjoint.  The adjoint kernels are bound by instruction issue, the
# fill by HBM bandwidth, and the only thing that ties them is the static row: static-parameter and routing
# gradients accumulate into the LAST row of the [T,B,ny] gradient.  So those go to a separate [B,ny] row
# (the C ABI takes any pointer + stride for them), the big buffer is filled on a second HIP stream while
# the adjoint runs on the caller's, and one small add joins them at the end.  Dynamic rows are written
# (not accumulated) by the adjoint and skipped by hbvx_zero_except: disjoint bytes, no ordering needed.
# HBVX_FILL_OVERLAP=0 restores the fill in front of the adjoint.
_FILL_OVERLAP = os.environ.get("HBVX_FILL_OVERLAP", "1") not in ("", "0")


def _side_stream(dev) -> "torch.cuda.Stream":
    side = _SIDE_STREAMS.get(dev.index)
    if side is None:
        side = _SIDE_STREAMS[dev.index] = torch.cuda.Stream(dev)
    return side


def _overlapped_grad_buffers(lib, cfg: "StepConfig", ptensors, needs, early=None):
    """(gp, rows, bases, start): gradient buffers, the separate last rows (None where a tensor keeps the
    plain path), their element offsets inside the big tensors, and `start()` -> event: begins the fills
    of the qualifying tensors on the side stream (call it when the caller's stream has nothing
    bandwidth-bound left in front of the adjoint) and returns the event that marks them complete;
    `start.gated`: that event must reach the library as `store_gate`.  Which tensors qualify: _row_split.
    `early`: (index, buffer) from the forward (_early_zero_request): that tensor's buffer is already zero."""
    dev = ptensors[0].device
    gp, rows, bases, todo = [], [], [], []
    gated = False
    for i, (p, need) in enumerate(zip(ptensors, needs)):
        if not need:
            gp.append(None); rows.append(None); bases.append(0)
            continue
        # A tensor with dynamic columns is filled DENSELY (the fastest fill) and the adjoint's stores into those
        # columns must land after it: the fill's event goes to the library as hbvx_bwd_io.store_gate, which holds
        # back the storing kernel only (the transfer-map and scan passes of the time-parallel adjoint run beside
        # the fill).  Without the gate the two race (measured: wrong gradients).  Worth it while few columns are
        # dynamic -- with most of them dynamic (config 3) hbvx_zero_except in front of the adjoint writes a
        # fraction of the bytes and stays.
        ok, base, cols = _row_split(p, cfg, i)
        if early is not None and early[0] == i and ok:
            # its front part was zeroed inside the forward's launch (_early_zero_request); what is missing is filled below,
            # beside the adjoint, like a whole buffer
            gp.append(early[1])
            rows.append(torch.zeros_like(p[0]))
            bases.append(base)
            todo.append((early[1], early[2]))
            gated = gated or cols[0] != 0
            continue
        if not ok:
            gp.append(_grad_like(lib, p, cfg, i)); rows.append(None); bases.append(0)
            continue
        if cfg.persistent_grad:
            # grad_buffer='persistent' (opt-in): the [T,B,W] gradient lives in a buffer this configuration keeps.  All
            # of it but the last row and the dynamic columns is zero and STAYS zero (nothing ever writes there), the
            # dynamic columns are overwritten by every adjoint, the last row is reset below -- so the dense zero fill
            # the autograd contract costs (3.8 GB per step at config 2, and the bandwidth it takes from the adjoint
            # beside it) happens once.  The contract of a captured graph's static outputs: the tensor handed to
            # autograd is valid until this module's next backward with the same shapes (no gradient accumulation across
            # calls on a leaf).  The memo keeps the STORAGE and hands out a fresh tensor each time, so that autograd's
            # AccumulateGrad may adopt it (a tensor someone else references would be cloned: 3.8 GB).
            memo = cfg.__dict__.setdefault("_memo", {})
            key = ("pgrad", i, tuple(p.shape), str(p.device))
            st = memo.get(key)
            if st is None:
                big = _zeros_like(lib, p)
                memo[key] = big.untyped_storage()
            else:
                big = torch.empty(0, dtype=p.dtype, device=p.device).set_(st, 0, p.shape)
                big[-1].zero_()
            gp.append(big)
            rows.append(torch.zeros_like(p[0]))
            bases.append(base)
            continue
        # on the caller's stream and pool: it is also where it dies
        big = torch.empty_like(p)
        gp.append(big)
        rows.append(torch.zeros_like(p[0]))
        bases.append(base)
        todo.append((big, None))
        gated = gated or cols[0] != 0

    def start():
        if not todo:
            return None
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for big, state in todo:
                if state is None:
                    _call(lib, 'hbvx_zero', lib.zero, big.data_ptr(), big.numel() * big.element_size(), _stream_of(lib, big))
                else:
                    _call(lib, 'hbvx_zero', lib.zero_rest, big.data_ptr(), big.numel() * big.element_size(), state.data_ptr(),
                          _stream_of(lib, big))
            done = torch.cuda.Event()
            done.record(side)
        return done
    start.gated = gated      # the fill covers columns the adjoint stores: pass the event as store_gate
    return gp, rows, bases, start


def _ptr(t: Optional[torch.Tensor], off: int = 0) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr() + 4 * off


def _stream_of(lib, t: torch.Tensor) -> int:
    if lib.is_device:
        return torch.cuda.current_stream(t.device).cuda_stream
    return 0


def _check_tensor(lib, t: torch.Tensor, name: str):
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if lib.is_device and not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); "
                           "hydrodl2_amd has no CPU path")
    if not lib.is_device and t.is_cuda:
        raise RuntimeError(f"{name}: test library expects host tensors")


def _mu_t0(cfg: StepConfig) -> int:
    return cfg.t0 if cfg.mu_t0 is None else cfg.mu_t0


# ---------------------------------------------------------------------------------------------
# Descriptors.  A StepConfig that a module keeps across calls (same shapes, same dynamic set) carries a
# plan: the hbvx_desc bytes with every shape / bound / stride field filled in once, and the byte offsets of
# the pointer fields.  A call copies the template (1.3 KB) and packs the pointers in place -- the step of the
# deltaMG minibatch shape spent more host time marshalling descriptors than the GPU spent on its kernels.
_PTR = struct.Struct("<Q")
_D = _abi.Desc
_PS_SIZE = _C.sizeof(_abi.ParamSrc)
_OFF_P = _D.p.offset
_OFF_DYN, _OFF_STA, _OFF_DROP = _abi.ParamSrc.dyn.offset, _abi.ParamSrc.sta.offset, _abi.ParamSrc.drop.offset


class _DescPlan:
    __slots__ = ("key", "template", "sta", "dyn", "drop")


def _desc_plan(cfg: StepConfig, x, has_mu: bool) -> _DescPlan:
    key = (x.stride(0), x.stride(1), has_mu, len(cfg.params), cfg.T, cfg.t0, cfg.B, cfg.M)
    plan = getattr(cfg, "_plan", None)
    if plan is not None and plan.key == key:
        return plan
    d = _abi.Desc()
    d.abi_version = _abi.ABI_VERSION
    d.model = cfg.model
    d.T, d.B, d.M = cfg.T, cfg.B, cfg.M
    d.n_param = cfg.n_param
    d.raw_sigmoid = 1 if cfg.raw_sigmoid else 0
    d.ch_prcp, d.ch_tmean, d.ch_pet = cfg.channels
    d.nearzero = cfg.nearzero
    d.x_t_stride, d.x_b_stride = x.stride(0), x.stride(1)
    if has_mu:  # full [Tx,B,M] contiguous (the module expands broadcasts)
        d.mu_t_stride, d.mu_b_stride = cfg.B * cfg.M, cfg.M
    plan = _DescPlan()
    plan.key, plan.sta, plan.dyn, plan.drop = key, [], [], []
    for i, ps in enumerate(cfg.params):
        s = d.p[ps.slot]
        base = _OFF_P + ps.slot * _PS_SIZE
        s.sta_b_stride = ps.sta_bs
        plan.sta.append((base + _OFF_STA, ps.tensor_idx, 4 * ps.sta_off))
        if ps.dyn_off >= 0:
            s.dyn_t_stride, s.dyn_b_stride = ps.dyn_ts, ps.dyn_bs
            plan.dyn.append((base + _OFF_DYN, ps.dyn_tensor_idx, 4 * ps.dyn_off))
            plan.drop.append((base + _OFF_DROP, i))
        s.lo, s.hi = ps.lo, ps.hi
    d.adj_gtol, d.adj_max_iter, d.adj_stop = cfg.adj_gtol, cfg.adj_max_iter, cfg.adj_stop
    plan.template = bytes(d)
    cfg._plan = plan
    return plan


def _fill_desc(cfg: StepConfig, x, state_in, muwts, ac, elev, ptensors) -> _abi.Desc:
    plan = _desc_plan(cfg, x, muwts is not None)
    buf = bytearray(plan.template)
    pk = _PTR.pack_into
    pk(buf, _D.x.offset, x.data_ptr() + 4 * cfg.t0 * x.stride(0))
    if ac is not None:
        pk(buf, _D.ac.offset, ac.data_ptr())
    if elev is not None:
        pk(buf, _D.elev.offset, elev.data_ptr())
    if muwts is not None:
        pk(buf, _D.muwts.offset, muwts.data_ptr() + 4 * _mu_t0(cfg) * cfg.B * cfg.M)
    if state_in is not None:
        pk(buf, _D.state_in.offset, state_in.data_ptr())
    bases = [t.data_ptr() for t in ptensors]
    for off, ti, boff in plan.sta:
        pk(buf, off, bases[ti] + boff)
    for off, ti, boff in plan.dyn:
        pk(buf, off, bases[ti] + boff)
    params = cfg.params
    for off, i in plan.drop:
        m = params[i].drop
        if m is not None:
            pk(buf, off, m.data_ptr())
    return _abi.Desc.from_buffer(buf)      # the struct keeps `buf` alive


# The library's layout / workspace answers also depend on its (test and tool) environment knobs; their values are
# part of the memo key, so a knob changed between two calls on the same module is seen (a stale TRAJ_PACKED under
# HBVX_STREAM=0 made hbvx_forward refuse the call; a stale workspace size silently demoted the adjoint).
_MEMO_ENV = ("HBVX_KERNEL", "HBVX_FWD", "HBVX_BWD", "HBVX_STREAM", "HBVX_STREAM_MIN", "HBVX_STREAM_MW_MIN", "HBVX_CHUNK",
             "HBVX_CKPT_BLOCK", "HBVX_CKPT_SCRATCH_MB", "HBVX_CKPT_BLOCKWISE", "HBVX_CKPT_ONCHIP")
# os.environ.get() encodes the key and decodes the value on every call (~1 us; ten knobs x several memo look-ups per
# step is a tenth of the delta-MG minibatch's enqueue time): read the mapping underneath it, plain bytes -> bytes
_ENV_DATA = getattr(os.environ, "_data", None)
_MEMO_ENV_B = tuple(os.fsencode(k) for k in _MEMO_ENV)


def _env_key() -> tuple:
    if isinstance(_ENV_DATA, dict):
        return tuple(map(_ENV_DATA.get, _MEMO_ENV_B))
    return tuple(os.environ.get(k) for k in _MEMO_ENV)


def _cached(cfg: StepConfig, key, fn):
    """Per-config memo for the library's size / layout queries (functions of the shape and the knobs above, not of
    the pointers)."""
    memo = cfg.__dict__.setdefault("_memo", {})
    key = (key, _env_key())
    v = memo.get(key)
    if v is None:
        v = memo[key] = fn()
    return v


def _route_desc(cfg: StepConfig, ptensors, S: int = 4) -> _abi.RouteDesc:
    r = _abi.RouteDesc()
    rs = cfg.route
    r.abi_version = _abi.ABI_VERSION
    r.T, r.B, r.S = cfg.T, cfg.B, S
    r.L = min(cfg.T, _abi.UH_MAXLEN)
    r.raw_sigmoid = 1 if cfg.raw_sigmoid else 0
    r.ra = _ptr(ptensors[rs.tensor_idx], rs.a_off)
    r.rb = _ptr(ptensors[rs.tensor_idx], rs.b_off)
    r.r_stride = rs.stride
    r.a_lo, r.a_hi = rs.a_bounds
    r.b_lo, r.b_hi = rs.b_bounds
    return r


class _NoCtx:
    """Stands in for the autograd context when a call needs no gradient (warm-up under no_grad, inference):
    the forward then runs as a plain function, without the ~20 us of autograd.Function.apply."""
    needs_input_grad = ()

    def save_for_backward(self, *a):
        pass

    def mark_non_differentiable(self, *a):
        pass

    def set_materialize_grads(self, v):
        pass


def _hbv_forward(ctx, cfg: StepConfig, x, state_in, muwts, ac, elev, ptensors):
    lib = get_library()
    _check_tensor(lib, x, "x_phy")
    for i, p in enumerate(ptensors):
        _check_tensor(lib, p, f"parameters[{i}]")
        if not p.is_contiguous():
            raise ValueError("parameter tensors must be contiguous")
    if x.stride(2) != 1:
        raise ValueError("x_phy must have unit stride on the forcing axis")
    if muwts is not None:
        _check_tensor(lib, muwts, "muwts")
        if not muwts.is_contiguous() or tuple(muwts.shape[1:]) != (cfg.B, cfg.M):
            raise ValueError("muwts must be a contiguous [T,B,nmul] tensor")
    dev = x.device
    T, B, M = cfg.T, cfg.B, cfg.M
    needs_grad = any(ctx.needs_input_grad)
    keep = needs_grad or cfg.want_traj
    stream = _stream_of(lib, x)

    desc = _fill_desc(cfg, x, state_in, muwts, ac, elev, ptensors)
    out = _abi.FwdOut()
    flux = _out((cfg.n_flux, T, B), dev) \
        if cfg.want_flux else None
    state_out = _out((5, B, M), dev)
    # same two buffers whatever the layout the library asks for (include/hbvx.h, hbvx_traj_layout)
    ckpt = cfg.ckpt_days if (needs_grad and cfg.want_flux and not cfg.want_traj) else 0
    if ckpt:
        # K-day checkpoints [ceil(T/K), 5, N] instead of the trajectory; the adjoint recomputes
        traj = _out(((T + ckpt - 1) // ckpt * 5, B * M), dev)
        aux = None
        layout = _abi.TRAJ_CKPT | (ckpt << 8)
    else:
        traj = _out((5, T + 1, B * M), dev) if keep else None
        # the two saved powers per lane-day: only for A/B builds of the library with -DHBVX_SAVE_POW=1
        # (csrc/hbv_step.h); the default build recomputes them in the adjoint and never touches `aux`
        aux = _out((2, T, B * M), dev) if (needs_grad and lib.saves_pow) else None
        layout = (_cached(cfg, ("layout", x.stride(0), x.stride(1)), lambda: lib.preferred_traj_layout(desc))
                  if (keep and cfg.want_flux) else _abi.TRAJ_ROWS)
    out.flux, out.state_out = _ptr(flux), _ptr(state_out)
    out.traj, out.aux = _ptr(traj), _ptr(aux)
    out.n_flux, out.traj_layout = cfg.n_flux, layout
    ctx.early_gp = None
    if needs_grad and cfg.want_flux and any(ctx.needs_input_grad[6:]):
        ctx.early_gp = _early_zero_request(lib, cfg, ptensors, ctx.needs_input_grad[6:], out)
    _call(lib, 'hbvx_forward', lib.forward, desc, out, stream)
    _early_zero_result(lib, cfg, ctx.early_gp)

    routed = uh = None
    if cfg.route is not None and cfg.want_flux:
        r = _route_desc(cfg, ptensors)
        routed = _out((4, T, B), dev)
        uh = _out((B, r.L), dev)
        _call(lib, 'hbvx_route_forward', lib.route_forward, r, _ptr(flux), _ptr(uh), _ptr(routed), stream)

    # BFI (hbv.py:562-567) of the routed (or, without routing, the raw) streamflow and groundwater series: one
    # more output of this node instead of a second autograd.Function on the caller's side
    bfi = None
    if cfg.want_bfi and cfg.want_flux:
        src = routed if routed is not None else flux
        k2 = 3 if routed is not None else _abi.F_Q2
        bfi = _out((B,), dev)
        _call(lib, 'hbvx_bfi', lib.bfi, T, B, _ptr(src), _ptr(src, k2 * T * B), float(cfg.nearzero), _ptr(bfi), stream)

    ctx.cfg = cfg
    ctx.traj_layout = layout
    ctx.desc = desc if needs_grad else None        # its pointers stay valid: every tensor it names is saved below
    ctx.has_bfi = bfi is not None
    ctx.set_materialize_grads(False)
    if needs_grad:
        ctx.save_for_backward(x, state_in, muwts, ac, elev, traj, aux, flux, uh, routed, *ptensors)
    nondiff = [state_out]
    if traj is not None:
        nondiff.append(traj)
    ctx.mark_non_differentiable(*nondiff)
    # every series leaves as its own [T,B,1] view (the shape the flux dictionary holds): one autograd
    # output each, no select / unsqueeze nodes on the caller's side (16 of them cost the host 0.1 ms
    # per call, a fifth of a deltaMG-sized step)
    rrows = tuple(routed.unsqueeze(-1).unbind(0)) if routed is not None else ()
    rows = tuple(flux.unsqueeze(-1).unbind(0)) if flux is not None else ()
    ctx.n_routed = len(rrows)
    return (state_out, traj, layout, bfi) + rrows + rows


class HbvPath(torch.autograd.Function):
    """state_out, traj, *routed_rows, *flux_rows = HbvPath.apply(cfg, x, state_in, muwts, ac, elev, *ptensors)
    (use `hbv_path`, which regroups the outputs).

    flux_rows  n_flux tensors [T, B, 1]: the ensemble-mean series (enum hbvx_flux order), views of one
               [n_flux, T, B] buffer.  Separate autograd outputs, so that the adjoint learns WHICH
               series carry gradient (a loss on streamflow touches 1-4 of 12) instead of receiving a
               dense, mostly zero [n_flux, T, B] gradient.
    routed_rows  four [T, B, 1] views: UH-routed Qsim, Q0, Q1, Q2 (none when cfg.route is None)
    state_out  [5, B, M]
    traj       saved trajectory or None; layout `cfg.traj_layout` (see `state_series`).
    """

    @staticmethod
    @_device_guard
    def forward(ctx, cfg: StepConfig, x, state_in, muwts, ac, elev, *ptensors):
        return _hbv_forward(ctx, cfg, x, state_in, muwts, ac, elev, ptensors)

    @staticmethod
    @_device_guard
    def backward(ctx, _g_state, _g_traj, _g_layout, g_bfi, *g_all):
        g_rr, g_rows = list(g_all[:ctx.n_routed]), list(g_all[ctx.n_routed:])
        if g_bfi is not None:
            # the gradient a streamflow loss never asks for: two broadcasts in torch, added to the series' own
            saved_ = ctx.saved_tensors
            flux_, routed_ = saved_[7], saved_[9]
            src = routed_ if routed_ is not None else flux_
            qs, q2 = src[0], src[3 if routed_ is not None else _abi.F_Q2]
            den = qs.sum(0) + ctx.cfg.nearzero
            g_q2 = (100.0 * g_bfi / den).unsqueeze(0).expand_as(q2).unsqueeze(-1)
            g_qs = (-100.0 * g_bfi * q2.sum(0) / (den * den)).unsqueeze(0).expand_as(qs).unsqueeze(-1)
            tgt = g_rr if routed_ is not None else g_rows
            for k, g in ((0, g_qs), (3 if routed_ is not None else _abi.F_Q2, g_q2)):
                tgt[k] = g if tgt[k] is None else tgt[k] + g
        # routed series with gradient: the routing adjoint runs over the leading n_live of the four (a
        # streamflow loss: one), the rows behind them stay zero
        g_routed, n_live = None, 0
        live = [k for k, g in enumerate(g_rr) if g is not None]
        if live:
            n_live = max(live) + 1
            first = g_rr[live[0]]
            if n_live == 1:
                g_routed = first[..., 0].unsqueeze(0)
            else:
                g_routed = torch.zeros((n_live,) + tuple(first.shape[:2]), dtype=torch.float32, device=first.device)
                for k in live:
                    g_routed[k] = g_rr[k][..., 0]
        g_rows = tuple(None if g is None else g[..., 0] for g in g_rows)
        lib = get_library()
        cfg: StepConfig = ctx.cfg
        saved = ctx.saved_tensors
        x, state_in, muwts, ac, elev, traj, aux, flux, uh = saved[:9]
        ptensors = saved[10:]
        dev = x.device
        T, B, M = cfg.T, cfg.B, cfg.M
        stream = _stream_of(lib, x)

        early, ctx.early_gp = ctx.early_gp, None
        rows = [None] * len(ptensors)
        bases = [0] * len(ptensors)
        fill_done, start_fill = None, None
        if _FILL_OVERLAP and lib.is_device:
            gp, rows, bases, start_fill = _overlapped_grad_buffers(lib, cfg, ptensors, ctx.needs_input_grad[6:], early)
        else:
            gp = [_grad_like(lib, p, cfg, i) if ctx.needs_input_grad[6 + i] else None
                  for i, p in enumerate(ptensors)]

        def sta_ptr(idx, off):
            """static / routing gradient address: in the separate last row when the tensor has one"""
            return _ptr(rows[idx], off - bases[idx]) if rows[idx] is not None else _ptr(gp[idx], off)

        def join():
            nonlocal fill_done
            if start_fill is not None and fill_done is None:
                fill_done = start_fill()       # (nothing ran in between: the fill simply starts now)
            if fill_done is not None:
                torch.cuda.current_stream(dev).wait_event(fill_done)
            for big, row in zip(gp, rows):      # (also without a fill: the persistent buffers have separate rows too)
                if row is not None:
                    big[-1] += row

        gq = None
        if g_routed is not None and cfg.route is not None:
            r = _route_desc(cfg, ptensors, S=n_live)
            gq = _out((4, T, B), dev)
            if n_live < 4:
                gq[n_live:].zero_()
            rs = cfg.route
            gt = gp[rs.tensor_idx]
            ws_bytes = _cached(cfg, ("route_ws", n_live), lambda: lib.route_workspace_bytes(r))
            ws = torch.empty((max(ws_bytes, 4) + 3) // 4, dtype=torch.float32, device=dev)
            g_routed = g_routed.contiguous()
            _call(lib, 'hbvx_route_backward', lib.route_backward, r, _ptr(flux), _ptr(uh),
                  _ptr(g_routed), _ptr(gq), sta_ptr(rs.tensor_idx, rs.a_off) if gt is not None else None,
                  sta_ptr(rs.tensor_idx, rs.b_off) if gt is not None else None,
                  _ptr(ws), ws_bytes, stream)
        # Which series carry gradient?  Only the four runoff series (the usual losses: streamflow,
        # with or without routing) -> a [4,T,B] buffer and the adjoint's 4-series kernels; anything
        # else -> the dense [n_flux,T,B] form.
        have = [k for k, g in enumerate(g_rows) if g is not None]
        g_flux = None
        if have and max(have) < 4:
            if gq is None:
                gq = torch.zeros((4, T, B), dtype=torch.float32, device=dev)
            for k in have:
                gq[k] += g_rows[k]
        elif have:
            g_flux = torch.zeros((cfg.n_flux, T, B), dtype=torch.float32, device=dev)
            for k in have:
                g_flux[k] = g_rows[k]

        io = _abi.BwdIO()
        io.traj, io.aux = _ptr(traj), _ptr(aux)
        io.grad_flux, io.grad_flux4 = _ptr(g_flux), _ptr(gq)
        io.n_flux, io.traj_layout = cfg.n_flux, ctx.traj_layout
        gx = gmu = None
        if ctx.needs_input_grad[1]:
            gx = _zeros_like(lib, x)
            if gx.stride() != x.stride():
                raise ValueError("x_phy must be dense for a forcing gradient")
            io.grad_x = _ptr(gx, cfg.t0 * x.stride(0))
        if muwts is not None and ctx.needs_input_grad[3]:
            gmu = torch.zeros_like(muwts)
            io.grad_muwts = _ptr(gmu, _mu_t0(cfg) * B * M)
        for ps in cfg.params:
            g = io.g[ps.slot]
            gs = gp[ps.tensor_idx]
            if gs is not None:
                g.sta = sta_ptr(ps.tensor_idx, ps.sta_off)
                g.sta_b_stride = ps.sta_bs
            if ps.dyn_off >= 0 and gp[ps.dyn_tensor_idx] is not None:
                g.dyn = _ptr(gp[ps.dyn_tensor_idx], ps.dyn_off)
                g.dyn_t_stride, g.dyn_b_stride = ps.dyn_ts, ps.dyn_bs
        desc = ctx.desc if ctx.desc is not None else _fill_desc(cfg, x, state_in, muwts, ac, elev, ptensors)
        if g_flux is None and gq is None:
            # nothing flows back through the series (a loss on nothing): gradients are zero
            join()
            for ps in cfg.params:
                if ps.dyn_off >= 0 and gp[ps.dyn_tensor_idx] is not None:
                    gp[ps.dyn_tensor_idx].zero_()
            return (None, gx, None, gmu, None, None, *gp)
        if (ctx.traj_layout & 0xFF) == _abi.TRAJ_CKPT:    # block-wise re-materialisation
            ws_bytes = _cached(cfg, ("ckpt_ws", ctx.traj_layout), lambda: lib.ckpt_workspace_bytes(desc, ctx.traj_layout >> 8))
        elif ctx.traj_layout == _abi.TRAJ_ROWS:
            ws_bytes = _cached(cfg, "bwd_ws", lambda: lib.backward_workspace_bytes(desc))
        else:
            ws_bytes = 0                                     # packed: single pass
        if ws_bytes:
            ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
            io.workspace, io.workspace_bytes = _ptr(ws), ws_bytes
        if start_fill is not None:
            fill_done = start_fill()           # beside the adjoint kernels, behind the routing adjoint
            if fill_done is not None and start_fill.gated:
                io.store_gate = fill_done.cuda_event      # the storing kernel waits for the fill; the others do not
        _call(lib, 'hbvx_backward', lib.backward, desc, io, stream)
        join()

        return (None, gx, None, gmu, None, None, *gp)


class PathOut(NamedTuple):
    """What one call of the path returns.  `flux`: tuple of the n_flux series [T,B,1] (index it with the hbvx_flux
    enum) or None when cfg.want_flux is False; `routed`: tuple of the four UH-routed runoff series [T,B,1] or
    None; `traj` + `traj_layout`: the saved trajectory (see `state_series`); `bfi` [B] when cfg.want_bfi."""
    flux: Optional[tuple]
    routed: Optional[tuple]
    state_out: torch.Tensor
    traj: Optional[torch.Tensor]
    traj_layout: int
    bfi: Optional[torch.Tensor]


def hbv_path(cfg: StepConfig, x, state_in, muwts, ac, elev, *ptensors) -> PathOut:
    """Run the path.  A call that cannot need a gradient (grad mode off, or no input requires one) runs the
    forward as a plain function; otherwise through the autograd node `HbvPath`."""
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in ptensors)
                                    or (muwts is not None and muwts.requires_grad)
                                    or (state_in is not None and state_in.requires_grad)):
        outs = HbvPath.apply(cfg, x, state_in, muwts, ac, elev, *ptensors)
    elif x.is_cuda and torch.cuda.current_device() != x.device.index:
        with torch.cuda.device(x.device):
            outs = _hbv_forward(_NoCtx(), cfg, x, state_in, muwts, ac, elev, ptensors)
    else:
        outs = _hbv_forward(_NoCtx(), cfg, x, state_in, muwts, ac, elev, ptensors)
    state_out, traj, layout, bfi = outs[:4]
    nr = 4 if (cfg.route is not None and cfg.want_flux) else 0
    return PathOut((outs[4 + nr:] or None), (outs[4:4 + nr] or None), state_out, traj, layout, bfi)


def state_series(traj: torch.Tensor, layout: int, T: int, B: int, M: int):
    """The five storages entering day t, t = 0..T (t = T: after the last day), as [T+1, B, M] views of
    the saved trajectory, whatever its layout (include/hbvx.h, hbvx_traj_layout).  Zero-copy."""
    flat = traj.reshape(-1)
    N = B * M
    if layout == _abi.TRAJ_PACKED:
        rec = flat[: 4 * (T + 1) * N].view(T + 1, B, M, 4)
        slz = flat[4 * (T + 1) * N: 5 * (T + 1) * N].view(T + 1, B, M)
        return tuple(rec[..., k] for k in range(4)) + (slz,)
    rows = flat.view(5, T + 1, B, M)
    return tuple(rows[k] for k in range(5))


class HbvAdjPath(torch.autograd.Function):
    """flux, routed, state_out = HbvAdjPath.apply(cfg, x, state_in, *ptensors)

    Implicit (backward-Euler) HBV, reference hbv_adj.py:227-330.  flux [1,T,B] = ensemble-mean
    q0+q1+q2 at the solved states; routed [1,T,B] = its UH routing; state_out [5,B,M] is
    differentiable so that a warm-up call can be chained in front of the main call (the reference
    differentiates through its warm-up, hbv_adj.py:257-274)."""

    @staticmethod
    @_device_guard
    def forward(ctx, cfg: StepConfig, x, state_in, *ptensors):
        lib = get_library()
        _check_tensor(lib, x, "x_phy")
        for i, p in enumerate(ptensors):
            _check_tensor(lib, p, f"parameters[{i}]")
            if not p.is_contiguous():
                raise ValueError("parameter tensors must be contiguous")
        dev = x.device
        T, B, M = cfg.T, cfg.B, cfg.M
        needs_grad = any(ctx.needs_input_grad)
        stream = _stream_of(lib, x)
        out = _abi.FwdOut()
        flux = _out((1, T, B), dev) if cfg.want_flux else None
        state_out = _out((5, B, M), dev)
        traj = _out((5, T + 1, B * M), dev) if needs_grad else None
        out.flux, out.state_out, out.traj, out.n_flux = _ptr(flux), _ptr(state_out), _ptr(traj), 1
        if state_in is not None:
            state_in = state_in.contiguous()
        desc = _fill_desc(cfg, x, state_in, None, None, None, ptensors)
        ctx.early_gp = None
        if needs_grad and any(ctx.needs_input_grad[3:]):
            ctx.early_gp = _early_zero_request(lib, cfg, ptensors, ctx.needs_input_grad[3:], out)
        _call(lib, 'hbvx_adj_forward', lib.adj_forward, desc, out, stream)
        _early_zero_result(lib, cfg, ctx.early_gp)
        routed = uh = None
        if cfg.route is not None and cfg.want_flux:
            r = _route_desc(cfg, ptensors, S=1)
            routed = _out((1, T, B), dev)
            uh = _out((B, r.L), dev)
            _call(lib, 'hbvx_route_forward', lib.route_forward, r, _ptr(flux), _ptr(uh),
                  _ptr(routed), stream)
        ctx.cfg = cfg
        ctx.set_materialize_grads(False)
        if needs_grad:
            ctx.save_for_backward(x, state_in, traj, flux, uh, *ptensors)
        return flux, routed, state_out

    @staticmethod
    @_device_guard
    def backward(ctx, g_flux, g_routed, g_state):
        lib = get_library()
        cfg: StepConfig = ctx.cfg
        saved = ctx.saved_tensors
        x, state_in, traj, flux, uh = saved[:5]
        ptensors = saved[5:]
        dev = x.device
        T, B, M = cfg.T, cfg.B, cfg.M
        stream = _stream_of(lib, x)
        # gradient buffers: the [T,B,ny] fill beside the adjoint kernels on a second stream, static rows apart
        # (as HbvPath.backward; _overlapped_grad_buffers)
        rows = [None] * len(ptensors)
        bases = [0] * len(ptensors)
        start_fill = None
        early, ctx.early_gp = ctx.early_gp, None
        if _FILL_OVERLAP and lib.is_device:
            gp, rows, bases, start_fill = _overlapped_grad_buffers(lib, cfg, ptensors, ctx.needs_input_grad[3:], early)
        else:
            gp = [_grad_like(lib, p, cfg, i) if ctx.needs_input_grad[3 + i] else None
                  for i, p in enumerate(ptensors)]

        def sta_ptr(idx, off):
            return _ptr(rows[idx], off - bases[idx]) if rows[idx] is not None else _ptr(gp[idx], off)

        gq = None
        if g_routed is not None and cfg.route is not None:
            r = _route_desc(cfg, ptensors, S=1)
            gq = _out((1, T, B), dev)
            rs = cfg.route
            gt = gp[rs.tensor_idx]
            ws_bytes = lib.route_workspace_bytes(r)
            ws = torch.empty((max(ws_bytes, 4) + 3) // 4, dtype=torch.float32, device=dev)
            g_routed_c = g_routed.contiguous()     # named: must outlive the launch that reads its pointer
            _call(lib, 'hbvx_route_backward', lib.route_backward, r, _ptr(flux), _ptr(uh),
                  _ptr(g_routed_c), _ptr(gq), sta_ptr(rs.tensor_idx, rs.a_off) if gt is not None else None,
                  sta_ptr(rs.tensor_idx, rs.b_off) if gt is not None else None,
                  _ptr(ws), ws_bytes, stream)
        # contiguous copies of incoming gradients are bound to locals that live until after the launch:
        # a temporary would be freed at once and the caching allocator could hand its block to gs_in / ws
        g_flux_c = g_flux.contiguous() if g_flux is not None else None
        g_state_c = g_state.contiguous() if g_state is not None else None
        io = _abi.BwdIO()
        io.traj = _ptr(traj)
        io.grad_flux = _ptr(g_flux_c)
        io.grad_flux4 = _ptr(gq)
        io.grad_state_out = _ptr(g_state_c)
        io.n_flux = 1
        gs_in = None
        if state_in is not None and ctx.needs_input_grad[2]:
            gs_in = _out((5, B, M), dev)
            io.grad_state_in = _ptr(gs_in)
        for ps in cfg.params:
            g = io.g[ps.slot]
            gs = gp[ps.tensor_idx]
            if gs is not None:
                g.sta = sta_ptr(ps.tensor_idx, ps.sta_off)
                g.sta_b_stride = ps.sta_bs
            if ps.dyn_off >= 0 and gp[ps.dyn_tensor_idx] is not None:
                g.dyn = _ptr(gp[ps.dyn_tensor_idx], ps.dyn_off)
                g.dyn_t_stride, g.dyn_b_stride = ps.dyn_ts, ps.dyn_bs
        desc = _fill_desc(cfg, x, state_in, None, None, None, ptensors)
        ws_bytes = lib.backward_workspace_bytes(desc)
        if ws_bytes:
            ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
            io.workspace, io.workspace_bytes = _ptr(ws), ws_bytes
        fill_done = start_fill() if start_fill is not None else None     # beside the adjoint kernels
        if fill_done is not None and start_fill.gated:
            io.store_gate = fill_done.cuda_event
        _call(lib, 'hbvx_adj_backward', lib.adj_backward, desc, io, stream)
        if fill_done is not None:
            torch.cuda.current_stream(dev).wait_event(fill_done)
        for big, row in zip(gp, rows):
            if row is not None:
                big[-1] += row
        return (None, None, gs_in, *gp)


class Bfi(torch.autograd.Function):
    """BFI = 100 * sum_t q2 / (sum_t qs + nearzero)  (hbv.py:562-567), qs/q2 [T,B] or [T,B,1] views.

    Forward in the library (one deterministic pass); the gradient, which a streamflow loss never
    asks for, is two broadcasts in torch."""

    @staticmethod
    @_device_guard
    def forward(ctx, qs, q2, nearzero: float):
        lib = get_library()
        ctx.cols = qs.dim() == 3
        T, B = qs.shape[:2]
        qs_c, q2_c = qs.contiguous().view(T, B), q2.contiguous().view(T, B)
        out = _out((B,), qs.device)
        _call(lib, 'hbvx_bfi', lib.bfi, T, B, _ptr(qs_c), _ptr(q2_c), float(nearzero), _ptr(out),
              _stream_of(lib, qs_c))
        ctx.save_for_backward(qs_c, q2_c)
        ctx.nearzero = float(nearzero)
        return out

    @staticmethod
    @_device_guard
    def backward(ctx, g):
        qs, q2 = ctx.saved_tensors
        den = qs.sum(0) + ctx.nearzero
        num = q2.sum(0)
        g_q2 = (100.0 * g / den).unsqueeze(0).expand_as(q2)
        g_qs = (-100.0 * g * num / (den * den)).unsqueeze(0).expand_as(qs)
        if ctx.cols:
            g_qs, g_q2 = g_qs.unsqueeze(-1), g_q2.unsqueeze(-1)
        return g_qs, g_q2, None


@dataclass
class GageTopology:
    """Index arrays of the (gage, unit) pairs of `outlet_topo == 1` (hbv_2_hourly.py:813-817), in
    the order `nonzero` yields them (by gage, then unit) plus the unit-side CSR of the backward."""

    T: int = 0
    U: int = 0
    G: int = 0
    n_pair: int = 0
    lag_uh: bool = True
    bounds: tuple = ((0.0, 5.0), (0.0, 12.0), (0.0, 48.0))
    pair_unit: Optional[torch.Tensor] = None
    pair_gage: Optional[torch.Tensor] = None
    gage_ptr: Optional[torch.Tensor] = None
    unit_ptr: Optional[torch.Tensor] = None
    unit_pairs: Optional[torch.Tensor] = None
    areas: Optional[torch.Tensor] = None
    denom: Optional[torch.Tensor] = None

    @staticmethod
    def from_outlet_topo(outlet_topo: torch.Tensor, areas: torch.Tensor, T: int, lag_uh: bool,
                         bounds) -> "GageTopology":
        G, U = int(outlet_topo.shape[0]), int(outlet_topo.shape[1])
        pairs = (outlet_topo == 1).nonzero(as_tuple=False)
        rows, cols = pairs[:, 0], pairs[:, 1]
        dev = outlet_topo.device

        def csr(idx, n):
            ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            ptr[1:] = torch.cumsum(torch.bincount(idx, minlength=n), 0)
            return ptr.to(torch.int32)

        areas32 = areas.to(torch.float32).contiguous()
        denom = (outlet_topo.to(torch.float32) * areas32[None, :]).sum(dim=1).clamp(min=1e-6)
        return GageTopology(
            T=T, U=U, G=G, n_pair=int(rows.numel()), lag_uh=lag_uh, bounds=tuple(map(tuple, bounds)),
            pair_unit=cols.to(torch.int32).contiguous(), pair_gage=rows.to(torch.int32).contiguous(),
            gage_ptr=csr(rows, G), unit_ptr=csr(cols, U),
            unit_pairs=torch.argsort(cols, stable=True).to(torch.int32).contiguous(),
            areas=areas32, denom=denom.contiguous())

    def desc(self, dp: torch.Tensor) -> _abi.GageDesc:
        (alo, ahi), (blo, bhi), (tlo, thi) = self.bounds
        return _abi.GageDesc(
            abi_version=_abi.ABI_VERSION, T=self.T, U=self.U, G=self.G, NPAIR=self.n_pair,
            L=min(self.T, _abi.GAGE_MAXLEN), lag_uh=int(self.lag_uh),
            pair_unit=self.pair_unit.data_ptr(), gage_ptr=self.gage_ptr.data_ptr(),
            pair_gage=self.pair_gage.data_ptr(), unit_ptr=self.unit_ptr.data_ptr(),
            unit_pairs=self.unit_pairs.data_ptr(), areas=self.areas.data_ptr(),
            denom=self.denom.data_ptr(), dp=dp.data_ptr(),
            a_lo=alo, a_hi=ahi, b_lo=blo, b_hi=bhi, tau_lo=tlo, tau_hi=thi)


class GageRoute(torch.autograd.Function):
    """Unit runoff [T,U] -> gage streamflow [T,G] through hbvx_gage_route_forward / _backward
    (hbv_2_hourly.py:800-897)."""

    @staticmethod
    @_device_guard
    def forward(ctx, topo: GageTopology, qs, dp):
        lib = get_library()
        qs_c, dp_c = qs.contiguous(), dp.contiguous()
        _check_tensor(lib, qs_c, 'Qs')
        _check_tensor(lib, dp_c, 'distributed routing parameters')
        if tuple(qs_c.shape) != (topo.T, topo.U):
            raise ValueError(f"Qs has shape {tuple(qs_c.shape)}, topology expects {(topo.T, topo.U)}")
        if tuple(dp_c.shape) != (topo.n_pair, 3):
            raise ValueError(f"distributed routing parameters have shape {tuple(dp_c.shape)}, "
                             f"outlet_topo has {topo.n_pair} (gage, unit) pairs x 3")
        L = min(topo.T, _abi.GAGE_MAXLEN)
        uh = _out((topo.n_pair, L), qs.device)
        out = _out((topo.T, topo.G), qs.device)
        r = topo.desc(dp_c)
        ws_bytes = lib.gage_route_workspace_bytes(r)
        ws = torch.empty((max(ws_bytes, 4) + 3) // 4, dtype=torch.float32, device=qs.device)
        _call(lib, 'hbvx_gage_route_forward', lib.gage_route_forward, r, _ptr(qs_c), _ptr(uh),
              _ptr(out), _ptr(ws), ws_bytes, _stream_of(lib, qs_c))
        ctx.topo = topo
        ctx.save_for_backward(qs_c, dp_c, uh)
        return out

    @staticmethod
    @_device_guard
    def backward(ctx, g):
        lib = get_library()
        qs, dp, uh = ctx.saved_tensors
        topo = ctx.topo
        g = g.contiguous()
        gqs = _out(tuple(qs.shape), qs.device)
        gdp = _out(tuple(dp.shape), dp.device)
        r = topo.desc(dp)
        ws_bytes = lib.gage_route_workspace_bytes(r)
        ws = torch.empty((max(ws_bytes, 4) + 3) // 4, dtype=torch.float32, device=qs.device)
        _call(lib, 'hbvx_gage_route_backward', lib.gage_route_backward, r, _ptr(qs), _ptr(uh),
              _ptr(g), _ptr(gqs), _ptr(gdp), _ptr(ws), ws_bytes, _stream_of(lib, qs))
        return None, gqs, gdp
