"""Sequence LSTM for the parameter network that feeds the HBV plug-in (the
caller side, delta-MG's CudnnLstmModel -- outside the reference repository).

`SeqLSTM` is a drop-in for `torch.nn.LSTM(input_size, hidden_size, num_layers)` on [T, B, I] input
with zero initial state: same parameter names and shapes (`weight_ih_l0`, `weight_hh_l0`,
`bias_ih_l0`, `bias_hh_l0`), same initialisation, same gate order, so state dicts interchange.

The recurrence runs through include/hbvx_lstm.h (`hbvx_lstm_forward` / `hbvx_lstm_backward`, one
persistent HIP kernel per direction); the time-parallel parts -- input projection, weight and input
gradients -- are plain library GEMMs here.  Like the rest of the package there is no CPU path: on a
host tensor, or without the HIP library, the call raises.
"""
from __future__ import annotations

import math

import torch

from . import _abi
from ._lib import get_library
from . import ops


_PERM = {}


def _gate_perm(H: int, device) -> torch.Tensor:
    """Row index that turns torch's gate-major [4H] (i|f|g|o blocks) into (unit, gate) order (one tensor per hidden
    size and device, built on first use: four small launches less per call)."""
    key = (H, str(device))
    hit = _PERM.get(key)
    if hit is None:
        if torch.cuda.is_available() and torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
            # a tensor created inside a capture belongs to that graph's pool: do not keep it
            return (torch.arange(4, device=device)[None, :] * H + torch.arange(H, device=device)[:, None]).reshape(-1)
        hit = _PERM[key] = (torch.arange(4, device=device)[None, :] * H + torch.arange(H, device=device)[:, None]).reshape(-1)
    return hit


def _wgrad(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a^T b for a [K, G], b [K, I] with K = T*B in the tens of thousands: the weight gradients.  The library's plain
    call covers the [G, I] output with a few dozen tiles and walks all of K in each (55 TFLOP/s at K = 73 000, G = 1024,
    I = 256); cut into S batches along K and summed it runs at 120 (tools/micro/wgrad_gemm.py,
    profiles/r04_lstm_wgrad.txt).  S: a divisor of K near 8, else the plain product."""
    K = a.shape[0]
    if K >= 8192:
        for S in (8, 10, 9, 12, 6, 7, 5, 4, 16, 14, 15, 11, 13):
            if K % S == 0:
                return torch.bmm(a.view(S, K // S, a.shape[1]).transpose(1, 2), b.view(S, K // S, b.shape[1])).sum(0)
    return a.t() @ b


class LstmSeq(torch.autograd.Function):
    """x [T,B,I], W_ih [4H,I], W_hh [4H,H], b_ih, b_hh [4H] -> h [T,B,H], c [T,B,H] (c carries no
    gradient).  `check=True` synchronises and verifies the kernels' hand-off status word."""

    @staticmethod
    @ops._device_guard
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, check: bool = False):
        lib = get_library()
        x_c, w_hh_c = x.contiguous(), w_hh.contiguous()
        for t, name in ((x_c, 'x'), (w_ih, 'weight_ih'), (w_hh_c, 'weight_hh'), (b_ih, 'bias_ih'), (b_hh, 'bias_hh')):
            ops._check_tensor(lib, t, name)
        if x_c.dim() != 3:
            raise ValueError(f"x must be [T, B, input_size], got {tuple(x_c.shape)}")
        T, B, I = x_c.shape
        H = w_hh_c.shape[1]
        if tuple(w_hh_c.shape) != (4 * H, H) or tuple(w_ih.shape) != (4 * H, I) or \
                tuple(b_ih.shape) != (4 * H,) or tuple(b_hh.shape) != (4 * H,):
            raise ValueError("LSTM parameter shapes do not match torch.nn.LSTM(input_size, hidden_size)")
        if lib.is_device and H not in _abi.LSTM_HIDDEN_SIZES:
            raise ValueError(f"hidden_size {H} not built; the HIP library has {_abi.LSTM_HIDDEN_SIZES}")
        perm = _gate_perm(H, x_c.device)
        w_ih_p = w_ih.index_select(0, perm)
        gx = torch.addmm((b_ih + b_hh).index_select(0, perm), x_c.reshape(T * B, I), w_ih_p.t())   # [T*B, 4H]
        r = _abi.LstmDesc(abi_version=_abi.LSTM_ABI_VERSION, T=T, B=B, H=H)
        ws_bytes = lib.lstm_workspace_bytes(r)
        ws = torch.empty((max(ws_bytes, 4) + 3) // 4, dtype=torch.float32, device=x_c.device)
        c_all = ops._out((T, B, H), x_c.device)
        h_all = ops._out((T, B, H), x_c.device)
        st = ops._stream_of(lib, x_c)
        ops._call(lib, 'hbvx_lstm_forward', lib.lstm_forward, r, ops._ptr(w_hh_c), ops._ptr(gx), ops._ptr(gx),
                  ops._ptr(c_all), ops._ptr(h_all), ops._ptr(ws), ws_bytes, st)
        if check:
            lib.lstm_check(r, ops._ptr(ws), st)
        ctx.save_for_backward(x_c, w_ih_p, w_hh_c, gx, c_all, h_all, perm)
        ctx.check = check
        ctx.mark_non_differentiable(c_all)
        return h_all, c_all

    @staticmethod
    @ops._device_guard
    def backward(ctx, gh, _gc):
        lib = get_library()
        x, w_ih_p, w_hh, gates, c_all, h_all, perm = ctx.saved_tensors
        T, B, I = x.shape
        H = w_hh.shape[1]
        gh = gh.contiguous()
        r = _abi.LstmDesc(abi_version=_abi.LSTM_ABI_VERSION, T=T, B=B, H=H)
        ws_bytes = lib.lstm_workspace_bytes(r)
        ws = torch.empty((max(ws_bytes, 4) + 3) // 4, dtype=torch.float32, device=x.device)
        dg = ops._out((T * B, 4 * H), x.device)
        st = ops._stream_of(lib, x)
        ops._call(lib, 'hbvx_lstm_backward', lib.lstm_backward, r, ops._ptr(w_hh), ops._ptr(gates), ops._ptr(c_all),
                  ops._ptr(gh), ops._ptr(dg), ops._ptr(ws), ws_bytes, st)
        if ctx.check:
            lib.lstm_check(r, ops._ptr(ws), st)
        need = ctx.needs_input_grad
        gx = dg @ w_ih_p if need[0] else None                                    # [T*B, I]
        gw_ih = gw_hh = gb = None
        if need[1]:
            gw_ih = torch.empty_like(w_ih_p)
            gw_ih[perm] = _wgrad(dg, x.reshape(T * B, I))
        if need[2]:
            gw_hh = torch.zeros_like(w_hh)
            if T > 1:
                gw_hh[perm] = _wgrad(dg[B:], h_all[:-1].reshape((T - 1) * B, H))
        if need[3] or need[4]:
            gb = torch.empty(4 * H, dtype=dg.dtype, device=dg.device)
            gb[perm] = dg.sum(0)
        return (gx.view(T, B, I) if gx is not None else None), gw_ih, gw_hh, gb, gb, None


def lstm_seq(x, w_ih, w_hh, b_ih, b_hh, check: bool = False):
    """Functional form: returns (h [T,B,H], c [T,B,H]) for torch.nn.LSTM-layout weights."""
    return LstmSeq.apply(x, w_ih, w_hh, b_ih, b_hh, check)


class SeqLSTM(torch.nn.Module):
    """LSTM over [T, B, input_size]; returns (output [T,B,H], (h_n [L,B,H], c_n [L,B,H])) like
    torch.nn.LSTM(input_size, hidden_size, num_layers) called without an initial state.  Layers are
    stacked on the host: layer l's output sequence is layer l+1's input, each layer one pair of
    persistent kernels.  Parameter names follow torch (`weight_ih_l{k}` ...), so state dicts interchange."""

    def __init__(self, input_size: int, hidden_size: int, check: bool = False, dr: float = 0.0,
                 num_layers: int = 1):
        super().__init__()
        if num_layers < 1:
            raise ValueError("SeqLSTM: num_layers must be >= 1")
        self.input_size, self.hidden_size, self.check = input_size, hidden_size, check
        self.num_layers = num_layers
        # dr: weight dropout as in hydroDL / delta-MG's CudnnLstm (one Bernoulli mask on W_ih and one on
        # W_hh per forward call, training mode only); 0 = torch.nn.LSTM behaviour
        self.dr = dr
        k = 1.0 / math.sqrt(hidden_size)
        # same creation order and distribution as torch.nn.LSTM.reset_parameters
        for layer in range(num_layers):
            n_in = input_size if layer == 0 else hidden_size
            self.register_parameter(f"weight_ih_l{layer}", torch.nn.Parameter(torch.empty(4 * hidden_size, n_in).uniform_(-k, k)))
            self.register_parameter(f"weight_hh_l{layer}", torch.nn.Parameter(torch.empty(4 * hidden_size, hidden_size).uniform_(-k, k)))
            self.register_parameter(f"bias_ih_l{layer}", torch.nn.Parameter(torch.empty(4 * hidden_size).uniform_(-k, k)))
            self.register_parameter(f"bias_hh_l{layer}", torch.nn.Parameter(torch.empty(4 * hidden_size).uniform_(-k, k)))

    def forward(self, x):
        if x.is_cuda and self.hidden_size not in _abi.LSTM_HIDDEN_SIZES:
            raise ValueError(f"SeqLSTM: hidden_size must be one of {_abi.LSTM_HIDDEN_SIZES} (the sizes the HIP "
                             f"kernels are instantiated for), got {self.hidden_size}")
        hn, cn = [], []
        for layer in range(self.num_layers):
            w_ih, w_hh = getattr(self, f"weight_ih_l{layer}"), getattr(self, f"weight_hh_l{layer}")
            if self.training and self.dr > 0:
                w_ih = torch.nn.functional.dropout(w_ih, self.dr, training=True)
                w_hh = torch.nn.functional.dropout(w_hh, self.dr, training=True)
            x, c = LstmSeq.apply(x, w_ih, w_hh, getattr(self, f"bias_ih_l{layer}"), getattr(self, f"bias_hh_l{layer}"),
                                 self.check)
            hn.append(x[-1])
            cn.append(c[-1])
        return x, (torch.stack(hn), torch.stack(cn))
