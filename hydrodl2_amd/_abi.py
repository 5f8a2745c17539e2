"""ctypes mirror of include/hbvx.h and a thin handle around a loaded library.

This is the binding a maintainer of the reference would add to call the C ABI
from Python.  Struct layouts are verified against the
library's own `hbvx_sizeof()` at load time.
"""
from __future__ import annotations

import ctypes as C
import os

ABI_VERSION = 10
MAX_PARAM = 20
GAGE_MAXLEN = 72
NSTATE = 5
MAX_FLUX = 12
UH_MAXLEN = 15

# enum hbvx_model
MODEL_HBV10, MODEL_HBV11P, MODEL_HBV20, MODEL_HBVADJ, MODEL_HOURLY = 0, 1, 2, 3, 4

# enum hbvx_traj_layout
TRAJ_ROWS, TRAJ_PACKED, TRAJ_CKPT = 0, 1, 2     # traj_layout = kind | (K << 8) for TRAJ_CKPT

# enum hbvx_flux
(F_QSIM, F_Q0, F_Q1, F_Q2, F_AET, F_SWE, F_RECHARGE, F_EXCS, F_EVAPFACTOR, F_TOSOIL, F_PERC,
 F_CAPILLARY) = range(12)

PARAM_SLOTS = ["parBETA", "parFC", "parK0", "parK1", "parK2", "parLP", "parPERC", "parUZL",
               "parTT", "parCFMAX", "parCFR", "parCWH", "parBETAET", "parC", "parRT", "parAC",
               "parF0", "parFMIN", "parALPHA"]

_fp = C.c_void_p  # every float*/uint8_t* travels as an address


class ParamSrc(C.Structure):
    _fields_ = [("dyn", _fp), ("sta", _fp), ("drop", _fp),
                ("dyn_t_stride", C.c_int64), ("dyn_b_stride", C.c_int64),
                ("sta_b_stride", C.c_int64), ("lo", C.c_float), ("hi", C.c_float)]


class ParamGrad(C.Structure):
    _fields_ = [("dyn", _fp), ("sta", _fp),
                ("dyn_t_stride", C.c_int64), ("dyn_b_stride", C.c_int64),
                ("sta_b_stride", C.c_int64)]


class Desc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("model", C.c_int32), ("T", C.c_int32),
                ("B", C.c_int32), ("M", C.c_int32), ("n_param", C.c_int32),
                ("raw_sigmoid", C.c_int32), ("ch_prcp", C.c_int32), ("ch_tmean", C.c_int32),
                ("ch_pet", C.c_int32), ("nearzero", C.c_float), ("adj_stop", C.c_int32),
                ("x", _fp), ("x_t_stride", C.c_int64), ("x_b_stride", C.c_int64),
                ("ac", _fp), ("elev", _fp), ("muwts", _fp),
                ("mu_t_stride", C.c_int64), ("mu_b_stride", C.c_int64),
                ("state_in", _fp), ("p", ParamSrc * MAX_PARAM),
                ("adj_gtol", C.c_float), ("adj_max_iter", C.c_int32)]


class FwdOut(C.Structure):
    _fields_ = [("flux", _fp), ("state_out", _fp), ("traj", _fp), ("aux", _fp),
                ("n_flux", C.c_int32), ("traj_layout", C.c_int32),
                ("zero_ptr", _fp), ("zero_bytes", C.c_uint64), ("zero_state", _fp)]


class BwdIO(C.Structure):
    _fields_ = [("traj", _fp), ("aux", _fp), ("grad_flux", _fp), ("grad_flux4", _fp),
                ("grad_state_out", _fp), ("grad_x", _fp),
                ("grad_muwts", _fp), ("grad_state_in", _fp),
                ("n_flux", C.c_int32), ("traj_layout", C.c_int32),
                ("g", ParamGrad * MAX_PARAM),
                ("workspace", _fp), ("workspace_bytes", C.c_uint64), ("store_gate", _fp)]


class RouteDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("T", C.c_int32), ("B", C.c_int32),
                ("S", C.c_int32), ("L", C.c_int32), ("raw_sigmoid", C.c_int32),
                ("ra", _fp), ("rb", _fp), ("r_stride", C.c_int64),
                ("a_lo", C.c_float), ("a_hi", C.c_float), ("b_lo", C.c_float),
                ("b_hi", C.c_float)]


class GageDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("T", C.c_int32), ("U", C.c_int32), ("G", C.c_int32),
                ("NPAIR", C.c_int32), ("L", C.c_int32), ("lag_uh", C.c_int32),
                ("reserved0", C.c_int32),
                ("pair_unit", _fp), ("gage_ptr", _fp), ("pair_gage", _fp), ("unit_ptr", _fp),
                ("unit_pairs", _fp), ("areas", _fp), ("denom", _fp), ("dp", _fp),
                ("a_lo", C.c_float), ("a_hi", C.c_float), ("b_lo", C.c_float), ("b_hi", C.c_float),
                ("tau_lo", C.c_float), ("tau_hi", C.c_float)]


class LstmDesc(C.Structure):
    """include/hbvx_lstm.h"""
    _fields_ = [("abi_version", C.c_int32), ("T", C.c_int32), ("B", C.c_int32), ("H", C.c_int32)]


LSTM_ABI_VERSION = 1
LSTM_HIDDEN_SIZES = (64, 128, 256)     # what the HIP library instantiates

EXPORTS = ["hbvx_zero", "hbvx_zero_except", "hbvx_preferred_traj_layout", "hbvx_lstm_workspace_bytes", "hbvx_lstm_forward", "hbvx_lstm_backward", "hbvx_lstm_check",
           "hbvx_version", "hbvx_last_error", "hbvx_backend", "hbvx_sizeof", "hbvx_forward",
           "hbvx_backward", "hbvx_backward_workspace_bytes", "hbvx_ckpt_workspace_bytes", "hbvx_route_forward", "hbvx_route_workspace_bytes",
           "hbvx_route_backward", "hbvx_adj_forward", "hbvx_adj_backward", "hbvx_bfi",
           "hbvx_gage_route_forward", "hbvx_gage_route_backward",
           "hbvx_gage_route_workspace_bytes"]


class HbvxError(RuntimeError):
    pass


class Library:
    """A loaded implementation of include/hbvx.h."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.dll = C.CDLL(path)
        d = self.dll
        for name in EXPORTS:
            if not hasattr(d, name):
                raise HbvxError(f"{path}: missing export {name}")
        d.hbvx_version.restype = C.c_int
        d.hbvx_last_error.restype = C.c_char_p
        d.hbvx_backend.restype = C.c_char_p
        d.hbvx_sizeof.restype = C.c_uint64
        d.hbvx_sizeof.argtypes = [C.c_int]
        d.hbvx_forward.restype = C.c_int
        d.hbvx_forward.argtypes = [C.POINTER(Desc), C.POINTER(FwdOut), C.c_void_p]
        d.hbvx_backward.restype = C.c_int
        d.hbvx_backward.argtypes = [C.POINTER(Desc), C.POINTER(BwdIO), C.c_void_p]
        d.hbvx_backward_workspace_bytes.restype = C.c_uint64
        d.hbvx_backward_workspace_bytes.argtypes = [C.POINTER(Desc)]
        d.hbvx_route_forward.restype = C.c_int
        d.hbvx_route_forward.argtypes = [C.POINTER(RouteDesc), _fp, _fp, _fp, C.c_void_p]
        d.hbvx_route_backward.restype = C.c_int
        d.hbvx_route_workspace_bytes.restype = C.c_uint64
        d.hbvx_route_workspace_bytes.argtypes = [C.POINTER(RouteDesc)]
        d.hbvx_route_backward.argtypes = [C.POINTER(RouteDesc), _fp, _fp, _fp, _fp, _fp, _fp,
                                          _fp, C.c_uint64, C.c_void_p]
        for fn in (d.hbvx_adj_forward, d.hbvx_adj_backward):
            fn.restype = C.c_int
        d.hbvx_adj_forward.argtypes = [C.POINTER(Desc), C.POINTER(FwdOut), C.c_void_p]
        d.hbvx_adj_backward.argtypes = [C.POINTER(Desc), C.POINTER(BwdIO), C.c_void_p]
        d.hbvx_gage_route_forward.restype = C.c_int
        d.hbvx_gage_route_workspace_bytes.restype = C.c_uint64
        d.hbvx_gage_route_workspace_bytes.argtypes = [C.POINTER(GageDesc)]
        d.hbvx_gage_route_forward.argtypes = [C.POINTER(GageDesc), _fp, _fp, _fp, C.c_void_p, C.c_uint64, C.c_void_p]
        d.hbvx_gage_route_backward.restype = C.c_int
        d.hbvx_gage_route_backward.argtypes = [C.POINTER(GageDesc), _fp, _fp, _fp, _fp, _fp, C.c_void_p, C.c_uint64,
                                               C.c_void_p]
        d.hbvx_bfi.restype = C.c_int
        d.hbvx_bfi.argtypes = [C.c_int32, C.c_int32, _fp, _fp, C.c_float, _fp, C.c_void_p]
        d.hbvx_ckpt_workspace_bytes.restype = C.c_uint64
        d.hbvx_ckpt_workspace_bytes.argtypes = [C.POINTER(Desc), C.c_int32]
        d.hbvx_preferred_traj_layout.restype = C.c_int
        d.hbvx_preferred_traj_layout.argtypes = [C.POINTER(Desc)]
        d.hbvx_zero_in_launch.restype = C.c_int
        d.hbvx_zero_in_launch.argtypes = []
        d.hbvx_zero_rest.restype = C.c_int
        d.hbvx_zero_rest.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        d.hbvx_last_dispatch.restype = C.c_char_p
        d.hbvx_last_dispatch.argtypes = [C.c_int]
        d.hbvx_zero.restype = C.c_int
        d.hbvx_zero.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        d.hbvx_zero_except.restype = C.c_int
        d.hbvx_zero_except.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
                                       C.c_uint32, C.c_void_p]
        d.hbvx_lstm_workspace_bytes.restype = C.c_uint64
        d.hbvx_lstm_workspace_bytes.argtypes = [C.POINTER(LstmDesc)]
        d.hbvx_lstm_forward.restype = C.c_int
        d.hbvx_lstm_forward.argtypes = [C.POINTER(LstmDesc), _fp, _fp, _fp, _fp, _fp, C.c_void_p, C.c_uint64,
                                        C.c_void_p]
        d.hbvx_lstm_backward.restype = C.c_int
        d.hbvx_lstm_backward.argtypes = [C.POINTER(LstmDesc), _fp, _fp, _fp, _fp, _fp, C.c_void_p, C.c_uint64,
                                         C.c_void_p]
        d.hbvx_lstm_check.restype = C.c_int
        d.hbvx_lstm_check.argtypes = [C.POINTER(LstmDesc), C.c_void_p, C.c_void_p]
        if d.hbvx_version() != ABI_VERSION:
            raise HbvxError(f"{path}: ABI version {d.hbvx_version()} != {ABI_VERSION}")
        for which, st in enumerate([Desc, FwdOut, BwdIO, RouteDesc, ParamSrc, ParamGrad, GageDesc]):
            if d.hbvx_sizeof(which) != C.sizeof(st):
                raise HbvxError(f"{path}: layout mismatch for {st.__name__}: "
                                f"{d.hbvx_sizeof(which)} != {C.sizeof(st)}")
        self.backend = d.hbvx_backend().decode()
        # A/B builds with -DHBVX_SAVE_POW=1 (csrc/hbv_step.h) want the `aux` rows next to the trajectory
        self.saves_pow = "+savepow" in self.backend

    @property
    def is_device(self) -> bool:
        return self.backend.startswith("hip")

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.dll.hbvx_last_error().decode(errors="replace")
            raise HbvxError(f"{what} failed ({rc}): {msg}")

    def forward(self, desc: Desc, out: FwdOut, stream: int):
        self._check(self.dll.hbvx_forward(C.byref(desc), C.byref(out), C.c_void_p(stream)),
                    "hbvx_forward")

    def backward(self, desc: Desc, io: BwdIO, stream: int):
        self._check(self.dll.hbvx_backward(C.byref(desc), C.byref(io), C.c_void_p(stream)),
                    "hbvx_backward")

    def zero_rest(self, ptr: int, nbytes: int, state_ptr: int, stream: int):
        """Zero what the forward's launch left of FwdOut.zero_ptr (include/hbvx.h)."""
        self._check(self.dll.hbvx_zero_rest(C.c_void_p(ptr), nbytes, C.c_void_p(state_ptr), C.c_void_p(stream)), "hbvx_zero_rest")

    def zero_in_launch(self) -> bool:
        """Whether the last forward call of this thread wrote FwdOut.zero_ptr inside its own launch (include/hbvx.h)."""
        return bool(self.dll.hbvx_zero_in_launch())

    def last_dispatch(self, direction: int) -> str:
        """Kernel family of the last forward (0) / adjoint (1) call (diagnostic, include/hbvx.h)."""
        return self.dll.hbvx_last_dispatch(direction).decode()

    def preferred_traj_layout(self, desc: Desc) -> int:
        return int(self.dll.hbvx_preferred_traj_layout(C.byref(desc)))

    def ckpt_workspace_bytes(self, desc: Desc, K: int) -> int:
        return int(self.dll.hbvx_ckpt_workspace_bytes(C.byref(desc), K))

    def backward_workspace_bytes(self, desc: Desc) -> int:
        return int(self.dll.hbvx_backward_workspace_bytes(C.byref(desc)))

    def adj_forward(self, desc: Desc, out: FwdOut, stream: int):
        self._check(self.dll.hbvx_adj_forward(C.byref(desc), C.byref(out), C.c_void_p(stream)),
                    "hbvx_adj_forward")

    def adj_backward(self, desc: Desc, io: BwdIO, stream: int):
        self._check(self.dll.hbvx_adj_backward(C.byref(desc), C.byref(io), C.c_void_p(stream)),
                    "hbvx_adj_backward")

    def gage_route_workspace_bytes(self, r: GageDesc) -> int:
        return int(self.dll.hbvx_gage_route_workspace_bytes(C.byref(r)))

    def gage_route_forward(self, r: GageDesc, qs: int, uh: int, out: int, ws, ws_bytes: int, stream: int):
        self._check(self.dll.hbvx_gage_route_forward(C.byref(r), qs, uh, out, ws, C.c_uint64(ws_bytes),
                                                     C.c_void_p(stream)), "hbvx_gage_route_forward")

    def gage_route_backward(self, r: GageDesc, qs: int, uh: int, go: int, gqs: int, gdp: int, ws,
                            ws_bytes: int, stream: int):
        self._check(self.dll.hbvx_gage_route_backward(C.byref(r), qs, uh, go, gqs, gdp, ws,
                                                      C.c_uint64(ws_bytes), C.c_void_p(stream)),
                    "hbvx_gage_route_backward")

    def zero(self, ptr: int, nbytes: int, stream: int):
        self._check(self.dll.hbvx_zero(ptr, C.c_uint64(nbytes), C.c_void_p(stream)), "hbvx_zero")

    def zero_except(self, ptr: int, rows: int, width: int, r0: int, r1: int, group_w: int, keep: int,
                    stream: int):
        self._check(self.dll.hbvx_zero_except(ptr, rows, width, r0, r1, group_w, C.c_uint32(keep),
                                              C.c_void_p(stream)), "hbvx_zero_except")

    def lstm_workspace_bytes(self, r: LstmDesc) -> int:
        return int(self.dll.hbvx_lstm_workspace_bytes(C.byref(r)))

    def lstm_forward(self, r: LstmDesc, w_hh: int, gx: int, gates: int, c_all: int, h_all: int, ws,
                     ws_bytes: int, stream: int):
        self._check(self.dll.hbvx_lstm_forward(C.byref(r), w_hh, gx, gates, c_all, h_all, ws,
                                               C.c_uint64(ws_bytes), C.c_void_p(stream)), "hbvx_lstm_forward")

    def lstm_backward(self, r: LstmDesc, w_hh: int, gates: int, c_all: int, gh: int, gg: int, ws,
                      ws_bytes: int, stream: int):
        self._check(self.dll.hbvx_lstm_backward(C.byref(r), w_hh, gates, c_all, gh, gg, ws,
                                                C.c_uint64(ws_bytes), C.c_void_p(stream)), "hbvx_lstm_backward")

    def lstm_check(self, r: LstmDesc, ws, stream: int):
        self._check(self.dll.hbvx_lstm_check(C.byref(r), ws, C.c_void_p(stream)), "hbvx_lstm_check")

    def bfi(self, T: int, B: int, qs: int, q2: int, nearzero: float, out: int, stream: int):
        self._check(self.dll.hbvx_bfi(T, B, qs, q2, C.c_float(nearzero), out, C.c_void_p(stream)),
                    "hbvx_bfi")

    def route_forward(self, r: RouteDesc, q: int, uh: int, q_rout: int, stream: int):
        self._check(self.dll.hbvx_route_forward(C.byref(r), q, uh, q_rout, C.c_void_p(stream)),
                    "hbvx_route_forward")

    def route_workspace_bytes(self, r: RouteDesc) -> int:
        return int(self.dll.hbvx_route_workspace_bytes(C.byref(r)))

    def route_backward(self, r: RouteDesc, q: int, uh: int, gqr: int, gq: int, gra, grb,
                       ws, ws_bytes: int, stream: int):
        self._check(self.dll.hbvx_route_backward(C.byref(r), q, uh, gqr, gq, gra, grb, ws,
                                                 C.c_uint64(ws_bytes), C.c_void_p(stream)),
                    "hbvx_route_backward")
