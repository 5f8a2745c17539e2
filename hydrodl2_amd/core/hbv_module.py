"""Host-side mirror of the reference's HBV plug-in interface.

The classes built on `HbvModule` keep the reference's constructor, attribute,
`forward(x_dict, parameters)`, `get_states()` / `load_states()` contract
(reference: src/hydrodl2/models/hbv/hbv.py:37-361) so that they are drop-ins for
`hydrodl2.load_model('hbv')` in a dMG-style caller.  What differs is where the
work happens: nothing is unpacked, repeated or looped over in Python -- the raw
NN output, the forcings and the states go straight to the HIP library
(`hydrodl2_amd.ops.HbvPath`) as pointers + strides.
"""
from __future__ import annotations

from typing import Any, Optional, Union

import copy
import os

import torch

from .. import _abi
from ..ops import Bfi, ParamSource, RouteSource, StepConfig, hbv_path


class HbvModule(torch.nn.Module):
    """Shared implementation of HBV 1.0 / 1.1p (single raw `parameters` tensor)."""

    # subclasses set these
    _model_id = _abi.MODEL_HBV10
    _display_name = "HBV 1.0"
    _extra_bounds: dict = {}
    _has_capillary = False
    _default_routing = True

    def __init__(self, config: Optional[dict[str, Any]] = None,
                 device: Optional[torch.device] = None) -> None:
        super().__init__()
        # reference: hbv.py:43-59
        self.name = self._display_name
        self.config = config
        self.initialize = False
        self.warm_up = 0
        self.pred_cutoff = 0
        self.warm_up_states = True
        self.dynamic_params = []
        self.dy_drop = 0.0
        self.variables = ['prcp', 'tmean', 'pet']
        self.routing = self._default_routing
        self.comprout = False
        self.nearzero = 1e-5
        self.nmul = 1
        self.cache_states = False
        self.device = device
        self.muwts = None
        # not in the reference: 0 (save the state trajectory for the adjoint, fastest) or 4 / 8 / 16
        # (keep K-day checkpoints and recompute: 20/K instead of 28 bytes per lane-day); config key
        # 'adjoint_checkpoint' or, when unset, the environment variable HBVX_CKPT_DAYS
        self.adjoint_checkpoint = int(os.environ.get('HBVX_CKPT_DAYS', '0') or 0)
        # not in the reference: the step's clamps (fmaxf / fminf) drop a NaN operand where torch.clamp /
        # torch.minimum propagate it, so a missing forcing value gives finite output here and a NaN basin
        # upstream.  With this key (or HBVX_CHECK_FINITE=1) non-finite forcings or
        # parameters raise instead of being absorbed; off by default (it costs two reductions per call).
        self.check_finite = os.environ.get('HBVX_CHECK_FINITE', '0') not in ('', '0')
        # not in the reference: replay the call's launch sequence (forward and backward) as HIP graphs captured per
        # input shape -- for steps whose kernels are shorter than their enqueue time (hydrodl2_amd/graphed.py)
        self.graph = False
        # 'fresh' (default): every backward returns a newly allocated, zero-filled gradient for `parameters`, as
        # autograd does.  'persistent': the dense [T,B,ny] gradient lives in a buffer the module keeps per input
        # shape and only the parts that change are rewritten (ops._overlapped_grad_buffers) -- valid until the next
        # backward with that shape; not for gradient accumulation across calls on a leaf tensor.
        self.grad_buffer = 'fresh'

        self.states, self._states_cache = None, None
        self._cfg_cache, self._pmat_cache = {}, {}     # per input shape: step configs / Bernoulli probabilities

        self.state_names = ['SNOWPACK', 'MELTWATER', 'SM', 'SUZ', 'SLZ']  # hbv.py:61-67
        self.flux_names = [  # hbv.py:68-86
            'streamflow', 'srflow', 'ssflow', 'gwflow', 'AET_hydro', 'PET_hydro', 'SWE',
            'streamflow_no_rout', 'srflow_no_rout', 'ssflow_no_rout', 'gwflow_no_rout',
            'recharge', 'excs', 'evapfactor', 'tosoil', 'percolation',
        ]
        if self._has_capillary:
            self.flux_names.append('capillary')  # hbv_1_1p.py:83
        self.flux_names.append('BFI')

        self.parameter_bounds = {  # hbv.py:88-101
            'parBETA': [1.0, 6.0], 'parFC': [50, 1000], 'parK0': [0.05, 0.9],
            'parK1': [0.01, 0.5], 'parK2': [0.001, 0.2], 'parLP': [0.2, 1],
            'parPERC': [0, 10], 'parUZL': [0, 100], 'parTT': [-2.5, 2.5],
            'parCFMAX': [0.5, 10], 'parCFR': [0, 0.1], 'parCWH': [0, 0.2],
        }
        self.parameter_bounds.update(self._extra_bounds)
        self.routing_parameter_bounds = {'route_a': [0, 2.9], 'route_b': [0, 6.5]}  # :102-105

        if not device:
            self.device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')

        if config is not None:
            self._read_config(config)
        self._set_parameters()

    # -- configuration --------------------------------------------------
    # config key -> attribute of the same name; absent keys keep the constructor default
    _CONFIG_KEYS = ('warm_up', 'warm_up_states', 'dy_drop', 'variables', 'routing', 'comprout',
                    'nearzero', 'nmul', 'cache_states', 'adjoint_checkpoint', 'check_finite', 'graph', 'grad_buffer')

    def _read_config(self, config: dict) -> None:
        """Same keys and defaults as hbv.py:110-125; `dynamic_params` is REQUIRED once a config is
        given and is keyed by the class name (hbv.py:115-117)."""
        for key in self._CONFIG_KEYS:
            if key in config:
                setattr(self, key, config[key])
        per_class = config['dynamic_params']
        self.dynamic_params = per_class.get(type(self).__name__, self.dynamic_params)
        if self._model_id == _abi.MODEL_HBV10 and 'parBETAET' in self.dynamic_params:
            self.parameter_bounds['parBETAET'] = [0.3, 5]  # hbv.py:124-125

    def _set_parameters(self) -> None:
        """Name views and the width of the raw NN output: n_phys * nmul (+ 2 routing columns),
        hbv.py:170-180."""
        self.phy_param_names = self.parameter_bounds.keys()
        self.routing_param_names = self.routing_parameter_bounds.keys() if self.routing else []
        n_phys, n_route = len(self.phy_param_names), len(self.routing_param_names)
        self.learnable_param_count = n_phys * self.nmul + n_route

    # -- state API (hbv.py:128-168) --------------------------------------
    def _init_states(self, ngrid: int):
        """None = let the kernel start every storage at 0.001 (hbv.py:128-136)."""
        return None

    def get_states(self) -> Optional[tuple[torch.Tensor, ...]]:
        return self._states_cache

    def load_states(self, states: tuple[torch.Tensor, ...]) -> None:
        """Same checks and messages as hbv.py:150-168; stored detached, fp32, on the model device."""
        want = len(self.state_names)
        if any(not isinstance(s, torch.Tensor) for s in states):
            raise ValueError("Each element in `states` must be a tensor.")
        if not isinstance(states, tuple) or len(states) != want:
            raise ValueError(f"`states` must be a tuple of {want} tensors.")
        self.states = tuple(s.detach().to(self.device, dtype=torch.float32) for s in states)

    # -- helpers ---------------------------------------------------------
    def _channels(self):
        return (self.variables.index('prcp'), self.variables.index('tmean'),
                self.variables.index('pet'))

    def _n_flux(self) -> int:
        return 12 if self._has_capillary else 11

    def _draw_drop_mask(self, ngrid: int, device) -> Optional[torch.Tensor]:
        """One Bernoulli(dy_drop) draw per basin from the CPU global RNG, exactly as the
        reference consumes it (hbv.py:240,245) -- also when dy_drop == 0."""
        pmat = torch.ones([1, ngrid, 1]) * self.dy_drop
        drmask = torch.bernoulli(pmat)      # drawn even for dy_drop == 0: same RNG consumption
        if self.dy_drop <= 0:
            return None                     # nothing dropped: the kernels take their mask-free paths
        return drmask.reshape(ngrid).to(torch.uint8).to(device)

    def _draw_drop_masks(self, n_dyn: int, ngrid: int, device):
        """The n_dyn draws of `_draw_drop_mask`, in the reference's order (one per dynamic parameter, hbv.py:236-
        246), as ONE call: the CPU generator hands out its numbers element by element, so an [n_dyn, B] draw
        consumes the stream exactly like n_dyn successive [1, B, 1] draws (tests: golden case hbv_dyn2_drop).
        Returns None per parameter when nothing can be dropped."""
        if n_dyn == 0:
            return []
        key = (n_dyn, ngrid, float(self.dy_drop))
        pm = self._pmat_cache.get(key)
        if pm is None:
            pm = self._pmat_cache[key] = torch.full((n_dyn, ngrid), float(self.dy_drop))
        drmask = torch.bernoulli(pm)
        if self.dy_drop <= 0:
            return [None] * n_dyn
        dm = drmask.to(torch.uint8).to(device)
        return [dm[k] for k in range(n_dyn)]

    def _stack_states(self, states, ngrid: int, device):
        if states is None:
            return None
        st = torch.stack([s.to(device=device, dtype=torch.float32).reshape(ngrid, self.nmul)
                          for s in states])
        return st.contiguous()

    def _expand_muwts(self, muwts, T: int, T_total: int, B: int):
        """Ensemble weights for the T simulated (post warm-up) days, dense [T,B,nmul].
        The reference multiplies `muwts` against Qsimmu of length T (hbv.py:497-511): broadcast
        shapes ([B,nmul], [1,B,nmul]) and a time-resolved [T,B,nmul] are what it accepts; a
        [T_total,B,nmul] tensor (rows aligned with x_phy) is accepted here too and cut at warm_up."""
        if muwts is None:
            return None
        mu = muwts.to(torch.float32)
        if mu.dim() == 3 and mu.shape[0] == T_total and T_total != T:
            mu = mu[T_total - T:]
        return mu.expand(T, B, self.nmul).contiguous()

    def _param_sources(self, T_total: int, B: int, ny: int, t_first: int, sta_row: int,
                       dy_list, device, draw: bool = True):
        """Addressing of every physical parameter inside raw `parameters[T,B,ny]`:
        column i*nmul + j (hbv.py:201-208); static value = row `sta_row` (hbv.py:242)."""
        M = self.nmul
        srcs = []
        for i, name in enumerate(self.parameter_bounds.keys()):
            lo, hi = self.parameter_bounds[name]
            slot = _abi.PARAM_SLOTS.index(name)
            ps = ParamSource(slot=slot, lo=float(lo), hi=float(hi), tensor_idx=0,
                             sta_off=sta_row * B * ny + i * M, sta_bs=ny)
            if name in dy_list:
                ps.dyn_tensor_idx = 0
                ps.dyn_off = t_first * B * ny + i * M
                ps.dyn_ts, ps.dyn_bs = B * ny, ny
                if draw:
                    ps.drop = self._draw_drop_mask(B, device)
            srcs.append(ps)
        return srcs

    def _step_configs(self, T_total: int, ngrid: int, ny: int, warm_up: int, device):
        """(warm-up config or None, main config) for this input shape.  Everything in them -- shapes, bounds,
        element offsets of every parameter inside `parameters[T,B,ny]`, the routing columns -- is a function of
        the shape and of the module's settings, so they are built once and reused (their descriptor plans with
        them, ops._desc_plan): a deltaMG minibatch step spent more host time rebuilding them than the GPU spent on
        the kernels.  dy_drop masks are per call; a config that carries them is a copy."""
        M, n = self.nmul, len(self.parameter_bounds)
        key = (T_total, ngrid, ny, warm_up, M, n, tuple(self.dynamic_params), bool(self.routing), self._model_id,
               tuple(self.variables), float(self.nearzero), int(self.adjoint_checkpoint),
               tuple(map(tuple, self.parameter_bounds.values())),
               tuple(map(tuple, self.routing_parameter_bounds.values())), str(device), str(self.grad_buffer))
        hit = self._cfg_cache.get(key)
        if hit is not None:
            return hit
        base = dict(model=self._model_id, n_param=n, n_flux=self._n_flux(), B=ngrid, M=M,
                    raw_sigmoid=True, channels=self._channels(), nearzero=float(self.nearzero),
                    ckpt_days=int(self.adjoint_checkpoint), persistent_grad=self._persistent_grad())
        cfg_w = None
        if warm_up > 0:     # hbv.py:327-346: all parameters static from row warm_up-1, states only
            cfg_w = StepConfig(T=warm_up, t0=0, want_flux=False, **base)
            cfg_w.params = self._param_sources(T_total, ngrid, ny, 0, warm_up - 1, [], device, draw=False)
        T = T_total - warm_up
        cfg = StepConfig(T=T, t0=warm_up, want_bfi=True, **base)
        cfg.params = self._param_sources(T_total, ngrid, ny, warm_up, T_total - 1, self.dynamic_params, device,
                                         draw=False)
        if self.routing:
            off = (T_total - 1) * ngrid * ny + n * M  # hbv.py:212-214: last row only
            cfg.route = RouteSource(0, off, off + 1, ny,
                                    self.routing_parameter_bounds['route_a'],
                                    self.routing_parameter_bounds['route_b'])
        cfg.mu_t0 = 0                      # muwts rows are post-warm-up days already
        if len(self._cfg_cache) > 16:
            self._cfg_cache.clear()
        self._cfg_cache[key] = (cfg_w, cfg)
        return cfg_w, cfg

    def _persistent_grad(self) -> bool:
        if self.grad_buffer not in ('fresh', 'persistent'):
            raise ValueError("grad_buffer must be 'fresh' or 'persistent'")
        return self.grad_buffer == 'persistent'

    def _settings_key(self):
        """Everything besides the input shapes that decides what a call launches (key of the step-config and graph
        caches)."""
        return (self.nmul, len(self.parameter_bounds), tuple(self.dynamic_params), bool(self.routing), self._model_id,
                tuple(self.variables), float(self.nearzero), int(self.adjoint_checkpoint), int(self.warm_up),
                bool(self.warm_up_states), float(self.dy_drop), bool(self.comprout), str(self.grad_buffer),
                tuple(map(tuple, self.parameter_bounds.values())),
                tuple(map(tuple, self.routing_parameter_bounds.values())))

    def _advance_rng(self, ngrid: int) -> None:
        """Consume the CPU generator as one call of forward() does (the dy_drop draws, hbv.py:240-246)."""
        self._draw_drop_masks(len(self.dynamic_params), ngrid, torch.device('cpu'))

    # -- forward ---------------------------------------------------------
    def forward(self, x_dict: dict[str, torch.Tensor], parameters: torch.Tensor
                ) -> Union[tuple, dict[str, torch.Tensor]]:
        """Reference: hbv.py:284-361 (orchestration) + :363-596 (`_PBM`)."""
        if self.graph and x_dict['x_phy'].is_cuda:
            from hydrodl2_amd.graphed import graphed_forward
            return graphed_forward(self, x_dict, parameters)
        return self._forward_eager(x_dict, parameters)

    def _forward_eager(self, x_dict: dict[str, torch.Tensor], parameters: torch.Tensor
                       ) -> Union[tuple, dict[str, torch.Tensor]]:
        x = x_dict['x_phy']
        self.__dict__['muwts'] = x_dict.get('muwts', None)     # (plain attribute: nn.Module.__setattr__ costs 5 us)
        T_total, ngrid = x.shape[0], x.shape[1]
        M = self.nmul
        n = len(self.parameter_bounds)
        if not parameters.is_contiguous():
            parameters = parameters.contiguous()
        ny = parameters.shape[2]
        if ny < n * M + (2 if self.routing else 0):
            raise ValueError(f"parameters has {ny} columns, need {n * M + 2}")
        if self.comprout and M != 1:
            # reference: uh_conv's grouped conv has `ngrid` groups but ngrid*nmul channels
            # (hbv.py:516-530) -> RuntimeError there as well.
            raise RuntimeError("comprout=True is only consistent for nmul == 1")

        # hbv.py:314-319
        if self.warm_up_states:
            warm_up = self.warm_up
        else:
            self.pred_cutoff = self.warm_up
            warm_up = 0

        # hbv.py:321-324
        if (not self.states) or (not self.cache_states):
            state_in = self._init_states(ngrid)
        else:
            state_in = self._stack_states(self.states, ngrid, x.device)

        if self.adjoint_checkpoint not in (0, 4, 8, 16):
            raise ValueError("adjoint_checkpoint must be 0, 4, 8 or 16 days")
        if self.check_finite:
            for name, t in (('x_phy', x), ('parameters', parameters)):
                if not bool(torch.isfinite(t).all()):
                    raise ValueError(f"{name} holds non-finite values (check_finite is set)")
        cfg_w, cfg = self._step_configs(T_total, ngrid, ny, warm_up, x.device)

        # hbv.py:327-346: state warm-up, all parameters static from row warm_up-1, no grad
        if cfg_w is not None:
            with torch.no_grad():
                state_in = hbv_path(cfg_w, x, state_in, None, None, None, parameters.detach()).state_out

        # hbv.py:349-353.  The dy_drop masks are drawn per call (hbv.py:240-246), also when nothing can drop
        T = T_total - warm_up
        dyn_idx = [i for i, ps in enumerate(cfg.params) if ps.dyn_off >= 0]
        masks = self._draw_drop_masks(len(dyn_idx), ngrid, x.device)
        if any(m is not None for m in masks):
            # a config with masks belongs to this call (and its backward) alone -- but not what is memoised on it (the
            # library's size / layout answers, the descriptor plan, the persistent gradient buffer): functions of the
            # shapes, shared with the cached original.  (Until round 5 a first call WITH masks left its memo on the copy:
            # every later call rebuilt it, and grad_buffer='persistent' allocated and filled a new buffer per step.)
            shared = cfg
            shared.__dict__.setdefault("_memo", {})
            cfg = copy.copy(cfg)
            cfg.params = [copy.copy(ps) for ps in cfg.params]
            for i, m in zip(dyn_idx, masks):
                cfg.params[i].drop = m
        muwts = self._expand_muwts(self.muwts, T, T_total, ngrid)
        res = hbv_path(cfg, x, state_in, muwts, None, None, parameters)
        flux, routed, state_out = res.flux, res.routed, res.state_out
        if any(m is not None for m in masks) and "_plan" in cfg.__dict__:
            shared.__dict__.setdefault("_plan", cfg.__dict__["_plan"])      # the plan holds offsets, no pointers

        # hbv.py:356-359
        self.__dict__['_states_cache'] = list(state_out.detach().unbind(0))
        if self.cache_states:
            self.states = self._states_cache

        if self.initialize:
            return tuple(self._states_cache)
        return self._assemble(flux, routed, x, warm_up, res.bfi)

    def _assemble(self, flux, routed, x, t0, bfi=None) -> dict[str, torch.Tensor]:
        """hbv.py:555-596: flux dictionary, BFI and the optional prediction cut-off."""
        F = _abi
        # the kernels' series arrive as [T,B,1] views already (ops.HbvPath)
        if routed is not None:
            Qs, Q0r, Q1r, Q2r = routed
        else:
            # The reference's Hbv crashes here (hbv.py:550-567); follow Hbv_2's
            # handling of routing=False instead (hbv_2.py:620-626).
            Qs, Q0r, Q1r, Q2r = (flux[k] for k in (F.F_QSIM, F.F_Q0, F.F_Q1, F.F_Q2))
        BFI = bfi if bfi is not None else Bfi.apply(Qs, Q2r, float(self.nearzero))
        pet = x[t0:, :, self.variables.index('pet')]
        out = {
            'streamflow': Qs, 'srflow': Q0r, 'ssflow': Q1r, 'gwflow': Q2r,
            'AET_hydro': flux[F.F_AET],
            'PET_hydro': pet.unsqueeze(-1),
            'SWE': flux[F.F_SWE],
            'streamflow_no_rout': flux[F.F_QSIM],
            'srflow_no_rout': flux[F.F_Q0],
            'ssflow_no_rout': flux[F.F_Q1],
            'gwflow_no_rout': flux[F.F_Q2],
            'recharge': flux[F.F_RECHARGE],
            'excs': flux[F.F_EXCS],
            'evapfactor': flux[F.F_EVAPFACTOR],
            'tosoil': flux[F.F_TOSOIL],
            'percolation': flux[F.F_PERC],
        }
        if self._has_capillary:
            out['capillary'] = flux[F.F_CAPILLARY]
        out['BFI'] = BFI
        if not self.warm_up_states:
            for key in out.keys():
                if key != 'BFI':
                    out[key] = out[key][self.pred_cutoff:, :, :]
        return out
