"""HBV 2.0 hourly on the MI355X-native time-stepper.

Counterpart of `hydrodl2.load_model('hbv_2_hourly')`
(src/hydrodl2/models/hbv/hbv_2_hourly.py:8-898): HBV 2.0 in rate form with dt = 1/24 day,
storage guard-rails, Hortonian infiltration excess, 19 physical parameters; `parameters` is the
tuple (dynamic [T,B,n_dy*nmul], static [B,n_st*nmul(+2)], distributed-routing [n_pairs,3]) in
[0,1]; needs `ac_all`, `elev_all`, `outlet_topo` [gages,units] and `areas` [units]; returns
{'Qs': unit runoff [T,B,1], 'streamflow': gage streamflow [T,G,1]}.

Everything numerical runs in the HIP library: the sub-daily recurrence, parameter prep,
ensemble mean and their adjoint (`Step<MODEL_HOURLY>`, csrc/hbv_step_hourly.h), and the 72-tap
lagged-UH gage routing (hbv_2_hourly.py:800-897) through `hbvx_gage_route_*`; the optional
72-tap per-unit routing (:693-700) is the same entry point with the identity topology.
"""
from typing import Any, Optional

import torch

from hydrodl2_amd import _abi
from hydrodl2_amd.core.hbv_module import HbvModule
from hydrodl2_amd.ops import GageRoute, GageTopology, ParamSource, StepConfig, hbv_path, state_series


class Hbv_2_hourly(HbvModule):
    """HBV 2.0 hourly: 19 physical parameters x nmul, 3 distributed-routing parameters per pair."""

    _model_id = _abi.MODEL_HOURLY
    _display_name = 'HBV 2.0 Hourly'
    _has_capillary = True
    _default_routing = False  # hbv_2_hourly.py:44

    def __init__(self, config: Optional[dict[str, Any]] = None,
                 device: Optional[torch.device] = None) -> None:
        self.dt = 1.0 / 24          # hbv_2_hourly.py:58
        self.lenF = 72              # :45
        self.use_distr_routing = True
        self.infiltration = True
        self.lag_uh = True
        self._extra_bounds = {      # :104-114
            'parBETAET': [0.3, 5], 'parC': [0, 1], 'parRT': [0, 20], 'parAC': [0, 2500],
            'parF0': [5.0 / self.dt, 120.0 / self.dt], 'parFMIN': [0.0, 1.0], 'parALPHA': [0.5, 5.0],
        }
        self.distr_parameter_bounds = {'route_a': [0, 5.0], 'route_b': [0, 12.0],
                                       'route_tau': [0, 48.0]}  # :120-124
        self._qs_buffer = []
        self._max_history = 100
        self._topo_cache, self._eye_cache = {}, {}
        super().__init__(config, device)
        self.routing_parameter_bounds = {'route_a': [0, 5.0], 'route_b': [0, 12.0]}  # :116-119
        self._state_cache = None
        self._set_parameters()

    def _read_config(self, config: dict) -> None:
        super()._read_config(config)
        self.cache_states = config.get('cache_states', self.cache_states)

    def _set_parameters(self) -> None:
        """hbv_2_hourly.py:194-211."""
        self.phy_param_names = self.parameter_bounds.keys()
        self.routing_param_names = self.routing_parameter_bounds.keys() if self.routing else []
        self.learnable_param_count1 = len(self.dynamic_params) * self.nmul
        self.learnable_param_count2 = (
            len(self.phy_param_names) - len(self.dynamic_params)
        ) * self.nmul + len(self.routing_param_names)
        self.learnable_param_count3 = len(getattr(self, 'distr_parameter_bounds', {}))
        self.learnable_param_count = (self.learnable_param_count1 + self.learnable_param_count2
                                      + self.learnable_param_count3)

    def get_states(self):
        # The reference returns `_states_cache` although it stores `_state_cache`
        # (hbv_2_hourly.py:170 vs :444); the stored series is what a caller wants.
        return self._state_cache

    def _tuple_param_sources(self, T: int, ngrid: int, wd: int, ws: int, device) -> list:
        """Where each of the 19 physical parameters lives in (dynamic [T,B,wd], static [B,ws])
        (hbv_2_hourly.py:213-322): dynamic in config order, static in bounds order."""
        M = self.nmul
        dy = list(self.dynamic_params)
        if wd != len(dy) * M:
            raise ValueError(f"dynamic parameters have {wd} columns, need {len(dy) * M}")
        srcs = []
        drops = {name: self._draw_drop_mask(ngrid, device) for name in dy}  # :283-288
        stat_list = [name for name in self.parameter_bounds if name not in dy]
        for name in self.parameter_bounds:
            lo, hi = self.parameter_bounds[name]
            slot = _abi.PARAM_SLOTS.index(name)
            if name in dy:
                i = dy.index(name)
                srcs.append(ParamSource(slot=slot, lo=float(lo), hi=float(hi), tensor_idx=0,
                                        sta_off=(T - 1) * ngrid * wd + i * M, sta_bs=wd,
                                        dyn_tensor_idx=0, dyn_off=i * M, dyn_ts=ngrid * wd,
                                        dyn_bs=wd, drop=drops[name]))
            else:
                i = stat_list.index(name)
                srcs.append(ParamSource(slot=slot, lo=float(lo), hi=float(hi), tensor_idx=1,
                                        sta_off=i * M, sta_bs=ws))
        return srcs

    _graph_state_attrs = ('_state_cache',)
    _graph_structural = ('outlet_topo', 'areas')     # their content shapes the launch sequence (the pair lists)

    def forward(self, x_dict: dict[str, torch.Tensor], parameters):
        """Reference: hbv_2_hourly.py:376-449.  `graph=True`: HIP-graph replay (graphed.py)."""
        if self.graph and x_dict['x_phy'].is_cuda:
            from hydrodl2_amd.graphed import graphed_forward
            return graphed_forward(self, x_dict, parameters)
        return self._forward_eager(x_dict, parameters)

    def _topology(self, tag, outlet_topo, areas, T, lag, bounds):
        """GageTopology of (outlet_topo, areas), built once per pair of tensor objects + versions: building it
        costs a `nonzero` (a host synchronisation) and a handful of small kernels per call, and cannot be captured
        into a graph.  Keyed by identity and version counter; weak references guard against a recycled id()."""
        import weakref
        key = (tag, id(outlet_topo), outlet_topo._version, id(areas), areas._version, T, lag)
        hit = self._topo_cache.get(key)
        if hit is not None and hit[0]() is outlet_topo and hit[1]() is areas:
            return hit[2]
        topo = GageTopology.from_outlet_topo(outlet_topo, areas, T, lag, bounds)
        if len(self._topo_cache) > 8:
            self._topo_cache.clear()
        self._topo_cache[key] = (weakref.ref(outlet_topo), weakref.ref(areas), topo)
        return topo

    def _forward_eager(self, x_dict: dict[str, torch.Tensor], parameters):
        """Reference: hbv_2_hourly.py:376-449."""
        x = x_dict['x_phy']
        self.muwts = x_dict.get('muwts', None)
        T, ngrid = x.shape[0], x.shape[1]
        p_dyn, p_sta, p_distr = parameters[0].contiguous(), parameters[1].contiguous(), parameters[2]
        srcs = self._tuple_param_sources(T, ngrid, p_dyn.shape[-1], p_sta.shape[-1], x.device)
        if (not self.states) or (not self.cache_states):
            state_in = None
        else:
            state_in = self._stack_states(self.states, ngrid, x.device)
        n_sta = len(self.parameter_bounds) - len(self.dynamic_params)
        p_route = p_sta[:, n_sta * self.nmul: n_sta * self.nmul + 2] if self.routing else None
        out, series = self._PBM(x, x_dict['ac_all'], x_dict['elev_all'], state_in, srcs, (p_dyn, p_sta),
                                x_dict['outlet_topo'], x_dict['areas'], p_distr, p_route)
        self._state_cache = series                                        # :444
        if self.cache_states:
            self.states = tuple(s[-1].detach() for s in series)
        return out

    def _PBM(self, x, ac, elev, state_in, srcs, ptensors, outlet_topo, areas, p_distr, p_route=None):
        """One pass of the sub-daily recurrence + routing (hbv_2_hourly.py:451-798) with the
        physical parameters described by `srcs` over `ptensors`.  Returns (flux dict, the five
        state series [T,B,M], detached)."""
        T, ngrid, M = x.shape[0], x.shape[1], self.nmul
        ac = ac.to(torch.float32).contiguous()
        elev = elev.to(torch.float32).contiguous()
        cfg = StepConfig(model=self._model_id, n_param=len(self.parameter_bounds), n_flux=12, T=T, t0=0,
                         B=ngrid, M=M, raw_sigmoid=False, channels=self._channels(),
                         nearzero=float(self.nearzero), params=srcs,
                         want_flux=not self.initialize, want_traj=True)
        muwts = self._expand_muwts(self.muwts, T, T, ngrid)
        res = hbv_path(cfg, x, state_in, muwts, ac, elev, *ptensors)
        flux, traj = res.flux, res.traj
        series = tuple(s[1:] for s in state_series(traj.detach(), res.traj_layout, T, ngrid, M))   # :725
        if self.initialize:
            return {}, series

        Qs = flux[_abi.F_QSIM][:, :, 0]                                   # [T,B] rate per day
        if self.routing:                                                  # :684-700
            rb_ = self.routing_parameter_bounds
            eye = self._eye_cache.get((ngrid, str(x.device)))
            if eye is None:
                eye = self._eye_cache[(ngrid, str(x.device))] = (torch.eye(ngrid, device=x.device),
                                                                 torch.ones(ngrid, device=x.device))
            topo = self._topology('self', eye[0], eye[1], T, False, (rb_['route_a'], rb_['route_b'], (0.0, 0.0)))
            dp = torch.cat([p_route, torch.zeros_like(p_route[:, :1])], dim=1)
            Qs = GageRoute.apply(topo, Qs, dp)
        Qs = (Qs * self.dt).unsqueeze(-1)                                 # :741
        out = {'Qs': Qs}
        if not self.warm_up_states:
            out['Qs'] = out['Qs'][self.pred_cutoff:, :, :]
        if self.use_distr_routing:                                        # :766-796
            if self.cache_states:
                self._qs_buffer.append(Qs.detach())
                if len(self._qs_buffer) > self._max_history:
                    self._qs_buffer.pop(0)
                hist = torch.cat(self._qs_buffer, dim=0)
            else:
                hist = Qs
            routed = self.distr_routing(hist, p_distr, outlet_topo, areas)
            out['streamflow'] = routed[-1:] if self.cache_states else routed
        return out, series

    def distr_routing(self, Qs, p_distr, outlet_topo, areas):
        """Gage streamflow from unit runoff: area-weighted, per (gage, unit) pair a gamma unit
        hydrograph of 72 taps shifted by route_tau, summed per gage and normalised by the upstream
        area (hbv_2_hourly.py:800-855).  Qs [T,U,1] -> [T,G,1]."""
        b = self.distr_parameter_bounds
        topo = self._topology('gages', outlet_topo, areas, int(Qs.shape[0]), self.lag_uh,
                              (b['route_a'], b['route_b'], b['route_tau']))
        return GageRoute.apply(topo, Qs[:, :, 0], p_distr).unsqueeze(-1)
