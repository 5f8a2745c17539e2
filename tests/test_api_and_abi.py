"""CPU tier: the plug-in API mirrors the reference's (tests/test_methods.py there), the C-ABI
library loads and exports every symbol include/hbvx.h declares, and host-side validation."""
import ctypes as C
import os
import re

import pytest
import torch
from torch.nn import Module

import hydrodl2_amd
from hydrodl2_amd import _abi, available_models, load_model
from hydrodl2_amd.api.methods import _list_available_models

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# --- the reference's own API tests, restated for this package (tests/test_methods.py:16-47) ---
def test_available_models():
    models = available_models()
    assert isinstance(models, dict)
    assert all(isinstance(v, list) for v in models.values())
    assert all(isinstance(v, str) for v in models.keys())
    assert len(models) > 0


@pytest.mark.parametrize('model', _list_available_models())
def test_load_model(model):
    loaded_model = load_model(model)
    assert loaded_model is not None, f"Failed to load '{model}'."
    assert isinstance(loaded_model, type), f"Loaded '{model}' is not a class."
    assert issubclass(loaded_model, Module)


@pytest.mark.parametrize("model, ver_name", [("hbv", "Hbv"), ("hbv_1_1p", "Hbv_1_1p"),
                                             ("hbv_2", "Hbv_2"), ("Hbv_2", "Hbv_2")])
def test_load_model_with_version(model, ver_name):
    cls = load_model(model, ver_name=ver_name)
    assert isinstance(cls, type) and cls.__name__ == ver_name


def test_load_model_unknown():
    with pytest.raises(ImportError):
        load_model("does_not_exist")
    with pytest.raises(NotImplementedError):
        hydrodl2_amd.load_module()


# --- constructor / attribute contract (hbv.py:37-126,170-180) ---
def test_defaults_and_param_counts():
    Hbv = load_model("hbv", "Hbv")
    m = Hbv(None, torch.device("cpu"))
    assert (m.nmul, m.warm_up, m.routing, m.nearzero, m.dy_drop) == (1, 0, True, 1e-5, 0.0)
    assert m.learnable_param_count == 14 and m.variables == ['prcp', 'tmean', 'pet']
    assert list(m.parameter_bounds)[:3] == ['parBETA', 'parFC', 'parK0'] and 'BFI' in m.flux_names
    m = Hbv({"nmul": 16, "dynamic_params": {"Hbv": ["parBETA", "parBETAET"]}}, torch.device("cpu"))
    assert m.learnable_param_count == 13 * 16 + 2 and m.parameter_bounds['parBETAET'] == [0.3, 5]
    with pytest.raises(KeyError):  # config without dynamic_params: same failure as the reference
        Hbv({"nmul": 2}, torch.device("cpu"))
    H11 = load_model("hbv_1_1p", "Hbv_1_1p")
    assert H11({"nmul": 16, "dynamic_params": {}}, torch.device("cpu")).learnable_param_count == 226
    H2 = load_model("hbv_2", "Hbv_2")
    m2 = H2({"nmul": 4, "dynamic_params": {"Hbv_2": ["parBETA", "parK0"]}}, torch.device("cpu"))
    assert (m2.learnable_param_count1, m2.learnable_param_count2, m2.routing) == (8, 56, False)


def test_load_states_validation():
    m = load_model("hbv", "Hbv")(None, torch.device("cpu"))
    with pytest.raises(ValueError):
        m.load_states((torch.zeros(2, 1),) * 4)
    with pytest.raises(ValueError):
        m.load_states([torch.zeros(2, 1)] * 5)
    with pytest.raises(ValueError):
        m.load_states((1, 2, 3, 4, 5))
    m.load_states(tuple(torch.zeros(2, 1, dtype=torch.float64) for _ in range(5)))
    assert all(s.dtype == torch.float32 for s in m.states) and m.get_states() is None


# --- the C ABI library itself (no compute: there is no GPU here) ---
def _declared_symbols():
    inc = os.path.join(ROOT, "include")
    text = "\n".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    return sorted(set(re.findall(r"^(?:int|uint64_t|const char \*)\s*(hbvx_[a-z_0-9]+)\s*\(", text,
                                 flags=re.M)))


def test_header_symbols_are_exported_by_the_hip_library():
    import __graft_entry__ as ge
    lib_path = ge.build_hip()
    dll = C.CDLL(lib_path)
    for sym in _declared_symbols():
        assert hasattr(dll, sym), f"{sym} declared in include/*.h but not exported"
    lib = _abi.Library(lib_path)           # version + struct layout checks
    assert lib.backend == "hip:gfx950"


def test_hip_library_rejects_bad_descriptors():
    """Validation happens on the host before any launch, so it is testable without a GPU."""
    import __graft_entry__ as ge
    lib = _abi.Library(ge.build_hip())
    d, o = _abi.Desc(), _abi.FwdOut()
    with pytest.raises(_abi.HbvxError, match="abi_version"):
        lib.forward(d, o, 0)
    d.abi_version, d.model, d.T, d.B, d.M, d.n_param = _abi.ABI_VERSION, 0, 4, 2, 65, 12
    with pytest.raises(_abi.HbvxError, match="T/B/M"):
        lib.forward(d, o, 0)
    d.M, d.n_param = 4, 14
    with pytest.raises(_abi.HbvxError, match="n_param"):
        lib.forward(d, o, 0)
    d.n_param = 12
    with pytest.raises(_abi.HbvxError, match="forcing"):
        lib.forward(d, o, 0)


def test_hip_library_refuses_bounds_outside_the_short_forms():
    """The forward's evaporation factor (from the storage before the excess leaves it) and its powers (no lower clamp
    of the base) equal the reference only for LP <= 1 and positive lower bounds of BETA / BETAET / ALPHA
    (csrc/hbv_step.h::fwd_soil, pow_unit_; hbv.py:462-477): a caller-edited parameter_bounds table outside that is
    refused by check_desc with a message, before any launch."""
    import numpy as np
    import __graft_entry__ as ge
    from tests.abi_util import BOUNDS
    lib = _abi.Library(ge.build_hip())
    buf = np.zeros(64, np.float32)

    def desc(model, names, **edit):
        d = _abi.Desc()
        d.abi_version, d.model, d.T, d.B, d.M, d.n_param = _abi.ABI_VERSION, model, 4, 2, 4, len(names)
        d.x = buf.ctypes.data
        d.ch_prcp, d.ch_tmean, d.ch_pet = 0, 1, 2
        d.nearzero = 1e-5
        d.ac = d.elev = buf.ctypes.data
        for i, n in enumerate(names):
            d.p[i].sta = buf.ctypes.data
            d.p[i].lo, d.p[i].hi = edit.get(n, BOUNDS[n])
        return d
    from tests.golden_cases import PHY_NAMES
    hbv = PHY_NAMES["Hbv"] + ["parBETAET"]
    o = _abi.FwdOut()
    with pytest.raises(_abi.HbvxError, match="state_out"):          # the reference's own table passes the bounds check
        lib.forward(desc(_abi.MODEL_HBV10, hbv), o, 0)
    for names, model, edit, msg in [
            (hbv, _abi.MODEL_HBV10, {"parLP": (0.2, 1.5)}, "parLP"),
            (hbv, _abi.MODEL_HBV10, {"parBETA": (0.0, 6.0)}, "parBETA "),
            (hbv, _abi.MODEL_HBV10, {"parBETAET": (0.0, 5.0)}, "parBETAET"),
            (PHY_NAMES["Hbv_2_hourly"], _abi.MODEL_HOURLY, {"parALPHA": (-0.5, 5.0)}, "parALPHA")]:
        with pytest.raises(_abi.HbvxError, match=msg):
            lib.forward(desc(model, names, **edit), o, 0)


def test_product_fails_loudly_without_gpu_tensors():
    from hydrodl2_amd import _lib
    from tests import seam
    seam.use_library(None)
    m = load_model("hbv", "Hbv")(None, torch.device("cpu"))
    with pytest.raises(RuntimeError, match="GPU|HIP"):
        m({"x_phy": torch.rand(6, 2, 3)}, torch.randn(6, 2, 14))


def _nan_forcing_case(device):
    import torch
    import hydrodl2_amd
    T, B, M = 40, 3, 4
    g = torch.Generator().manual_seed(3)
    x = torch.stack([torch.rand((T, B), generator=g) * 10, torch.randn((T, B), generator=g) * 5 + 3,
                     torch.rand((T, B), generator=g) * 4], -1)
    x[7, 1, 1] = float("nan")            # one missing temperature
    model = hydrodl2_amd.load_model("hbv", "Hbv")({"nmul": M, "dynamic_params": {"Hbv": []}}, device)
    p = torch.randn((T, B, model.learnable_param_count), generator=g)
    return model, x.to(device), p.to(device)


def _check_nan_policy(model, x, p):
    """Pins the documented divergence: the step's clamps are fmaxf / fminf, which drop a NaN
    operand, so a missing forcing value is absorbed (finite output; the reference turns that basin NaN
    from that day on) -- unless `check_finite` is set, which raises instead."""
    import torch
    out = model({"x_phy": x}, p)
    assert torch.isfinite(out["streamflow"]).all()
    model.check_finite = True
    with pytest.raises(ValueError, match="non-finite"):
        model({"x_phy": x}, p)
    x2 = torch.nan_to_num(x, nan=1.0)
    assert torch.isfinite(model({"x_phy": x2}, p)["streamflow"]).all()


def test_nan_forcing_policy_on_oracle(oracle_backend):
    _check_nan_policy(*_nan_forcing_case("cpu"))


@pytest.mark.gpu
def test_nan_forcing_policy_on_gpu(hip_backend):
    _check_nan_policy(*_nan_forcing_case("cuda:0"))


def _zero_length_record(device):
    """T = 0: nothing to step; the final storages are the initial ones and nothing else is touched."""
    import torch
    from hydrodl2_amd import _abi
    from hydrodl2_amd.ops import ParamSource, StepConfig, hbv_path
    dev = torch.device(device)
    B, M, n = 3, 4, 12
    ny = n * M + 2
    p = torch.randn((1, B, ny), device=dev)
    x = torch.zeros((0, B, 3), device=dev)
    srcs = [ParamSource(slot=i, lo=0.5, hi=1.0, tensor_idx=0, sta_off=i * M, sta_bs=ny) for i in range(n)]   # (positive bounds: check_desc)
    cfg = StepConfig(model=_abi.MODEL_HBV10, n_param=n, n_flux=11, T=0, t0=0, B=B, M=M, raw_sigmoid=True,
                     channels=(0, 1, 2), nearzero=1e-5, params=srcs)
    state_in = torch.rand((5, B, M), device=dev)
    flux, routed, state_out, traj = hbv_path(cfg, x, state_in, None, None, None, p)[:4]
    assert torch.equal(state_out, state_in)
    assert all(f.shape[0] == 0 for f in flux)


def test_zero_length_record_on_oracle(oracle_backend):
    _zero_length_record("cpu")


@pytest.mark.gpu
def test_zero_length_record_on_gpu(hip_backend):
    _zero_length_record("cuda:0")


def _memo_ids_over_three_dy_drop_calls(family, cls, device):
    """A call with dy_drop masks runs on a per-call step configuration (the masks belong to that call and its backward);
    what is memoised on a configuration -- the library's size / layout answers, the descriptor plan, the persistent
    gradient buffer -- is a function of the shapes and must survive the call.  (Until round 5 a module whose FIRST call
    carried masks left the memo on the copy: every later call rebuilt it, and grad_buffer='persistent' allocated and
    zero-filled a new [T,B,ny] buffer per step.)"""
    import torch
    import hydrodl2_amd
    from hydrodl2_amd import ops
    T, B, M = 40, 3, 4
    dyn = ["parBETA", "parBETAET"] if cls == "Hbv" else ["parBETAET"]
    model = hydrodl2_amd.load_model(family, cls)({"nmul": M, "dy_drop": 0.5, "dynamic_params": {cls: dyn}}, device)
    seen = []
    real = ops._cached

    def spy(cfg, key, fn):
        seen.append(id(cfg.__dict__.setdefault("_memo", {})))
        return real(cfg, key, fn)
    ops._cached = spy
    try:
        for rnd in range(3):
            g = torch.Generator().manual_seed(rnd)
            x = torch.stack([torch.rand((T, B), generator=g) * 10, torch.randn((T, B), generator=g) * 5 + 3,
                             torch.rand((T, B), generator=g) * 4], -1)
            p = torch.randn((T, B, model.learnable_param_count), generator=g).to(device).requires_grad_(True)
            torch.manual_seed(rnd)
            out = model({"x_phy": x.to(device)}, p)
            next(iter(out.values())).sum().backward()
    finally:
        ops._cached = real
    return seen


def test_dy_drop_calls_share_the_memo_of_their_configuration(oracle_backend):
    seen = _memo_ids_over_three_dy_drop_calls("hbv", "Hbv", "cpu")
    assert seen and len(set(seen)) == 1, seen


@pytest.mark.gpu
def test_dy_drop_calls_share_the_memo_on_gpu(hip_backend):
    seen = _memo_ids_over_three_dy_drop_calls("hbv", "Hbv", "cuda:0")
    assert seen and len(set(seen)) == 1, seen


@pytest.mark.gpu
def test_implicit_scheme_keeps_one_memo_per_shape_under_dy_drop(hip_backend):
    """HbvAdj builds its masked step configurations per call; their memo (the persistent gradient buffer) is kept per
    shape on the module and handed to every one of them."""
    import torch
    import hydrodl2_amd
    T, B, M = 40, 3, 4
    model = hydrodl2_amd.load_model("hbv_adj", "HbvAdj")({"nmul": M, "dy_drop": 0.5, "dynamic_params": {"HbvAdj": ["parBETAET"]}}, "cuda:0")
    memos = []
    for rnd in range(3):
        g = torch.Generator().manual_seed(rnd)
        x = torch.stack([torch.rand((T, B), generator=g) * 10, torch.randn((T, B), generator=g) * 5 + 3,
                         torch.rand((T, B), generator=g) * 4], -1).to("cuda:0")
        p = torch.randn((T, B, model.learnable_param_count), generator=g).to("cuda:0").requires_grad_(True)
        out = model({"x_phy": x}, p)
        next(iter(out.values())).sum().backward()
        assert len(model._memo_cache) == 1
        memos.append(id(next(iter(model._memo_cache.values()))))
    assert len(model._memo_cache) == 1 and len(set(memos)) == 1, (memos, len(model._memo_cache))


def test_comprout_with_an_ensemble_is_refused_like_the_reference(oracle_backend):
    """comprout=True reshapes [T, B, nmul] to B * nmul routed channels while the unit hydrograph has B groups: the
    reference's grouped convolution raises a RuntimeError for nmul > 1 (hbv.py:516-530; checked against it: "shape
    '[12, 1, 15]' is invalid for input of size 45").  Same exception type here; nmul == 1 runs (golden case
    hbv_comprout_m1)."""
    import torch
    import hydrodl2_amd
    model = hydrodl2_amd.load_model("hbv", "Hbv")({"nmul": 4, "comprout": True, "dynamic_params": {"Hbv": []}}, "cpu")
    x = torch.rand(30, 3, 3)
    p = torch.randn(30, 3, model.learnable_param_count)
    with pytest.raises(RuntimeError, match="comprout"):
        model({"x_phy": x}, p)
