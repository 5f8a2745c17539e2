"""CPU tier, authoring container only: the public attribute surface of every model class, read off the REFERENCE itself
(imported from /root/reference as tests/golden/make_golden.py does) and compared with this package's class built from the
same config.  SURVEY.md §8b lists the attributes callers read (hbv.py:43-105,172-180; hbv_2.py:182-188).  Skipped where the
reference is not present (the GPU box: nothing of the reference travels)."""
import os
import sys
import types

import pytest
import torch

REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")

ATTRS = ["name", "initialize", "warm_up", "pred_cutoff", "warm_up_states", "dynamic_params", "dy_drop", "variables",
         "routing", "comprout", "nearzero", "nmul", "cache_states", "states", "state_names", "flux_names",
         "parameter_bounds", "routing_parameter_bounds", "phy_param_names", "routing_param_names",
         "learnable_param_count", "learnable_param_count1", "learnable_param_count2"]

CONFIGS = [
    ("hbv", "Hbv", None),
    ("hbv", "Hbv", {"nmul": 16, "warm_up": 30, "dy_drop": 0.2, "dynamic_params": {"Hbv": ["parBETA", "parBETAET"]}}),
    ("hbv", "Hbv", {"nmul": 2, "warm_up": 10, "warm_up_states": False, "routing": False, "nearzero": 1e-6,
                    "variables": ["tmean", "pet", "prcp"], "cache_states": True, "dynamic_params": {"Hbv": []}}),
    ("hbv_1_1p", "Hbv_1_1p", None),
    ("hbv_1_1p", "Hbv_1_1p", {"nmul": 4, "dynamic_params": {"Hbv_1_1p": ["parC", "parBETA"]}}),
    ("hbv_2", "Hbv_2", None),
    ("hbv_2", "Hbv_2", {"nmul": 4, "routing": True, "dynamic_params": {"Hbv_2": ["parBETA", "parK0", "parBETAET"]}}),
    ("hbv_2_hourly", "Hbv_2_hourly", None),
    ("hbv_2_hourly", "Hbv_2_hourly", {"nmul": 2, "routing": True, "dynamic_params": {"Hbv_2_hourly": ["parBETA", "parF0"]}}),
]


@pytest.fixture(scope="module")
def reference():
    sys.path.insert(0, REF)
    v = types.ModuleType("hydrodl2._version")
    v.__version__ = "1.2.0"
    sys.modules.setdefault("hydrodl2._version", v)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import hydrodl2
    return hydrodl2


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if torch.is_tensor(v):
        return v.tolist()
    return v


@pytest.mark.parametrize("family,cls,cfg", CONFIGS, ids=[f"{c[1]}-{i}" for i, c in enumerate(CONFIGS)])
def test_public_attributes_equal_the_references(family, cls, cfg, reference):
    import warnings
    import hydrodl2_amd
    dev = torch.device("cpu")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = reference.load_model(family, cls)(None if cfg is None else dict(cfg), dev)
        own = hydrodl2_amd.load_model(family, cls)(None if cfg is None else dict(cfg), dev)
    diffs = []
    for a in ATTRS:
        if not hasattr(ref, a):
            continue
        assert hasattr(own, a), f"{cls}: attribute {a!r} of the reference is missing"
        rv, ov = _plain(getattr(ref, a)), _plain(getattr(own, a))
        if rv != ov:
            diffs.append(f"{a}: reference {rv!r} != {ov!r}")
    assert not diffs, f"{cls} {cfg}: " + "; ".join(diffs)
    # dict ORDER matters to callers that zip names with columns
    assert list(ref.parameter_bounds) == list(own.parameter_bounds)
    assert list(ref.flux_names) == list(own.flux_names) and list(ref.state_names) == list(own.state_names)


def test_model_discovery_equals_the_references(reference):
    """`available_models()` (api/methods.py:78-139): same families and class files, minus what cannot be imported upstream
    (hbv_adj needs an encrypted dependency there; here it loads)."""
    import hydrodl2_amd
    ref, own = reference.available_models(), hydrodl2_amd.available_models()
    assert set(ref) <= set(own), (ref, own)
    for fam, names in ref.items():
        assert set(names) <= set(own[fam]), (fam, names, own[fam])
    # an unknown name: the reference means to raise ImportError("Model '...' not found.") (api/methods.py:116-117) but only
    # catches ImportError around the file load, so the missing FILE surfaces as FileNotFoundError; this package raises the
    # ImportError the reference words.  Pinned here so that the difference is a recorded one
    with pytest.raises(FileNotFoundError):
        reference.load_model("no_such_family")
    with pytest.raises(ImportError, match="not found"):
        hydrodl2_amd.load_model("no_such_family")
    # an unknown class in a known file: both fall back to the first class of the module (api/methods.py:123-137)
    assert reference.load_model("hbv", "Nope").__name__ == hydrodl2_amd.load_model("hbv", "Nope").__name__ == "Hbv"


@pytest.mark.parametrize("bad", [(torch.zeros(2, 1),) * 4, [torch.zeros(2, 1)] * 5, (1, 2, 3, 4, 5)])
def test_load_states_rejects_what_the_reference_rejects(bad, reference):
    import hydrodl2_amd
    dev = torch.device("cpu")
    for mod in (reference, hydrodl2_amd):
        m = mod.load_model("hbv", "Hbv")(None, dev)
        with pytest.raises(ValueError):
            m.load_states(bad)


def test_multi_timescale_wrapper_has_the_references_surface(reference):
    """Hbv_2_mts(low_freq_config, high_freq_config, device) (hbv_2_mts.py:31-75): every public data attribute the
    reference's instance carries exists here with the same value; sub-models are the same classes in the same roles.
    (The reference indexes high_freq_config for its chunk sizes: a config without them is a TypeError there and here.)"""
    import warnings
    import hydrodl2_amd
    from . import golden_mts
    dev = torch.device("cpu")
    low, high = golden_mts.configs("mts_chunked")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = reference.load_model("hbv_2_mts", "Hbv_2_mts")(dict(low), dict(high), dev)
        own = hydrodl2_amd.load_model("hbv_2_mts", "Hbv_2_mts")(dict(low), dict(high), dev)
        for mod in (reference, hydrodl2_amd):
            with pytest.raises(TypeError):
                mod.load_model("hbv_2_mts", "Hbv_2_mts")(None, None, dev)
    diffs = []
    for a, rv in vars(ref).items():
        if a.startswith("_") or a == "training":
            continue
        assert hasattr(own, a), f"Hbv_2_mts: attribute {a!r} of the reference is missing"
        ov = getattr(own, a)
        if isinstance(rv, torch.nn.Module):
            if type(rv).__name__ != type(ov).__name__:
                diffs.append(f"{a}: {type(rv).__name__} != {type(ov).__name__}")
        elif _plain(rv) != _plain(ov):
            diffs.append(f"{a}: reference {_plain(rv)!r} != {_plain(ov)!r}")
    assert not diffs, diffs
    assert ref.low_freq_model.initialize is True and own.low_freq_model.initialize is True
    for sub in ("low_freq_model", "high_freq_model"):
        r, o = getattr(ref, sub), getattr(own, sub)
        for a in ATTRS:
            if hasattr(r, a):
                assert _plain(getattr(r, a)) == _plain(getattr(o, a)), (sub, a)
