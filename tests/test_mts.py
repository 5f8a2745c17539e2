"""Hbv_2_mts against reference-generated fixtures
(tests/golden/make_golden_mts.py) -- CPU tier through the oracle, GPU tier through the HIP library --
plus properties of the chunked driver.  Tolerances as in tests/helpers.py."""
import numpy as np
import pytest
import torch

import hydrodl2_amd

from . import golden_mts as gm
from .helpers import compare, load_golden


def _model(name, dev):
    low, high = gm.configs(name)
    cls = hydrodl2_amd.load_model("hbv_2_mts", "Hbv_2_mts")
    return cls(low, high, torch.device(dev))


def _check(name, dev):
    ref = load_golden(name)
    res = gm.run(_model(name, dev), name, torch.device(dev))
    res = {k: v for k, v in res.items()}
    want = {k: ref[k] for k in ref.files if k.startswith(("out/", "grad/"))}

    class R:  # the subset of the npz interface compare() uses
        files = list(want)

        def __getitem__(self, k):
            return want[k]
    compare(name, res, R())


@pytest.mark.parametrize("name", list(gm.CASES))
def test_mts_matches_reference_oracle(name, oracle_backend):
    _check(name, "cpu")


def test_reference_chunked_path_needed_the_alias():
    ref = load_golden("mts_chunked")
    assert not bool(ref["upstream_runs"])
    assert "unpack_parameters" in str(ref["upstream_error"])


def test_spatial_chunking_does_not_change_runoff(oracle_backend):
    """Units are independent: the blocks' runoff equals the one-block runoff bit for bit."""
    name = "mts_chunked"
    m = _model(name, "cpu")
    a = gm.run(m, name, torch.device("cpu"))
    m2 = _model(name, "cpu")
    m2.simulate_spatial_chunk_size = 100
    b = gm.run(m2, name, torch.device("cpu"))
    assert np.array_equal(a["out/Qs"], b["out/Qs"])
    assert np.array_equal(a["out/streamflow"], b["out/streamflow"])


def test_states_and_mode(oracle_backend):
    name = "mts_train"
    m = _model(name, "cpu")
    gm.run(m, name, torch.device("cpu"))
    lo, hi = m.get_states()
    assert len(lo) == 5 and len(hi) == 5
    assert hi[0].shape == (gm.CASES[name]["T_high"], gm.CASES[name]["B"], gm.M)
    with pytest.raises(ValueError):
        m.load_states((lo,))
    m.set_mode(True)
    assert m.simulate_mode and m.spatial_chunk_size == m.simulate_spatial_chunk_size
    bad_low, high = gm.configs(name)
    bad_low["cache_states"] = False
    m3 = hydrodl2_amd.load_model("hbv_2_mts", "Hbv_2_mts")(bad_low, high, torch.device("cpu"))
    with pytest.raises(ValueError, match="cache_states"):
        gm.run(m3, name, torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(gm.CASES))
def test_mts_matches_reference_gpu(name, hip_backend):
    _check(name, "cuda")


def _hourly_inputs(T, B, G, seed, dev):
    import numpy as np
    from . import synth
    x = synth.forcing(T, B, seed) * np.array([1.0 / 8.0, 1.0, 1.0 / 24.0], np.float32)
    topo = (synth.uniform((G, B), seed, 13) < np.float32(0.5)).astype(np.float32)
    topo[np.arange(B) % G, np.arange(B)] = 1.0
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xd = {"x_phy": t(x), "ac_all": t((synth.uniform((B,), seed, 7) * np.float32(5000)).astype(np.float32)),
          "elev_all": t((synth.uniform((B,), seed, 8) * np.float32(3000)).astype(np.float32)),
          "outlet_topo": t(topo), "areas": t((synth.uniform((B,), seed, 14) * np.float32(90) + 5).astype(np.float32))}
    pd = t(np.zeros((T, B, 0), np.float32))
    ps = t(synth.unit_parameters((B, 19 * gm.M), seed, 6))
    pr = t(synth.unit_parameters((int(topo.sum()), 3), seed, 15))
    return xd, (pd, ps, pr)


def _hourly_streaming(dev):
    """cache_states (hbv_2_hourly.py:424-447,766-796): calls on consecutive windows carry the storages
    and the runoff history.  With static parameters the unit runoff equals the one-call run bit for
    bit, and the gage value returned for the last hour equals the one-call value (the history
    buffer holds the whole record here, so the routing sees the same series)."""
    T, B, G = 30, 5, 2
    cls = hydrodl2_amd.load_model("hbv_2_hourly", "Hbv_2_hourly")
    cfg = {"nmul": gm.M, "dynamic_params": {"Hbv_2_hourly": []}}
    xd, (pd, ps, pr) = _hourly_inputs(T, B, G, 77, torch.device(dev))
    with torch.no_grad():
        one = cls(dict(cfg), torch.device(dev))(xd, (pd, ps, pr))
        m = cls(dict(cfg, cache_states=True), torch.device(dev))
        qs, last = [], None
        for t0 in range(0, T, 10):
            sl = slice(t0, t0 + 10)
            out = m(dict(xd, x_phy=xd["x_phy"][sl]), (pd[sl].contiguous(), ps, pr))
            qs.append(out["Qs"])
            last = out["streamflow"]
    assert torch.equal(torch.cat(qs, 0), one["Qs"])
    assert last.shape == (1, G, 1)
    assert torch.equal(last[0], one["streamflow"][-1])


def test_hourly_cache_states_streaming_oracle(oracle_backend):
    _hourly_streaming("cpu")


@pytest.mark.gpu
def test_hourly_cache_states_streaming_gpu(hip_backend):
    _hourly_streaming("cuda")
