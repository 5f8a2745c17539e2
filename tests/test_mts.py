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
