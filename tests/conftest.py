import os

# every output buffer of the library starts as NaN in the tests: an element a kernel skips cannot
# pass on stale memory from the caching allocator (hydrodl2_amd/ops.py::_out)
os.environ.setdefault("HBVX_DEBUG_POISON", "1")
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle_hbv.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def oracle_path():
    """Build (if needed) and return the CPU oracle shared library."""
    src = os.path.join(ORACLE_DIR, "hbv_oracle.c")
    if (not os.path.exists(ORACLE_LIB)
            or os.path.getmtime(ORACLE_LIB) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return ORACLE_LIB


@pytest.fixture()
def oracle_backend(oracle_path):
    """Route the package's host logic to the CPU oracle (tests only)."""
    from tests import seam
    seam.use_library(oracle_path)
    yield
    seam.use_library(None)


@pytest.fixture(scope="session")
def hip_backend():
    """The product path: the HIP library on cuda:0.  Fails loudly if absent."""
    import torch
    from hydrodl2_amd import _lib
    from tests import seam
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    seam.use_library(None)
    lib = _lib.get_library()
    assert lib.is_device
    return lib
