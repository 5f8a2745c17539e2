"""Test harness only.  Python imports `sitecustomize` at start-up from the first sys.path entry that
has one; tests/test_bench_launch.py puts this directory on PYTHONPATH so that EVERY process of a
`python bench.py --gpus N --device cpu` run -- the launcher parent and the ranks it starts -- gets the
host implementation of the C ABI named in HBVX_TEST_ABI_LIBRARY installed through tests/seam.py.
Nothing in the product or in bench.py knows about this file."""
import os
import sys

_lib = os.environ.get("HBVX_TEST_ABI_LIBRARY")
_root = os.environ.get("HBVX_TEST_ROOT")
if _lib and _root and os.environ.get("WORLD_SIZE"):   # ranks only: the launcher parent computes nothing
    sys.path.insert(0, _root)
    from tests import seam
    seam.use_library(_lib)
