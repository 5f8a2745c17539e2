"""Test-side seam: point the package's host logic at another implementation of the C ABI.

The product (`hydrodl2_amd`) loads exactly one library, its in-tree HIP build, and has no switch
for anything else.  The CPU tier still wants to drive the package's *host* logic (parameter
unpacking, warm-up orchestration, autograd plumbing, sharding) without a GPU, against the oracle
(`oracle/liboracle_hbv.so`, same ABI on host memory).  That substitution lives here, in tests/:
it replaces the module-level handle `hydrodl2_amd._lib._lib` and is never imported by the package.
"""
from __future__ import annotations

from typing import Optional


def use_library(path: Optional[str]) -> None:
    """`path` = a shared library exporting include/hbvx.h (the oracle, a host-compiled or probe
    build); None = back to the product's own HIP library (loaded lazily by get_library())."""
    from hydrodl2_amd import _lib
    from hydrodl2_amd._abi import Library
    _lib._lib = Library(path) if path is not None else None
