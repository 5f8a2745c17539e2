"""Locate and load libhbvx.so (the HIP implementation of include/hbvx.h).

The product has exactly one compute path: the HIP library built in-tree by
`__graft_entry__.build()` (hipcc --offload-arch=gfx950).  There is no CPU
fallback: if the library is missing, or a tensor is not on the GPU, the call
fails loudly.

`_lib` is the one module-level handle; nothing in this package assigns it except
`get_library()` (the CPU tier's substitution of a host implementation of the ABI
lives in tests/seam.py, not here).
"""
from __future__ import annotations

import os
from typing import Optional

from ._abi import Library

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libhbvx.so")

_lib: Optional[Library] = None


def get_library() -> Library:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"hydrodl2_amd: HIP library not built ({LIB_PATH} missing). "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950) first. There is no CPU fallback."
            )
        lib = Library(LIB_PATH)
        if not lib.is_device:
            raise RuntimeError(f"{LIB_PATH}: backend {lib.backend!r} is not a HIP build")
        _lib = lib
    return _lib
