"""Bit-reproducible synthetic inputs for the HBV parity tests.

Every array is derived from integer hashing (splitmix64) and exactly
representable float arithmetic only, so the golden-fixture generator
(tests/golden/make_golden.py, run once in the authoring container beside the
reference) and the tests (run anywhere, including the GPU box where the
reference does not exist) see the very same bits without the inputs having to
be stored in the fixtures.

Shapes follow the reference's conventions:
  x_phy       [T, B, 3]   forcings ordered (prcp, tmean, pet)
  parameters  [T, B, ny]  raw NN output, column = i_param * nmul + j_member
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(shape, seed: int, stream: int = 0) -> np.ndarray:
    """U[0,1) with 24 random bits per element (exact in float32)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        key = np.uint64((seed * 0x100000001B3 + stream * 0x9E3779B1 + 0x1234567) & 0xFFFFFFFFFFFFFFFF)
        bits = _splitmix64(_splitmix64(idx ^ key) + key)
    u = (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return u.reshape(shape).astype(np.float32)


def normalish(shape, seed: int, stream: int = 0) -> np.ndarray:
    """Approximately N(0,1): Irwin-Hall sum of 4 uniforms, exact arithmetic."""
    acc = np.zeros(shape, dtype=np.float64)
    for k in range(4):
        acc += uniform(shape, seed, stream * 8 + k + 101).astype(np.float64)
    # var of one U is 1/12 -> sum of 4 has var 1/3; scale by sqrt(3) rounded to
    # a short binary fraction so the product stays exact in float64.
    return ((acc - 2.0) * 1.734375).astype(np.float32)


def forcing(T: int, B: int, seed: int, cold: bool = False) -> np.ndarray:
    """CAMELS-shaped synthetic forcing [T,B,3] (prcp mm/d, tmean degC, pet mm/d).

    ~30 % wet days, temperature with a triangular seasonal cycle that crosses
    the parTT range [-2.5, 2.5] so both rain and snow branches are exercised.
    cold=True forces a cold, dry start (ties: melt = SNOWPACK = 0).
    """
    u = uniform((T, B, 3), seed, 1).astype(np.float64)
    day = np.arange(T, dtype=np.float64)[:, None]
    phase = (day % 365.0) / 365.0
    tri = np.where(phase < 0.5, 4.0 * phase - 1.0, 3.0 - 4.0 * phase)  # [-1,1]
    boff = uniform((1, B), seed, 2).astype(np.float64) * 20.0 - 8.0
    P = np.maximum(0.0, (u[:, :, 0] - 0.7) * 60.0)
    Tm = 12.0 * tri + (u[:, :, 1] * 14.0 - 7.0) + boff
    PET = np.maximum(0.0, 2.5 + 2.5 * tri + (u[:, :, 2] - 0.5) * 2.0)
    if cold:
        n = min(T, 12)
        P[:n] = 0.0
        Tm[:n] = -15.0 - u[:n, :, 1]
        PET[: n // 2] = 0.0
    return np.stack([P, Tm, PET], axis=-1).astype(np.float32)


def raw_parameters(T: int, B: int, ny: int, seed: int, scale: float = 1.0) -> np.ndarray:
    """Raw (pre-sigmoid) NN output [T,B,ny]."""
    return (normalish((T, B, ny), seed, 3) * np.float32(scale)).astype(np.float32)


def unit_parameters(shape, seed: int, stream: int = 4) -> np.ndarray:
    """Parameters already in (0,1) (Hbv_2 tuple form, no sigmoid)."""
    return (uniform(shape, seed, stream) * np.float32(0.96) + np.float32(0.02)).astype(np.float32)


def loss_weights(shape, seed: int, stream: int = 5) -> np.ndarray:
    return normalish(shape, seed, stream)
