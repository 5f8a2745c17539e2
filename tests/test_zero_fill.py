"""hbvx_zero (include/hbvx.h): the dense-gradient zero fill, odd sizes and unaligned starts; bytes next to
the range stay untouched."""
import pytest
import torch

from hydrodl2_amd._lib import get_library

SIZES = [0, 1, 3, 15, 16, 17, 4095, 4096, 4097, 65536 + 5, (1 << 22) + 11]


def _check(dev):
    lib = get_library()
    stream = torch.cuda.current_stream().cuda_stream if dev == "cuda" else 0
    for n in SIZES:
        for off in (0, 1, 7, 16):
            buf = torch.full((n + 64,), 0x5A, dtype=torch.uint8, device=dev)
            lib.zero(buf.data_ptr() + off, n, stream)
            if dev == "cuda":
                torch.cuda.synchronize()
            assert int(buf[off:off + n].max()) == 0 if n else True
            assert int(buf[:off].min()) == 0x5A if off else True
            assert int(buf[off + n:].min()) == 0x5A


def test_zero_fill_oracle(oracle_backend):
    _check("cpu")


@pytest.mark.gpu
def test_zero_fill_hip(hip_backend):
    _check("cuda")
