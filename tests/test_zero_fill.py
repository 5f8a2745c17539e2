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


PIECE = 256 * 1024        # HBVX_ZERO_PIECE (include/hbvx.h)


@pytest.mark.gpu
@pytest.mark.parametrize("nbytes", [16, PIECE - 16, PIECE, PIECE + 16, 3 * PIECE + 48, 5 * PIECE + 7 * 16 + 5])
def test_zero_rest_fills_from_the_first_missing_piece(nbytes, hip_backend):
    """hbvx_zero_rest (ABI 10): pieces 0 .. zero_state[0]-1 belong to the forward's launch and are left alone, everything
    behind them -- whole pieces, the ragged last one, a tail shorter than 16 bytes -- becomes zero, bytes next to the
    buffer stay untouched; a count beyond the buffer's pieces (fill waves that found no piece left) is everything done."""
    lib = get_library()
    stream = torch.cuda.current_stream().cuda_stream
    npiece = (nbytes // 16 * 16 + PIECE - 1) // PIECE
    for claimed in sorted({0, 1, max(npiece - 1, 0), npiece, npiece + 37}):
        buf = torch.full((nbytes + 64,), 0x5A, dtype=torch.uint8, device="cuda")
        state = torch.tensor([claimed, 0], dtype=torch.int32, device="cuda")
        lib.zero_rest(buf.data_ptr(), nbytes, state.data_ptr(), stream)
        torch.cuda.synchronize()
        kept = min(claimed * PIECE, nbytes // 16 * 16)
        assert int(buf[:kept].min()) == 0x5A if kept else True
        assert int(buf[kept:nbytes].max()) == 0 if nbytes > kept else True
        assert int(buf[nbytes:].min()) == 0x5A


def test_zero_rest_on_the_oracle_is_a_plain_fill(oracle_backend):
    lib = get_library()
    buf = torch.full((1000 + 64,), 0x5A, dtype=torch.uint8)
    state = torch.zeros(2, dtype=torch.int32)
    lib.zero_rest(buf.data_ptr(), 1000, state.data_ptr(), 0)
    assert int(buf[:1000].max()) == 0 and int(buf[1000:].min()) == 0x5A
