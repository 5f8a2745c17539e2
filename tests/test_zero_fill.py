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


@pytest.mark.gpu
def test_fill_workgroups_of_the_forward_launch_claim_whole_pieces_and_stop(hip_backend, monkeypatch):
    """The contract of hbvx_fwd_out.zero_ptr at the ABI's level of detail: after the forward call zero_state[1] counts the
    recurrence's workgroups (150 here), zero_state[0] the 256 KB pieces the fill waves claimed; every claimed piece is
    zero, every byte behind them still holds what the caller put there (NaN)."""
    import math
    import hydrodl2_amd
    from hydrodl2_amd import ops
    dev = torch.device("cuda:0")
    T, B, M = 640, 600, 16
    monkeypatch.setattr(ops, "_EARLY_ZERO", "1")
    real = ops._early_zero_request
    seen = []

    def spy(lib, cfg, ptensors, needs, out):
        r = real(lib, cfg, ptensors, needs, out)
        if r is not None:
            r[1].fill_(float("nan"))      # on the launch stream, in front of the forward
            seen.append(r)
        return r
    monkeypatch.setattr(ops, "_early_zero_request", spy)
    model = hydrodl2_amd.load_model("hbv", "Hbv")({"nmul": M, "dynamic_params": {"Hbv": []}}, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand((T, B, 3), generator=g, device=dev) * torch.tensor([20.0, 30.0, 5.0], device=dev) - torch.tensor([12.0, 10.0, 0.0], device=dev)
    x[..., 0].clamp_(min=0.0)
    p = torch.randn((T, B, model.learnable_param_count), generator=g, device=dev, requires_grad=True)
    out = model({"x_phy": x}, p)
    torch.cuda.synchronize()
    assert len(seen) == 1 and ops.get_library().zero_in_launch()
    _, big, state = seen[0]
    claimed, done = (int(v) for v in state.tolist())
    nbytes = big.numel() * 4
    npiece = math.ceil(nbytes / PIECE)
    assert done == (B + 3) // 4                       # one count per workgroup of the recurrence (4 basins x 16 members per wave)
    assert claimed > 0
    flat = big.view(-1)
    edge = min(claimed, npiece) * (PIECE // 4)
    assert float(flat[:edge].abs().max()) == 0.0
    if edge < flat.numel():
        assert bool(torch.isnan(flat[edge:]).all())
    # and backward completes the buffer: the gradient is finite and zero outside the last row
    out["streamflow"].sum().backward()
    assert bool(torch.isfinite(p.grad).all()) and float(p.grad[:-1].abs().max()) == 0.0 and float(p.grad[-1].abs().max()) > 0.0
