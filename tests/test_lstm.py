"""Sequence LSTM of the parameter network (include/hbvx_lstm.h, hydrodl2_amd/lstm.py; SURVEY.md §8f
rank 4).  Semantics are torch.nn.LSTM's, so torch's own fp32 / fp64 LSTM on the CPU is the reference:
CPU tier pins oracle/hbv_oracle.c's restatement against it, GPU tier the HIP kernels.

Tolerances (fp32 recurrence against an fp64 torch LSTM): outputs 2e-5 absolute (|h| <= 1), gradients
2e-4 relative to the largest entry of each gradient tensor."""
import pytest
import torch

from hydrodl2_amd import _abi
from hydrodl2_amd.lstm import SeqLSTM


def _reference(mod: SeqLSTM, x, gh, dtype=torch.float64):
    ref = torch.nn.LSTM(mod.input_size, mod.hidden_size).to(dtype)
    ref.load_state_dict({k: v.detach().cpu().to(dtype) for k, v in mod.state_dict().items()})
    xr = x.detach().cpu().to(dtype).requires_grad_(True)
    out, (hn, cn) = ref(xr)
    (out * gh.cpu().to(dtype)).sum().backward()
    g = {k: p.grad for k, p in ref.named_parameters()}
    g["x"] = xr.grad
    return out, hn, cn, g


def _run(device, T, B, I, H, seed=0):
    torch.manual_seed(seed)
    mod = SeqLSTM(I, H, check=True).to(device)
    x = torch.randn(T, B, I, device=device, requires_grad=True)
    gh = torch.randn(T, B, H, device=device)
    out, (hn, cn) = mod(x)
    (out * gh).sum().backward()
    ref_out, ref_hn, ref_cn, g = _reference(mod, x, gh)
    assert out.shape == (T, B, H) and hn.shape == (1, B, H) and cn.shape == (1, B, H)
    assert torch.allclose(out.cpu().double(), ref_out, rtol=0, atol=2e-5), (out.cpu().double() - ref_out).abs().max()
    assert torch.allclose(hn.cpu().double(), ref_hn, rtol=0, atol=2e-5)
    assert torch.allclose(cn.cpu().double(), ref_cn, rtol=1e-5, atol=2e-5)
    got = {k: p.grad for k, p in mod.named_parameters()}
    got["x"] = x.grad
    for k, want in g.items():
        err = (got[k].cpu().double() - want).abs().max().item()
        assert err <= 2e-4 * max(want.abs().max().item(), 1e-3), (k, err, want.abs().max().item())


@pytest.mark.parametrize("T,B,I,H", [(1, 3, 5, 8), (7, 5, 4, 16), (40, 19, 11, 64)])
def test_oracle_lstm_matches_torch(oracle_backend, T, B, I, H):
    _run("cpu", T, B, I, H)


def test_weight_dropout_matches_masked_torch_lstm(oracle_backend):
    """dr > 0 (hydroDL's CudnnLstm weight dropout): same as torch.nn.LSTM run on the masked weights, and
    the gradient reaches the parameters through the mask; eval mode ignores dr."""
    torch.manual_seed(3)
    mod = SeqLSTM(4, 8, dr=0.5)
    x = torch.randn(6, 3, 4)
    torch.manual_seed(11)
    out, _ = mod(x)
    out.sum().backward()
    torch.manual_seed(11)
    w_ih = torch.nn.functional.dropout(mod.weight_ih_l0.detach(), 0.5, training=True)
    w_hh = torch.nn.functional.dropout(mod.weight_hh_l0.detach(), 0.5, training=True)
    ref = torch.nn.LSTM(4, 8)
    ref.load_state_dict({"weight_ih_l0": w_ih, "weight_hh_l0": w_hh, "bias_ih_l0": mod.bias_ih_l0.detach(),
                         "bias_hh_l0": mod.bias_hh_l0.detach()})
    want, _ = ref(x)
    assert torch.allclose(out, want, atol=1e-5)
    g = mod.weight_hh_l0.grad
    assert torch.all(g[w_hh == 0] == 0) and g.abs().sum() > 0
    mod.eval()
    a, _ = mod(x)
    b, _ = mod(x)
    assert torch.equal(a, b)


def test_state_dict_interchanges_with_torch_lstm():
    torch.manual_seed(0)
    ref = torch.nn.LSTM(6, 64)
    mod = SeqLSTM(6, 64)
    mod.load_state_dict(ref.state_dict())
    ref.load_state_dict(mod.state_dict())
    assert [tuple(p.shape) for p in mod.parameters()] == [tuple(p.shape) for p in ref.parameters()]


def test_lstm_fails_loudly_without_gpu_tensors():
    from hydrodl2_amd import _lib
    from tests import seam
    seam.use_library(None)
    with pytest.raises(RuntimeError, match="GPU|HIP"):
        SeqLSTM(4, 64)(torch.randn(3, 2, 4))


def test_lstm_descriptor_validation():
    import __graft_entry__ as ge
    lib = _abi.Library(ge.build_hip())
    r = _abi.LstmDesc(abi_version=0, T=4, B=2, H=64)
    with pytest.raises(_abi.HbvxError, match="abi_version"):
        lib.lstm_forward(r, 1, 1, 1, 1, 1, 1, 1 << 30, 0)
    r.abi_version = _abi.LSTM_ABI_VERSION
    r.H = 48
    with pytest.raises(_abi.HbvxError, match="hidden size"):
        lib.lstm_forward(r, 1, 1, 1, 1, 1, 1, 1 << 30, 0)
    r.H = 64
    with pytest.raises(_abi.HbvxError, match="workspace"):
        lib.lstm_forward(r, 1, 1, 1, 1, 1, 1, 16, 0)
    assert lib.lstm_workspace_bytes(r) >= 4 * 1 * 64 * 16 * 4 * 4


@pytest.mark.gpu
@pytest.mark.parametrize("T,B,I,H", [(1, 1, 3, 64), (2, 16, 7, 64), (50, 23, 11, 64), (33, 40, 64, 128),
                                     (64, 100, 16, 256), (730, 100, 256, 256)])
def test_hip_lstm_matches_torch(hip_backend, T, B, I, H):
    _run("cuda", T, B, I, H)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"HBVX_LSTM_UNITS": "16"}, {"HBVX_LSTM_WGS_PER_CU": "3"}])
def test_hip_lstm_other_kernel_forms(hip_backend, monkeypatch, env):
    """The forms the host does not pick by itself at this size: 16-unit forward workgroups, three
    workgroups per CU."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _run("cuda", 90, 100, 12, 256, seed=5)
    _run("cuda", 20, 37, 5, 64, seed=6)


@pytest.mark.gpu
def test_hip_lstm_more_row_tiles_than_one_launch_holds(hip_backend):
    # 19 row tiles x 16 workgroups > 256 CUs: two workgroups per CU
    _run("cuda", 24, 300, 8, 256, seed=3)
    # 63 row tiles x 16 workgroups > 3 x 256: three per CU and two balanced launches
    _run("cuda", 12, 1000, 8, 256, seed=4)


@pytest.mark.gpu
def test_hip_lstm_is_deterministic_under_load(hip_backend):
    """Same inputs, twice, with a bandwidth-heavy kernel running beside the second call on another
    stream (uneven arrival at the hand-offs): bit-identical results."""
    torch.manual_seed(1)
    mod = SeqLSTM(32, 256, check=True).cuda()
    x = torch.randn(200, 100, 32, device="cuda")
    with torch.no_grad():
        a, _ = mod(x)
        big = torch.empty(64 << 20, device="cuda")
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(20):
                big.mul_(1.0001)
        b, _ = mod(x)
        torch.cuda.synchronize()
    assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("direction", ["forward", "backward"])
def test_handoff_timeout_is_never_silent(direction, hip_backend, monkeypatch):
    """Fault injection: one workgroup of a row tile never publishes its slab, so its partners' polls run
    out.  The launch must not return finite-but-wrong values: the affected rows carry NaN in the last
    output step (forward) / the first gradient step (backward) -- also for the waves that did NOT time
    out themselves (with 8 units per workgroup two of the four waves only multiply) -- and
    hbvx_lstm_check reports the error word."""
    import hydrodl2_amd._abi as _abi
    T, B, I, H = 6, 20, 5, 64
    torch.manual_seed(1)
    net = SeqLSTM(I, H).cuda()
    x = torch.randn(T, B, I, device="cuda", requires_grad=True)
    if direction == "backward":
        h, _ = net(x)                                    # clean forward
    monkeypatch.setenv("HBVX_LSTM_DEBUG_DROP_WG", "1")
    monkeypatch.setenv("HBVX_LSTM_SPIN_LIMIT", "3000")
    if direction == "forward":
        h, _ = net(x)
        torch.cuda.synchronize()
        assert torch.isnan(h[-1, :16]).any(), "rows of the starved tile must be poisoned"
        with pytest.raises(_abi.HbvxError, match="timed out"):
            SeqLSTM(I, H, check=True).cuda()(x)
    else:
        h.square().sum().backward()
        torch.cuda.synchronize()
        assert torch.isnan(x.grad).any() or torch.isnan(net.weight_hh_l0.grad).any()


def _two_layer(device, T, B, I, H):
    torch.manual_seed(3)
    ref = torch.nn.LSTM(I, H, num_layers=2).double()
    mod = SeqLSTM(I, H, num_layers=2)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod = mod.to(device)
    x = torch.randn(T, B, I)
    xr = x.double().requires_grad_(True)
    xo = x.to(device).requires_grad_(True)
    yr, (hr, cr) = ref(xr)
    yo, (ho, co) = mod(xo)
    assert ho.shape == hr.shape == (2, B, H)
    assert torch.allclose(yo.cpu().double(), yr, rtol=1e-4, atol=2e-5)
    assert torch.allclose(ho.cpu().double(), hr, rtol=1e-4, atol=2e-5) and torch.allclose(co.cpu().double(), cr, rtol=1e-4, atol=5e-5)
    w = torch.randn_like(yr)
    (yr * w).sum().backward()
    (yo * w.float().to(device)).sum().backward()
    for (k, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
        err = (p.grad.cpu().double() - q.grad).abs().max().item()
        assert err <= 3e-4 * max(q.grad.abs().max().item(), 1e-3), (k, err)
    assert torch.allclose(xo.grad.cpu().double(), xr.grad, rtol=1e-3, atol=1e-5)


def test_two_layers_match_torch_on_oracle(oracle_backend):
    _two_layer("cpu", 12, 5, 6, 16)


@pytest.mark.gpu
def test_two_layers_match_torch_on_gpu(hip_backend):
    _two_layer("cuda", 40, 21, 9, 64)


@pytest.mark.gpu
def test_weight_gradients_at_split_k_size_match_torch(hip_backend):
    """T * B = 9 600 rows: above the threshold where lstm._wgrad cuts the weight-gradient products into batches along
    K (and (T - 1) * B = 9 500 takes a different divisor); same tolerance as the small cases."""
    _two_layer("cuda", 96, 100, 12, 64)


def test_wgrad_split_matches_the_plain_product():
    from hydrodl2_amd.lstm import _wgrad
    torch.manual_seed(5)
    for K in (9600, 9500, 8209):          # divisible by 8 / by 10 only ... / a prime: the plain product
        a, b = torch.randn(K, 24, dtype=torch.float64), torch.randn(K, 7, dtype=torch.float64)
        assert torch.allclose(_wgrad(a, b), a.t() @ b, rtol=1e-12, atol=1e-10)
