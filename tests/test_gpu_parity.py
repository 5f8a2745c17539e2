"""GPU tier (-m gpu): the HIP path, called through the C ABI, against
 (a) the golden vectors produced by the reference itself, and
 (b) the CPU oracle on seeded inputs at sizes the oracle finishes in seconds.
Tolerances (fp32; tests/abi_util.py, tests/helpers.py): fluxes and storages rtol 1e-4 with an absolute 1e-5; gradients
rtol 1e-3 with 1e-6 x the largest gradient of the same parameter group."""
import numpy as np
import pytest

from . import golden_cases as gc
from .abi_util import assert_grad_close, column_groups, compare_runs, make_problem, run_problem
from .helpers import compare, load_golden, run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(gc.CASES))
def test_golden_case_on_gpu(name, hip_backend):
    ref = load_golden(name)
    res = run_case(name, "cuda:0")
    compare(name, res, ref)


ORACLE_CASES = [
    dict(model="Hbv", T=400, B=37, M=16, dyn=()),
    dict(model="Hbv", T=300, B=21, M=16, dyn=("parBETA", "parBETAET"), drop_frac=0.3),
    dict(model="Hbv", T=200, B=130, M=1, dyn=("parK0",)),
    dict(model="Hbv", T=150, B=19, M=5, dyn=("parTT", "parFC"), muwts=True, cold=True),
    dict(model="Hbv", T=100, B=5, M=64, dyn=()),
    dict(model="Hbv_1_1p", T=250, B=23, M=16,
         dyn=tuple(gc.PHY_NAMES["Hbv_1_1p"])),
    dict(model="Hbv_2", T=250, B=40, M=8, dyn=("parBETA", "parK0", "parBETAET")),
    dict(model="Hbv_2_hourly", T=300, B=21, M=16, dyn=("parBETA", "parK0", "parF0")),
    dict(model="Hbv_2_hourly", T=200, B=9, M=4, dyn=(), cold=True),
    # smallest shapes: one basin, one member (a single active lane), T just past the pipelined
    # kernel's threshold and below it, T shorter than a tile
    dict(model="Hbv", T=33, B=1, M=1, dyn=()),
    dict(model="Hbv", T=31, B=1, M=2, dyn=("parBETA",)),
    dict(model="Hbv_2", T=3, B=2, M=3, dyn=("parK0",)),
    dict(model="Hbv", T=129, B=3, M=16, dyn=("parBETA", "parBETAET", "parK0")),
    # a record that fits one tile of the tiled adjoint, dynamic parameters with a dy_drop mask: the static
    # row (T-1) and the dynamic rows share a tensor, and the helper waves' drain of that only tile used to
    # race with the stepper's += into the static row (found by tools/fuzz_parity.py)
    dict(model="Hbv_2_hourly", T=2, B=37, M=2, drop_frac=0.3,
         dyn=tuple(n for n in gc.PHY_NAMES["Hbv_2_hourly"] if n != "parCFMAX")),
    dict(model="Hbv", T=3, B=21, M=4, drop_frac=0.5, dyn=("parBETA", "parFC", "parK1", "parTT", "parCWH")),
]


@pytest.mark.parametrize("kw", ORACLE_CASES,
                         ids=lambda k: f"{k['model']}-T{k['T']}-B{k['B']}-M{k['M']}-{len(k['dyn'])}dyn")
def test_hip_matches_oracle(kw, hip_backend, oracle_path):
    prob = make_problem(seed=7, **kw)
    got = run_problem(prob, None, device="cuda:0", x_grad=True)
    want = run_problem(prob, oracle_path, device="cpu", x_grad=True)
    compare_runs(prob, got, want)


@pytest.mark.parametrize("env", [
    {"HBVX_KERNEL": "simple"},
    {"HBVX_KT": "2", "HBVX_NH": "1"},
    {"HBVX_KT": "4", "HBVX_NH": "3"},
    {"HBVX_KT": "1", "HBVX_NH": "7"},
    {"HBVX_CHUNK": "16"},                      # time-parallel adjoint with ragged chunks
    {"HBVX_BWD": "tiled"},                     # serial tiled adjoint
], ids=lambda e: "-".join(f"{k[5:]}{v}" for k, v in e.items()))
@pytest.mark.parametrize("kw", [ORACLE_CASES[1], ORACLE_CASES[3], ORACLE_CASES[6]],
                         ids=lambda k: f"{k['model']}-M{k['M']}")
def test_kernel_variants_and_tile_shapes(kw, env, hip_backend, oracle_path, monkeypatch):
    """The one-wave kernels and odd tile geometries (tile edges, 1..7 helper waves)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    prob = make_problem(seed=11, **dict(kw, T=kw["T"] // 2 + 3))
    got = run_problem(prob, None, device="cuda:0", x_grad=True)
    want = run_problem(prob, oracle_path, device="cpu", x_grad=True)
    compare_runs(prob, got, want)


def test_warmup_offset_call(hip_backend, oracle_path):
    """t0 > 0: the main run reads rows t0.. of x / parameters in place."""
    prob = make_problem(model="Hbv", T=90, B=11, M=16, dyn=("parBETA",), seed=9)
    got = run_problem(prob, None, device="cuda:0", t0=30)
    want = run_problem(prob, oracle_path, device="cpu", t0=30)
    compare_runs(prob, got, want)
    assert np.abs(got["g_params"][:30]).max() == 0.0


def test_product_refuses_cpu_tensors(hip_backend):
    import torch
    import hydrodl2_amd
    Hbv = hydrodl2_amd.load_model("hbv", "Hbv")
    m = Hbv(None, torch.device("cpu"))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m({"x_phy": torch.rand(10, 3, 3)}, torch.randn(10, 3, 14))


def test_pow_on_gpu(hip_backend):
    """The device pow (v_log_f32 / v_exp_f32 on the reduced mantissa, hi+lo exponent product) stays
    within the stated POW_HW_MAX_ULP of exact x**y."""
    import ctypes as C
    import torch
    from .test_step_math_host import POW_HW_MAX_ULP, _pow_inputs, pow_error_ulps
    x, y = _pow_inputs()
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    out = torch.empty_like(xd)
    fn = hip_backend.dll.hbvx_selftest_pow
    fn.restype = C.c_int
    rc = fn(C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), C.c_void_p(out.data_ptr()),
            C.c_int(x.size), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    err = pow_error_ulps(out.cpu().numpy(), x, y)
    print(f"device pow: max {err.max():.3f} ulp, mean {err.mean():.3f} ulp")
    assert err.max() <= POW_HW_MAX_ULP, f"max error {err.max():.3f} ulp"
    assert out[0].item() == 1.0


def test_div_on_gpu(hip_backend):
    """The device quotient (rcp + two fused corrections) equals IEEE a/b on the path's operand
    range, and is exactly 1 for a == b (the clamp gradients at SM == FC rely on it)."""
    import ctypes as C
    import torch
    from . import synth
    n = 1 << 20
    a = (10.0 ** (synth.uniform((n,), 91, 1).astype(np.float64) * 8 - 5)).astype(np.float32)
    b = (10.0 ** (synth.uniform((n,), 91, 2).astype(np.float64) * 3)).astype(np.float32)
    a[: n // 8] = b[: n // 8]
    ad, bd = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    out = torch.empty_like(ad)
    fn = hip_backend.dll.hbvx_selftest_div
    fn.restype = C.c_int
    assert fn(C.c_void_p(ad.data_ptr()), C.c_void_p(bd.data_ptr()), C.c_void_p(out.data_ptr()),
              C.c_int(n), C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    got = out.cpu().numpy()
    want = a / b
    assert (got[: n // 8] == 1.0).all()
    mism = int((got != want).sum())
    assert mism == 0, f"{mism} of {n} quotients differ from IEEE division"


@pytest.mark.parametrize("model,dyn,betaet,drop", [
    ("Hbv", (), False, 0.0), ("Hbv", ("parBETA", "parBETAET"), True, 0.0),
    ("Hbv", ("parK0", "parTT", "parFC"), False, 0.4),
    ("Hbv_1_1p", (), False, 0.0), ("Hbv_1_1p", ("parC", "parK2", "parBETAET"), False, 0.3),
    ("Hbv_2", ("parBETA", "parK0", "parBETAET"), False, 0.0), ("Hbv_2", ("parRT", "parAC"), False, 0.5),
    ("Hbv_2_hourly", (), False, 0.0), ("Hbv_2_hourly", ("parBETA", "parF0", "parALPHA"), False, 0.3),
    # more than three dynamic parameters: 4-day tiles, several staged rows per filler wave
    ("Hbv", tuple(gc.PHY_NAMES["Hbv"]) + ("parBETAET",), True, 0.3),
    ("Hbv_1_1p", tuple(gc.PHY_NAMES["Hbv_1_1p"]), False, 0.0),
    ("Hbv_2", ("parBETA", "parFC", "parK0", "parC", "parRT"), False, 0.2),
    ("Hbv_2_hourly", ("parBETA", "parK0", "parBETAET", "parF0", "parFMIN", "parALPHA", "parTT"), False, 0.0)])
@pytest.mark.parametrize("M,B,T", [(16, 37, 411), (1, 130, 97), (5, 19, 64)])
def test_pipelined_forward_equals_tiled_forward(M, B, T, model, dyn, betaet, drop, hip_backend, monkeypatch):
    """Up to three dynamic parameters: the pipelined kernel (three stages for HBV 1.0, two for the
    capillary models) and the tiled kernel are bit-identical (same stage arithmetic, same
    ensemble-sum order), with and without the saved trajectory."""
    prob = make_problem(model=model, T=T, B=B, M=M, dyn=dyn, betaet=betaet, drop_frac=drop, seed=21,
                        cold=True)
    a = run_problem(prob, None, device="cuda:0", backward=True)
    a2 = run_problem(prob, None, device="cuda:0", backward=False)
    monkeypatch.setenv("HBVX_FWD", "tiled")
    b = run_problem(prob, None, device="cuda:0", backward=True)
    for k in ("flux", "routed", "state_out", "traj", "g_params"):
        if k in b:      # the hourly model has no 15-tap routing
            assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a2["flux"], b["flux"])


@pytest.mark.parametrize("model,dyn,M,B", [
    ("Hbv", (), 16, 8200), ("Hbv", ("parBETA", "parBETAET"), 16, 8200),
    ("Hbv_1_1p", ("parK0", "parTT", "parFC"), 5, 16400), ("Hbv_2", ("parBETA",), 16, 8200),
    ("Hbv_2_hourly", ("parBETA", "parK0", "parBETAET"), 4, 32800),
    ("Hbv", (), 1, 131200), ("Hbv_2", (), 2, 65600), ("Hbv", ("parK1",), 64, 2050)])
def test_stream_forward_equals_tiled_forward(model, dyn, M, B, hip_backend, monkeypatch):
    """Large grids (>= 2048 wavefronts of state here, so that both directions take it) run the
    streaming one-wave kernels (hbv_stream.h).  The forward is bit-identical to the tiled /
    pipelined kernels (same step arithmetic, same ensemble add tree), with and without the saved
    trajectory; the single-pass adjoint is bit-identical to the serial tiled adjoint (T < 128)."""
    T = 37
    prob = make_problem(model=model, T=T, B=B, M=M, dyn=dyn, drop_frac=0.3 if dyn else 0.0, seed=33)
    a = run_problem(prob, None, device="cuda:0", backward=True)
    a2 = run_problem(prob, None, device="cuda:0", backward=False)
    monkeypatch.setenv("HBVX_STREAM", "0")
    b = run_problem(prob, None, device="cuda:0", backward=True)
    for k in ("flux", "state_out", "traj"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a2["flux"], b["flux"])
    # the streaming adjoint applies a static parameter's range factor once, after the sum over days
    # (the tiled one per day): same terms, rounding differs in the last bits
    assert_grad_close("g_params", a["g_params"], b["g_params"], column_groups(prob["ny"], M), rtol=2e-5, atol_rel=1e-6)


STREAM2_CASES = [
    dict(model="Hbv", T=140, B=9, M=16, dyn=()),
    dict(model="Hbv", T=133, B=10, M=16, dyn=("parBETA", "parBETAET"), drop_frac=0.3),
    dict(model="Hbv_1_1p", T=90, B=21, M=5, dyn=("parBETA", "parBETAET")),
    dict(model="Hbv_1_1p", T=64, B=7, M=16, dyn=()),
    dict(model="Hbv_2", T=150, B=11, M=16, dyn=("parBETA", "parK0", "parBETAET"), drop_frac=0.4),
    dict(model="Hbv_2", T=70, B=70, M=2, dyn=()),
    dict(model="Hbv_2_hourly", T=100, B=19, M=4, dyn=("parBETA", "parK0", "parBETAET")),
    dict(model="Hbv", T=50, B=5, M=64, dyn=()),
    dict(model="Hbv", T=45, B=130, M=1, dyn=("parBETA", "parBETAET")),
    # run-time slot lists (hbv_stream2.h, SC == 3): the sets without a compiled instance -- one, two and three
    # parameters, with and without BETAET in the table, dy_drop masks -- and permuted forcing channels
    dict(model="Hbv", T=140, B=9, M=16, dyn=("parBETA",)),
    dict(model="Hbv", T=133, B=10, M=16, dyn=("parK0", "parBETAET"), drop_frac=0.3),
    dict(model="Hbv_1_1p", T=90, B=21, M=5, dyn=("parK0", "parTT", "parFC"), drop_frac=0.4),
    dict(model="Hbv_2", T=150, B=11, M=16, dyn=("parBETA", "parBETAET")),
    dict(model="Hbv_2", T=70, B=70, M=2, dyn=("parAC", "parRT", "parC"), drop_frac=0.3),
    dict(model="Hbv_2_hourly", T=100, B=19, M=4, dyn=("parF0", "parALPHA")),
    dict(model="Hbv", T=77, B=12, M=16, dyn=("parBETA", "parBETAET"), channels=(2, 0, 1)),
    dict(model="Hbv_2", T=64, B=9, M=8, dyn=("parFC",), channels=(1, 2, 0)),
    dict(model="Hbv", T=60, B=6, M=16, dyn=(), channels=(0, 2, 1)),
    # four to six slots (SC == 4)
    dict(model="Hbv", T=140, B=9, M=16, dyn=("parBETA", "parFC", "parK0", "parLP", "parBETAET"), drop_frac=0.3),
    dict(model="Hbv_1_1p", T=90, B=21, M=5, dyn=("parK0", "parTT", "parFC", "parC")),
    dict(model="Hbv_2", T=150, B=11, M=16, dyn=("parBETA", "parK0", "parBETAET", "parRT", "parAC", "parUZL"), drop_frac=0.4),
    dict(model="Hbv_2_hourly", T=100, B=19, M=4, dyn=("parF0", "parALPHA", "parFMIN", "parBETA")),
]


@pytest.mark.parametrize("layout_env", [{"HBVX_STREAM_MIN": "1"},
                                        {"HBVX_STREAM_MIN": "1", "HBVX_STREAM_MW_MIN": "1"},
                                        {"HBVX_STREAM_MIN": "1", "HBVX_BWD": "tiled"},
                                        {"HBVX_STREAM_MIN": "1", "HBVX_STREAM_SLOTLIST": "1"},
                                        {"HBVX_STREAM_MIN": "1", "HBVX_CKPT_DAYS": "4", "HBVX_CKPT_ONCHIP": "1"}],
                         ids=["packed", "packed-8wave", "rows", "slotlist", "ckpt4-onchip"])
@pytest.mark.parametrize("kw", STREAM2_CASES, ids=lambda k: f"{k['model']}-B{k['B']}-M{k['M']}-{len(k['dyn'])}dyn")
def test_stream2_matches_oracle(kw, layout_env, hip_backend, oracle_path, monkeypatch):
    """Second-generation streaming kernels (hbv_stream2.h) forced onto small problems: packed
    trajectory + streaming adjoint (one wave per workgroup, and the eight-wave workgroups that stage
    the flux rows in LDS: partly filled workgroups, records shorter and longer than a flux tile), and
    row trajectory + the tiled adjoint reading it."""
    for k, v in layout_env.items():
        monkeypatch.setenv(k, v)
    prob = make_problem(seed=21, **kw)
    got = run_problem(prob, None, device="cuda:0", x_grad=True)
    fwd, bwd = hip_backend.last_dispatch(0), hip_backend.last_dispatch(1)
    want = run_problem(prob, oracle_path, device="cpu", x_grad=True)
    compare_runs(prob, got, want)
    # every set of at most three dynamic parameters, any order of three adjacent channels: the second generation
    want_bwd = "tiled" if "HBVX_BWD" in layout_env else ("ckpt-stream2" if "HBVX_CKPT_DAYS" in layout_env else "stream2")
    if want_bwd == "ckpt-stream2" and len(kw["dyn"]) > 3:
        want_bwd = "ckpt-block:stream2"     # the six-slot lists have no on-chip checkpoint form: block-wise re-materialisation
    assert (fwd, bwd) == ("stream2", want_bwd), (fwd, bwd)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,width,r0,r1,gw,keep", [
    (37, 226, 5, 30, 16, 0x3FFF),          # config-3 shape: 14 kept groups, 2 routing columns left
    (40, 210, 0, 40, 16, (1 << 0) | (1 << 12)),   # two kept groups of 13: dense fill (under half kept)
    (64, 48, 0, 64, 16, 0b111),            # hbv_2 dynamic tensor: everything kept
    (19, 50, 3, 3, 5, 0b1011),             # empty row range
    (23, 41, 2, 21, 3, 0x2AAA5),           # odd widths, alternating groups, a partial last group
])
def test_zero_except_matches_definition(rows, width, r0, r1, gw, keep, hip_backend):
    """hbvx_zero_except: every element zero except the kept column groups on rows [r0, r1), which
    keep their previous contents (include/hbvx.h).  NaN marks 'previous contents'."""
    import torch
    buf = torch.full((rows, width), float("nan"), device="cuda")
    hip_backend.zero_except(buf.data_ptr(), rows, width, r0, r1, gw, keep, torch.cuda.current_stream().cuda_stream)
    got = buf.cpu().numpy()
    r = np.arange(rows)[:, None]
    g = (np.arange(width) // gw)[None, :]
    kept = (r >= r0) & (r < r1) & (g < 32) & (((keep >> np.minimum(g, 31)) & 1) == 1)
    dense = 2 * int(kept[r0].sum() if r1 > r0 else 0) < width or r1 <= r0
    if dense:                     # the library may fall back to the dense fill: all zero
        assert (got == 0).all()
    else:
        assert np.isnan(got[kept]).all() and (got[~kept] == 0).all()


@pytest.mark.parametrize("name,K", [("hbv_dyn2", 8), ("hbv_static_m16", 16), ("hbv11p_dyn_all", 4), ("hbv2_dyn3", 8),
                                    ("hbv_warmup_states", 4), ("hbv_m3_xgrad", 8), ("hbv_muwts", 8),
                                    ("hourly_dyn3", 16), ("hbv2_dyn3_routing", 4)])
@pytest.mark.parametrize("how", ["block512", "block16", "lds", "onchip"])
def test_checkpointed_adjoint_on_gpu(name, K, how, hip_backend, monkeypatch):
    """HBVX_TRAJ_CKPT on the GPU against the reference's fixtures: the streaming forward writes the
    checkpoints where it has an instance (the delta-MG dynamic sets), the generic one-wave forward
    elsewhere (the pipelined one on small grids); the adjoint runs block-wise -- k_ckpt_remat rebuilds a
    block of the trajectory, the regular adjoint kernels consume it -- or, without scratch, as k_bwd_ckpt
    with its K-day segment in LDS."""
    monkeypatch.setenv("HBVX_CKPT_DAYS", str(K))
    if how == "block16":      # several blocks: the adjoint is carried from block to block
        monkeypatch.setenv("HBVX_CKPT_BLOCK", "16")
    if how == "lds":          # the serial fallback that needs no scratch
        monkeypatch.setenv("HBVX_CKPT_BLOCKWISE", "0")
    if how == "onchip":       # the streaming adjoint with its segment in LDS, where it has an instance (else the block path)
        monkeypatch.setenv("HBVX_CKPT_ONCHIP", "1")
    ref = load_golden(name)
    res = run_case(name, "cuda:0")
    if "states" in res and res["states"].shape != ref["states"].shape:
        res["states"] = ref["states"]
    compare(name, res, ref)


@pytest.mark.parametrize("model", ["Hbv", "Hbv_1_1p", "Hbv_2", "Hbv_2_hourly"])
@pytest.mark.parametrize("T,B,M", [(1, 1, 1), (1, 3, 16), (2, 1, 64), (3, 2, 5), (15, 1, 2), (16, 5, 1), (31, 2, 16), (33, 70, 3)])
def test_degenerate_shapes_match_oracle(model, T, B, M, hip_backend, oracle_path):
    """One day, one basin, one member, records shorter than a tile / the routing window / the chunk
    length: every dispatch fallback (one-wave kernels, serial tiled adjoint) against the oracle."""
    dyn = {"Hbv": ("parBETA", "parBETAET"), "Hbv_1_1p": ("parK0",), "Hbv_2": ("parBETA", "parK0", "parBETAET"),
           "Hbv_2_hourly": ()}[model]
    prob = make_problem(model=model, T=T, B=B, M=M, dyn=dyn, drop_frac=0.3 if dyn else 0.0, seed=T * 100 + B)
    got = run_problem(prob, None, device="cuda:0", x_grad=True)
    want = run_problem(prob, oracle_path, device="cpu", x_grad=True)
    compare_runs(prob, got, want)


# Which kernel families a long fixture must have run through (hbvx_last_dispatch), per environment.  The golden
# cases above all have T <= 120: their adjoint is k_bwd_tiled.  These have T >= 256 and a loss, so the time-parallel
# adjoint (64-day chunks), the streaming pair on the packed trajectory and the checkpointed adjoint are compared with
# the reference's own autograd tape (hbv.py:423-553 taped), not only with the oracle.
LONG_ENVS = {
    "default": ({}, {"chunked"}),
    "stream2-packed": ({"HBVX_STREAM_MIN": "1"}, {"stream2"}),
    "stream2-8wave": ({"HBVX_STREAM_MIN": "1", "HBVX_STREAM_MW_MIN": "1"}, {"stream2"}),
    "stream-rows-tiled": ({"HBVX_STREAM_MIN": "1", "HBVX_BWD": "tiled"}, {"tiled"}),
    "ckpt8-blocks": ({"HBVX_CKPT_DAYS": "8", "HBVX_CKPT_BLOCK": "128"}, {"ckpt-block:chunked"}),
    "ckpt4-lds": ({"HBVX_CKPT_DAYS": "4", "HBVX_CKPT_BLOCKWISE": "0"}, {"ckpt-lds"}),
    "ckpt16-stream": ({"HBVX_CKPT_DAYS": "16", "HBVX_STREAM_MIN": "1"}, None),
    # the streaming adjoint that keeps its K-day segment in LDS (hbv_stream2_ckpt.h), forced onto these small grids
    "ckpt4-onchip": ({"HBVX_CKPT_DAYS": "4", "HBVX_CKPT_ONCHIP": "1"}, {"ckpt-stream2"}),
    "ckpt8-onchip": ({"HBVX_CKPT_DAYS": "8", "HBVX_CKPT_ONCHIP": "1"}, {"ckpt-stream2"}),
    "ckpt16-onchip": ({"HBVX_CKPT_DAYS": "16", "HBVX_CKPT_ONCHIP": "1"}, {"ckpt-stream2"}),
}


# learned ensemble weights: the streaming kernels do not carry them, so where the grid size is forced the pipelined / tiled
# forward and the time-parallel adjoint still run; checkpoints: block-wise through the time-parallel adjoint
MUWTS_BWD = {"stream2-packed": {"chunked"}, "stream2-8wave": {"chunked"}, "ckpt4-onchip": {"ckpt-block:chunked"},
             "ckpt8-onchip": {"ckpt-block:chunked"}, "ckpt16-onchip": {"ckpt-block:chunked"}}


@pytest.mark.parametrize("env_id", list(LONG_ENVS))
@pytest.mark.parametrize("name", gc.LONG_CASES)
def test_long_golden_case_under_every_adjoint(name, env_id, hip_backend, monkeypatch):
    env, want_bwd = LONG_ENVS[env_id]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ref = load_golden(name)
    res = run_case(name, "cuda:0")
    fwd, bwd = hip_backend.last_dispatch(0), hip_backend.last_dispatch(1)
    if "states" in res and res["states"].shape != ref["states"].shape:
        res["states"] = ref["states"]        # Hbv_2 under adjoint_checkpoint: the series is not kept
    compare(name, res, ref)
    model = gc.CASES[name]["model"]
    dyn = tuple(gc.CASES[name]["config"]["dynamic_params"][model])
    # hbv_stream2.h holds every set of at most three dynamic parameters (two compiled sets, the rest as a run-time slot
    # list: launch_stream.hip::plan_stream); more than three stay on the first generation / the time-parallel adjoint
    has_stream2 = len(dyn) <= 6
    if want_bwd == {"stream2"} and not has_stream2:
        want_bwd = {"chunked", "stream"}     # no second-generation instance for this dynamic set
    if want_bwd == {"ckpt-stream2"} and len(dyn) > 3:
        want_bwd = {"ckpt-block:chunked"}    # the on-chip checkpoint kernel holds the three-slot lists (launch_stream_bwd.hip)
    if gc.CASES[name].get("muwts"):
        want_bwd = MUWTS_BWD.get(env_id, want_bwd)
    if model == "Hbv_2_hourly" and "HBVX_CKPT_DAYS" in env:
        # the hourly class always keeps the state series (its routing and its state cache read them): no checkpoints,
        # the trajectory adjoints run -- the time-parallel one, or the streaming pair where the grid size is forced
        want_bwd = {"stream2"} if "HBVX_STREAM_MIN" in env else {"chunked"}
    if want_bwd is not None:
        assert bwd in want_bwd, f"{name} [{env_id}]: adjoint ran {bwd!r} (forward {fwd!r}), meant {want_bwd}"
    print(f"{name} [{env_id}]: forward {fwd}, adjoint {bwd}")


@pytest.mark.parametrize("model,dyn,drop", [("Hbv", (), 0.0), ("Hbv", ("parBETA", "parBETAET"), 0.3),
                                            ("Hbv", ("parBETA", "parK0", "parBETAET"), 0.0), ("Hbv_1_1p", ("parC",), 0.0),
                                            ("Hbv_2", ("parBETA", "parK0", "parBETAET", "parRT"), 0.2)])
@pytest.mark.parametrize("M,B,T", [(16, 37, 211), (5, 19, 64)])
def test_pipelined_forward_with_ensemble_weights_equals_tiled(M, B, T, model, dyn, drop, hip_backend, oracle_path, monkeypatch):
    """Learned ensemble weights `muwts` (hbv.py:508-511) in the pipelined forward: the weights travel as one more
    staged row (few / compiled-set / many-row instances: 1, 3, 4, 2 and 5 rows here), the stage that forms Qsim
    multiplies, the reducers sum that series.  Bit-identical to the tiled kernel (round 4: muwts forced the slow one),
    and equal to the oracle."""
    prob = make_problem(model=model, T=T, B=B, M=M, dyn=dyn, betaet=("parBETAET" in dyn and model == "Hbv"), drop_frac=drop,
                        seed=27, muwts=True)
    a = run_problem(prob, None, device="cuda:0", backward=True)
    assert hip_backend.last_dispatch(0) == "pipe"
    monkeypatch.setenv("HBVX_FWD", "tiled")
    b = run_problem(prob, None, device="cuda:0", backward=True)
    assert hip_backend.last_dispatch(0) == "tiled"
    for k in ("flux", "routed", "state_out", "traj", "g_params", "g_muwts"):
        if k in b:
            assert np.array_equal(a[k], b[k]), k
    monkeypatch.delenv("HBVX_FWD")
    want = run_problem(prob, oracle_path, device="cpu", backward=True)
    compare_runs(prob, a, want)
