// launch_stream_bwd.hip -- host dispatch of the streaming adjoints for large grids (hbv_stream2.h, hbv_stream2_ckpt.h;
// hbv_stream.h for forcing layouts the second generation does not take).  The forwards: launch_stream.hip.
#include "launch_stream_plan.h"
#include "hbv_stream2_ckpt.h"

using namespace hbvx;
using namespace hbvx_host;
using namespace hbvx_host::stream_plan;

namespace {

bool four_waves_pay(int64_t wgs)
{
    static int n_cu_of[64] = {0};
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        if (n_cu_of[dev] == 0) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
            n_cu_of[dev] = v;
        }
        n_cu = n_cu_of[dev];
    }
    const int64_t s3 = (int64_t)n_cu * 4 * 3, s4 = (int64_t)n_cu * 4 * 4;
    return (wgs + s4 - 1) / s4 < (wgs + s3 - 1) / s3;
}

template <int MODEL, bool BE, int SC>
void go_bwd2(int trj, bool gfull, const StreamBwdArgs &sa, dim3 grid, hipStream_t st)
{
    // the W4 form exists for the 4-series gradient of the explicit daily models (the hourly step and
    // the 12-series form spill too much at 128 registers)
    if constexpr (SC != 4) {      // (no four-wave form of the six-slot list: it would spill most of them)
        if (trj == 2 && !gfull && MODEL != MODEL_HOURLY && four_waves_pay((int64_t)sa.per_xcd * 8)) {
            hipLaunchKernelGGL((k_bwd_stream2<MODEL, BE, 2, SC, false, true, MODEL != MODEL_HOURLY>), grid, dim3(64), 0, st, sa);
            return;
        }
    }
    if (trj == 2) {
        if (gfull) hipLaunchKernelGGL((k_bwd_stream2<MODEL, BE, 2, SC, true, true>), grid, dim3(64), 0, st, sa);
        else hipLaunchKernelGGL((k_bwd_stream2<MODEL, BE, 2, SC, false, true>), grid, dim3(64), 0, st, sa);
    } else {
        if (gfull) hipLaunchKernelGGL((k_bwd_stream2<MODEL, BE, 1, SC, true, true>), grid, dim3(64), 0, st, sa);
        else hipLaunchKernelGGL((k_bwd_stream2<MODEL, BE, 1, SC, false, true>), grid, dim3(64), 0, st, sa);
    }
}

// the on-chip checkpoint adjoint (hbv_stream2_ckpt.h): LDS bytes of a wave, launch
template <int MODEL, bool BE, int SC>
void go_bwd2c(bool gfull, int K, const StreamBwdArgs &sa, dim3 grid, hipStream_t st, hipError_t *err)
{
    constexpr int NP = NParamT<MODEL, BE>::value;
    if constexpr (SC == 4) {      // (admission keeps the six-slot lists off this path: stream_ckpt_applicable)
        *err = hipErrorInvalidValue;
    } else {
        const int lds = s2c_lds_floats<NP, StreamDyn<SC>::nd>(K) * (int)sizeof(float);
        if (gfull) {
            *err = set_dynamic_lds((const void *)k_bwd_stream2_ckpt<MODEL, BE, SC, true>, lds);
            if (*err == hipSuccess) hipLaunchKernelGGL((k_bwd_stream2_ckpt<MODEL, BE, SC, true>), grid, dim3(64), lds, st, sa, K);
        } else {
            *err = set_dynamic_lds((const void *)k_bwd_stream2_ckpt<MODEL, BE, SC, false>, lds);
            if (*err == hipSuccess) hipLaunchKernelGGL((k_bwd_stream2_ckpt<MODEL, BE, SC, false>), grid, dim3(64), lds, st, sa, K);
        }
    }
}

// second-generation kernels exist for these (model, BETAET, dynamic set) combinations

} // namespace

bool hbvx_host::try_bwd_stream(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream, int *rc)
{
    // single-pass streaming adjoint, no workspace
    StreamPlan P = plan_stream(d);
    const int64_t lim = (int64_t)1 << 32;
    const bool packed = io->traj_layout == HBVX_TRAJ_PACKED;
    bool ok = P.ok && (io->grad_flux || io->grad_flux4) && (packed || P.rows_ok);
    for (int i = 0; i < d->n_param && ok; i++)
        if (d->p[i].dyn && io->g[i].dyn)
            ok = io->g[i].dyn_t_stride >= 0 && (int64_t)d->B * io->g[i].dyn_b_stride * 4 < lim &&
                 (P.sc >= 0 || ((int64_t)d->T * io->g[i].dyn_t_stride + (int64_t)d->B * io->g[i].dyn_b_stride) * 4 < lim);
    if (packed) {
        if (!(ok && P.sc >= 0 && P.packed_ok)) {
            *rc = fail(HBVX_E_UNSUPPORTED, "packed trajectory: no adjoint kernel for this call");
            return true;
        }
    } else if (!(ok && !adjoint_pinned_elsewhere() && P.wgs >= stream_min(P, true, true))) {
        return false;
    }
    StreamBwdArgs sa;
    sa.d = *d;
    sa.io = *io;
    sa.lgMp = P.lg;
    sa.nd = P.nd;
    sa.per_xcd = 0;
    for (int k = 0; k < 6; k++) sa.dslot[k] = k < P.nd ? P.dslot[k] : 0;
    const bool few = P.nd > 0, gfull = io->grad_flux != nullptr;
    dim3 grid_s((unsigned)P.wgs);
    hipStream_t st = (hipStream_t)stream;
    store_gate(io, st);          // single pass: the one kernel stores
    if (P.sc >= 0) {
        sa.per_xcd = (int)((P.wgs + 7) / 8);
        const dim3 grid2((unsigned)(8 * sa.per_xcd));
        STREAM2_DISPATCH(go_bwd2, d, P.sc, packed ? 2 : 1, gfull, sa, grid2, st);
    } else {
#define STREAM_GO(MODEL, BE)                                                                              \
    do {                                                                                                  \
        if (few) { if (gfull) hipLaunchKernelGGL((k_bwd_stream<MODEL, BE, true, true>), grid_s, dim3(64), 0, st, sa);    \
                   else hipLaunchKernelGGL((k_bwd_stream<MODEL, BE, true, false>), grid_s, dim3(64), 0, st, sa); }      \
        else { if (gfull) hipLaunchKernelGGL((k_bwd_stream<MODEL, BE, false, true>), grid_s, dim3(64), 0, st, sa);       \
               else hipLaunchKernelGGL((k_bwd_stream<MODEL, BE, false, false>), grid_s, dim3(64), 0, st, sa); }         \
    } while (0)
        if (d->model == HBVX_MODEL_HBV10 && d->n_param == 12) STREAM_GO(MODEL_HBV10, false);
        else if (d->model == HBVX_MODEL_HBV10) STREAM_GO(MODEL_HBV10, true);
        else if (d->model == HBVX_MODEL_HBV11P) STREAM_GO(MODEL_HBV11P, true);
        else if (d->model == HBVX_MODEL_HOURLY) STREAM_GO(MODEL_HOURLY, true);
        else STREAM_GO(MODEL_HBV20, true);
#undef STREAM_GO
    }
    hipError_t e = hipGetLastError();
    note_dispatch(1, P.sc >= 0 ? "stream2" : "stream");
    *rc = e != hipSuccess ? hip_fail(e, "hbvx_backward (stream) launch") : HBVX_OK;
    return true;
}

// ---------------------------------------------------------------------------------------------
// HBVX_TRAJ_CKPT on large grids: the streaming adjoint that keeps its K-day segment in LDS (hbv_stream2_ckpt.h).
// Same admission as the streaming pair on the packed trajectory -- a compiled dynamic set, forcing channels adjacent,
// one day's rows within a descriptor -- from the same grid size on (HBVX_CKPT_ONCHIP = 1 / 0: always / never, tests
// and tools).  Small grids keep the block-wise path: their adjoint wants the time-parallel kernels.
// ---------------------------------------------------------------------------------------------
bool hbvx_host::stream_ckpt_applicable(const hbvx_desc *d, int K)
{
    if (K != 4 && K != 8 && K != 16) return false;
    const int want = env_int("HBVX_CKPT_ONCHIP", -1);
    if (want == 0) return false;
    const StreamPlan P = plan_stream(d);
    if (!(P.ok && P.sc >= 0 && P.sc != 4 && P.packed_ok)) return false;
    if (want == 1) return true;
    return !adjoint_pinned_elsewhere() && P.wgs >= stream_min(P, true, true);
}

bool hbvx_host::try_bwd_stream_ckpt(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream, int *rc)
{
    const int K = HBVX_TRAJ_CKPT_DAYS(io->traj_layout);
    if (!stream_ckpt_applicable(d, K) || !(io->grad_flux || io->grad_flux4)) return false;
    const StreamPlan P = plan_stream(d);
    const int64_t lim = (int64_t)1 << 32;
    for (int i = 0; i < d->n_param; i++)
        if (d->p[i].dyn && io->g[i].dyn &&
            !(io->g[i].dyn_t_stride >= 0 && (int64_t)d->B * io->g[i].dyn_b_stride * 4 < lim))
            return false;
    StreamBwdArgs sa;
    sa.d = *d;
    sa.io = *io;
    sa.lgMp = P.lg;
    sa.nd = P.nd;
    for (int k = 0; k < 6; k++) sa.dslot[k] = k < P.nd ? P.dslot[k] : 0;
    sa.per_xcd = (int)((P.wgs + 7) / 8);
    const dim3 grid2((unsigned)(8 * sa.per_xcd));
    hipStream_t st = (hipStream_t)stream;
    store_gate(io, st);          // single pass: the one kernel stores
    hipError_t e = hipSuccess;
    const bool gfull = io->grad_flux != nullptr;
    STREAM2_DISPATCH(go_bwd2c, d, P.sc, gfull, K, sa, grid2, st, &e);
    if (e == hipSuccess) e = hipGetLastError();
    note_dispatch(1, "ckpt-stream2");
    *rc = e != hipSuccess ? hip_fail(e, "hbvx_backward (on-chip checkpoints) launch") : HBVX_OK;
    return true;
}

