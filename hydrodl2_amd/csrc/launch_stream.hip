// launch_stream.hip -- host dispatch of the streaming forward / adjoint for large grids
// (hbv_stream2.h; hbv_stream.h for dynamic sets or forcing layouts the second generation does not
// instantiate).
#include "hbvx_host.h"
#include "hbv_stream2.h"
#include "hbv_stream2_ckpt.h"

using namespace hbvx;
using namespace hbvx_host;

namespace {

struct StreamPlan {
    bool ok;        // the streaming family can take this problem at all
    int lg;
    int64_t wgs;    // wavefronts of state
    int nd;
    int sc;         // dynamic set of hbv_stream2.h: 0 none, 1 / 2 compiled sets, 3 run-time list of <= 3 slots; -1: first generation only
    bool xvec;      // three adjacent forcing channels
    bool gen1_ok;   // first generation: whole dynamic tensors within 32-bit offsets
    bool rows_ok;   // row trajectory: every offset fits 32 bits
    bool packed_ok; // packed trajectory / checkpoints: one day's rows fit 32 bits (descriptors rebased per day)
    int dslot[6];
};

StreamPlan plan_stream(const hbvx_desc *d)
{
    StreamPlan P{};
    P.lg = lg_members(d->M);
    const int bpw = 64 >> P.lg;
    P.wgs = ((int64_t)d->B + bpw - 1) / bpw;
    const int64_t N = (int64_t)d->B * d->M, lim = (int64_t)1 << 32;
    P.nd = count_dyn(d);
    const int nf = (d->model == HBVX_MODEL_HBV10) ? 11 : 12;
    P.ok = use_tiled(d) && env_int("HBVX_STREAM", 1) != 0 && P.nd <= STREAM2_LIST_MAX && !d->muwts && d->T > 0 &&
           (int64_t)nf * d->T * d->B * 4 < lim &&
           ((int64_t)d->T * d->x_t_stride + (int64_t)d->B * d->x_b_stride) * 4 < lim;
    P.gen1_ok = true;
    unsigned mask = 0;
    int k = 0;
    for (int i = 0; i < d->n_param; i++)
        if (d->p[i].dyn) {
            mask |= 1u << i;
            if (k < 6) P.dslot[k++] = i;
            // one day's row per descriptor (the kernels rebase it every day): the tensor itself may exceed 4 GiB
            P.gen1_ok = P.gen1_ok && ((int64_t)d->T * d->p[i].dyn_t_stride + (int64_t)d->B * d->p[i].dyn_b_stride) * 4 < lim;
            P.ok = P.ok && d->p[i].dyn_t_stride >= 0 && (int64_t)d->B * d->p[i].dyn_b_stride * 4 < lim;
        }
    const bool be = d->n_param >= 13;
    P.sc = mask == 0 ? 0
         : (mask == ((1u << P_BETA) | (1u << P_BETAET)) && be &&
            (d->model == HBVX_MODEL_HBV10 || d->model == HBVX_MODEL_HBV11P)) ? 1
         : (mask == ((1u << P_BETA) | (1u << P_K0) | (1u << P_BETAET)) &&
            (d->model == HBVX_MODEL_HBV20 || d->model == HBVX_MODEL_HOURLY)) ? 2 : (P.nd <= 3 ? 3 : 4);   // (nd <= 6: P.ok)
    if (env_int("HBVX_STREAM_SLOTLIST", 0) && P.sc > 0 && P.sc < 3) P.sc = 3;   // tests: the run-time list on a compiled set
    const int c0 = d->ch_prcp, c1 = d->ch_tmean, c2 = d->ch_pet;
    // second generation: a basin's three forcing values adjacent (one 12-byte load), the channels any permutation of
    // {0, 1, 2} (the kernels pick; config key `variables`)
    P.xvec = (unsigned)c0 < 3u && (unsigned)c1 < 3u && (unsigned)c2 < 3u && c0 != c1 && c0 != c2 && c1 != c2 &&
             d->x_b_stride >= 3;
    if (!P.xvec) P.sc = -1;
    P.rows_ok = 5 * (int64_t)(d->T + 1) * N * 4 < lim;
    P.packed_ok = N * 20 < lim;
    if (P.sc < 0) P.ok = P.ok && P.gen1_ok && P.rows_ok && P.nd <= 3;     // (the first generation lists three slots)
    return P;
}

// Grid size (wavefronts of state) from which the streaming kernels run.  Measured on MI355X with
// tools/grid_sweep.py (profiles/r02_grid_sweep.jsonl): second generation -- a training step (packed
// trajectory, both directions streaming) wins from 768 wavefronts for hbv and ties there for hbv_2;
// forward alone the pipelined kernel holds on to 1024.  First generation (the dynamic sets without a
// compiled instance): forward from 1024, adjoint from 2048 (round-1 measurements).
// HBVX_STREAM_MIN overrides all of them at once; HBVX_BWD=<family> pins the adjoint for tests.
int stream_min(const StreamPlan &P, bool training, bool adjoint)
{
    const int dflt = P.sc >= 0 ? (training ? 768 : 1024) : (adjoint ? 2048 : 1024);
    return env_int("HBVX_STREAM_MIN", dflt);
}
bool adjoint_pinned_elsewhere()
{
    const char *v = getenv("HBVX_BWD");
    return v && strcmp(v, "stream") != 0;
}

template <int MODEL, bool BE, int SC>
void go_fwd2(int trj, const StreamArgs &sa, dim3 grid, hipStream_t st)
{
    // 16-member ensembles: workgroups of eight waves that stage the flux rows in LDS (hbv_stream2.h, MW);
    // the row trajectory (not what hbvx_preferred_traj_layout asks for) keeps the one-wave form.  Eight-wave
    // workgroups leave CUs empty on small grids: measured cross-over between 1024 and 1536 wavefronts
    // (profiles/r02_grid_sweep.jsonl: hbv forward 0.70 / 0.90 / 1.15 / 2.18 ms with one wave per workgroup
    // against 0.81 / 0.86 / 0.95 / 1.87 ms at 1024 / 1536 / 2048 / 4096 wavefronts)
    const int mw_min = env_int("HBVX_STREAM_MW_MIN", 1280);
    if (sa.lgMp == 4 && trj != 1 && (int64_t)sa.per_xcd * 8 >= mw_min) {
        StreamArgs sm = sa;
        const int64_t waves = (int64_t)sa.per_xcd * 8;
        sm.per_xcd = (int)(((waves + 7) / 8 + 7) / 8);
        const dim3 gm((unsigned)(8 * sm.per_xcd));
        if (trj == 3) hipLaunchKernelGGL((k_fwd_stream2<MODEL, BE, 3, SC, true, 8>), gm, dim3(512), 0, st, sm);
        else if (trj == 2) hipLaunchKernelGGL((k_fwd_stream2<MODEL, BE, 2, SC, true, 8>), gm, dim3(512), 0, st, sm);
        else hipLaunchKernelGGL((k_fwd_stream2<MODEL, BE, 0, SC, true, 8>), gm, dim3(512), 0, st, sm);
        return;
    }
    if (trj == 3) hipLaunchKernelGGL((k_fwd_stream2<MODEL, BE, 3, SC, true>), grid, dim3(64), 0, st, sa);
    else if (trj == 2) hipLaunchKernelGGL((k_fwd_stream2<MODEL, BE, 2, SC, true>), grid, dim3(64), 0, st, sa);
    else if (trj == 1) hipLaunchKernelGGL((k_fwd_stream2<MODEL, BE, 1, SC, true>), grid, dim3(64), 0, st, sa);
    else hipLaunchKernelGGL((k_fwd_stream2<MODEL, BE, 0, SC, true>), grid, dim3(64), 0, st, sa);
}

// true when four waves per SIMD hold the grid in fewer rounds than three (hbv_stream2.h, W4)
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
#define STREAM2_DISPATCH(GO, d, sc, ...)                                                             \
    do {                                                                                             \
        const int m_ = (d)->model;                                                                   \
        const bool be_ = (d)->n_param == 13;                                                         \
        if ((sc) == 4) {                                                                             \
            if (m_ == HBVX_MODEL_HBV10 && !be_) GO<MODEL_HBV10, false, 4>(__VA_ARGS__);              \
            else if (m_ == HBVX_MODEL_HBV10) GO<MODEL_HBV10, true, 4>(__VA_ARGS__);                  \
            else if (m_ == HBVX_MODEL_HBV11P) GO<MODEL_HBV11P, true, 4>(__VA_ARGS__);                \
            else if (m_ == HBVX_MODEL_HBV20) GO<MODEL_HBV20, true, 4>(__VA_ARGS__);                  \
            else GO<MODEL_HOURLY, true, 4>(__VA_ARGS__);                                             \
        }                                                                                            \
        else if (m_ == HBVX_MODEL_HBV10 && !be_ && (sc) == 3) GO<MODEL_HBV10, false, 3>(__VA_ARGS__); \
        else if (m_ == HBVX_MODEL_HBV10 && !be_) GO<MODEL_HBV10, false, 0>(__VA_ARGS__);             \
        else if (m_ == HBVX_MODEL_HBV10 && (sc) == 0) GO<MODEL_HBV10, true, 0>(__VA_ARGS__);         \
        else if (m_ == HBVX_MODEL_HBV10 && (sc) == 3) GO<MODEL_HBV10, true, 3>(__VA_ARGS__);         \
        else if (m_ == HBVX_MODEL_HBV10) GO<MODEL_HBV10, true, 1>(__VA_ARGS__);                      \
        else if (m_ == HBVX_MODEL_HBV11P && (sc) == 0) GO<MODEL_HBV11P, true, 0>(__VA_ARGS__);       \
        else if (m_ == HBVX_MODEL_HBV11P && (sc) == 3) GO<MODEL_HBV11P, true, 3>(__VA_ARGS__);       \
        else if (m_ == HBVX_MODEL_HBV11P) GO<MODEL_HBV11P, true, 1>(__VA_ARGS__);                    \
        else if (m_ == HBVX_MODEL_HBV20 && (sc) == 0) GO<MODEL_HBV20, true, 0>(__VA_ARGS__);         \
        else if (m_ == HBVX_MODEL_HBV20 && (sc) == 3) GO<MODEL_HBV20, true, 3>(__VA_ARGS__);         \
        else if (m_ == HBVX_MODEL_HBV20) GO<MODEL_HBV20, true, 2>(__VA_ARGS__);                      \
        else if ((sc) == 0) GO<MODEL_HOURLY, true, 0>(__VA_ARGS__);                                  \
        else if ((sc) == 3) GO<MODEL_HOURLY, true, 3>(__VA_ARGS__);                                  \
        else GO<MODEL_HOURLY, true, 2>(__VA_ARGS__);                                                 \
    } while (0)

} // namespace

// include/hbvx.h: the trajectory layout hbvx_forward / hbvx_backward want for this problem
extern "C" int hbvx_preferred_traj_layout(const hbvx_desc *d)
{
    if (!d || check_desc(d) || d->model == HBVX_MODEL_HBVADJ) return HBVX_TRAJ_ROWS;
    const StreamPlan P = plan_stream(d);
    const bool stream_both = P.ok && P.sc >= 0 && P.packed_ok && !adjoint_pinned_elsewhere() &&
                             P.wgs >= stream_min(P, true, true);
    return stream_both ? HBVX_TRAJ_PACKED : HBVX_TRAJ_ROWS;
}

bool hbvx_host::try_fwd_stream(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream, int *rc, bool any_size)
{
    const StreamPlan P = plan_stream(d);
    const bool packed = out->traj && out->traj_layout == HBVX_TRAJ_PACKED;
    const bool ckpt = out->traj && HBVX_TRAJ_KIND(out->traj_layout) == HBVX_TRAJ_CKPT;
    if (ckpt) {
        // checkpoints: the second generation only; the offsets of the checkpoint rows must fit 32 bits
        if (!(P.ok && out->flux && P.sc >= 0 && P.packed_ok)) return false;
    }
    // the second generation also takes a trajectory without the saved powers (inference that keeps the state series)
    bool ok = P.ok && out->flux && (P.sc >= 0 ? (!SAVE_POW || out->traj || !out->aux) : aux_matches_traj(out));
    if (out->traj && !packed && !ckpt) ok = ok && P.rows_ok;
    if (packed) {
        if (!(ok && P.sc >= 0 && P.packed_ok)) {
            *rc = fail(HBVX_E_UNSUPPORTED, "packed trajectory asked for a problem hbvx_preferred_traj_layout does not pack");
            return true;
        }
    } else if (!(ok && (any_size || P.wgs >= stream_min(P, out->traj != nullptr, false)))) {
        return false;
    }
    StreamArgs sa;
    sa.d = *d;
    sa.o = *out;
    sa.lgMp = P.lg;
    sa.nd = P.nd;
    sa.per_xcd = 0;
    for (int k = 0; k < 6; k++) sa.dslot[k] = k < P.nd ? P.dslot[k] : 0;
    const bool tr = out->traj != nullptr, few = P.nd > 0;
    dim3 grid_s((unsigned)P.wgs);
    hipStream_t st = (hipStream_t)stream;
    if (P.sc >= 0) {
        const int trj = !tr ? 0 : (ckpt ? 3 : (packed ? 2 : 1));
        sa.per_xcd = (int)((P.wgs + 7) / 8);
        const dim3 grid2((unsigned)(8 * sa.per_xcd));
        STREAM2_DISPATCH(go_fwd2, d, P.sc, trj, sa, grid2, st);
    } else {
#define STREAM_GO(MODEL, BE)                                                                          \
    do {                                                                                              \
        if (tr) { if (few) hipLaunchKernelGGL((k_fwd_stream<MODEL, BE, true, true>), grid_s, dim3(64), 0, st, sa);   \
                  else hipLaunchKernelGGL((k_fwd_stream<MODEL, BE, true, false>), grid_s, dim3(64), 0, st, sa); }   \
        else { if (few) hipLaunchKernelGGL((k_fwd_stream<MODEL, BE, false, true>), grid_s, dim3(64), 0, st, sa);    \
               else hipLaunchKernelGGL((k_fwd_stream<MODEL, BE, false, false>), grid_s, dim3(64), 0, st, sa); }     \
    } while (0)
        if (d->model == HBVX_MODEL_HBV10 && d->n_param == 12) STREAM_GO(MODEL_HBV10, false);
        else if (d->model == HBVX_MODEL_HBV10) STREAM_GO(MODEL_HBV10, true);
        else if (d->model == HBVX_MODEL_HBV11P) STREAM_GO(MODEL_HBV11P, true);
        else if (d->model == HBVX_MODEL_HOURLY) STREAM_GO(MODEL_HOURLY, true);
        else STREAM_GO(MODEL_HBV20, true);
#undef STREAM_GO
    }
    hipError_t e = hipGetLastError();
    note_dispatch(0, P.sc >= 0 ? "stream2" : "stream");
    *rc = e != hipSuccess ? hip_fail(e, "hbvx_forward (stream) launch") : HBVX_OK;
    return true;
}

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
