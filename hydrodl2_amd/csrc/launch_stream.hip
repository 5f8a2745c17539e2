// launch_stream.hip -- host dispatch of the streaming forward / adjoint for large grids
// (hbv_stream2.h; hbv_stream.h for dynamic sets or forcing layouts the second generation does not
// instantiate).
#include "launch_stream_plan.h"

using namespace hbvx;
using namespace hbvx_host;
using namespace hbvx_host::stream_plan;

namespace {

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

