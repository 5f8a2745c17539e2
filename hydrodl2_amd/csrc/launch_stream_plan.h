// launch_stream_plan.h -- what the two streaming translation units share: admission of a problem to the streaming
// families (plan_stream), their grid-size cross-overs, and the (model, BETAET, dynamic set) -> template dispatch.
// launch_stream.hip holds the forwards and the layout query, launch_stream_bwd.hip the adjoints: two units because each
// instantiates ~100 kernels (a header change cost 3.3 minutes of serial compile in one unit).
#pragma once
#include "hbvx_host.h"
#include "hbv_stream2.h"

namespace hbvx_host {
namespace stream_plan {
using namespace hbvx;

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

inline StreamPlan plan_stream(const hbvx_desc *d)
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
inline int stream_min(const StreamPlan &P, bool training, bool adjoint)
{
    const int dflt = P.sc >= 0 ? (training ? 768 : 1024) : (adjoint ? 2048 : 1024);
    return env_int("HBVX_STREAM_MIN", dflt);
}
inline bool adjoint_pinned_elsewhere()
{
    const char *v = getenv("HBVX_BWD");
    return v && strcmp(v, "stream") != 0;
}

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


} // namespace stream_plan
} // namespace hbvx_host
