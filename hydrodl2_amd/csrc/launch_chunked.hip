// launch_chunked.hip -- host dispatch of the time-parallel adjoint (hbv_chunked.h).
#include "hbvx_host.h"
#include "hbv_chunked.h"

using namespace hbvx;
using namespace hbvx_host;

namespace hbvx_host {
void launch_chunk_scan(const hbvx::ChunkArgs &a, hipStream_t st)
{
    const int64_t N = (int64_t)a.d.B * a.d.M;
    hipLaunchKernelGGL(k_bwd_chunk_scan, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, st, a);
}
void launch_chunk_reduce(const hbvx::ChunkArgs &a, int n_param, hipStream_t st)
{
    const int64_t N = (int64_t)a.d.B * a.d.M;
    hipLaunchKernelGGL(k_bwd_chunk_reduce, dim3((unsigned)((N + 255) / 256), n_param), dim3(256), 0, st, a, n_param);
}
}

int hbvx_host::chunk_days() { int c = env_int("HBVX_CHUNK", 64); return c < 2 ? 2 : c; }

bool hbvx_host::chunked_applicable(const hbvx_desc *d)
{
    const char *v = getenv("HBVX_BWD");
    if (v && !strcmp(v, "tiled")) return false;
    return d->T >= 2 * chunk_days();
}

static int np_of(const hbvx_desc *d) { return d->n_param; }

extern "C" uint64_t hbvx_backward_workspace_bytes(const hbvx_desc *d)
{
    if (!d || d->T <= 0 || d->B <= 0 || d->M <= 0 || !chunked_applicable(d)) return 0;
    const int C = chunk_days();
    const uint64_t nchunk = (uint64_t)(d->T + C - 1) / C;
    return nchunk * (uint64_t)d->B * (uint64_t)d->M * (uint64_t)(35 + np_of(d)) * sizeof(float);
}

template <int MODEL, bool BETAET, int DYN, bool GFULL>
static hipError_t launch_chunked_t(const ChunkArgs &a, hipStream_t st)
{
    const hbvx_desc &d = a.d;
    const int bpw = 64 >> a.lgMp;
    const int64_t N = (int64_t)d.B * d.M;
    (void)bpw;
    dim3 g2((unsigned)(8 * a.per_xcd * a.nchunk));      // XCD-aware 1-D block map (hbv_chunked.h::chunk_block)
    // the two slot lists users actually run get compile-time slots (hbv_chunked.h::SlotCombo)
    int sc = 0;
    const bool mu = d.muwts != nullptr;     // (the compiled slot combos have no MU instance: the slot-list form below)
    if (!mu && DYN == 1 && MODEL == MODEL_HBV10 && BETAET && a.nd == 2 && a.dslot[0] == P_BETA && a.dslot[1] == P_BETAET) sc = 1;
    if (!mu && DYN == 1 && (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) && a.nd == 3 && a.dslot[0] == P_BETA &&
        a.dslot[1] == P_K0 && a.dslot[2] == P_BETAET)
        sc = 2;
    if constexpr (DYN == 1 && MODEL == MODEL_HBV10 && BETAET) {
        if (sc == 1) {
            hipLaunchKernelGGL((k_bwd_chunk_phi<MODEL, BETAET, DYN, GFULL, 1>), g2, dim3(64), 0, st, a);
            hipLaunchKernelGGL(k_bwd_chunk_scan, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, st, a);
            store_gate(&a.io, st);
            hipLaunchKernelGGL((k_bwd_chunk_sweep<MODEL, BETAET, DYN, GFULL, 1>), g2, dim3(64), 0, st, a);
        }
    }
    if constexpr (DYN == 1 && (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY)) {
        if (sc == 2) {
            hipLaunchKernelGGL((k_bwd_chunk_phi<MODEL, BETAET, DYN, GFULL, 2>), g2, dim3(64), 0, st, a);
            hipLaunchKernelGGL(k_bwd_chunk_scan, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, st, a);
            store_gate(&a.io, st);
            hipLaunchKernelGGL((k_bwd_chunk_sweep<MODEL, BETAET, DYN, GFULL, 2>), g2, dim3(64), 0, st, a);
        }
    }
    if (sc != 0) {
        hipLaunchKernelGGL(k_bwd_chunk_reduce, dim3((unsigned)((N + 255) / 256), d.n_param), dim3(256), 0, st,
                           a, d.n_param);
        return hipGetLastError();
    }
    if constexpr (DYN == 0 || DYN == 1) {
        if (d.muwts) {    // learned ensemble weights: the MU instances of the static / slot-list modes (hbv_chunked.h)
            hipLaunchKernelGGL((k_bwd_chunk_phi<MODEL, BETAET, DYN, GFULL, 0, true>), g2, dim3(64), 0, st, a);
            hipLaunchKernelGGL(k_bwd_chunk_scan, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, st, a);
            store_gate(&a.io, st);
            hipLaunchKernelGGL((k_bwd_chunk_sweep<MODEL, BETAET, DYN, GFULL, 0, false, true>), g2, dim3(64), 0, st, a);
            hipLaunchKernelGGL(k_bwd_chunk_reduce, dim3((unsigned)((N + 255) / 256), d.n_param), dim3(256), 0, st,
                               a, d.n_param);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((k_bwd_chunk_phi<MODEL, BETAET, DYN, GFULL>), g2, dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_bwd_chunk_scan, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, st, a);
    store_gate(&a.io, st);       // phi and scan only read; the sweep stores
    if constexpr (DYN == 3) {
        // whole-row gradient stores when every parameter's gradient sits in one [T,B,ny] tensor in the
        // reference's column order, rows 8-byte aligned (hbv_chunked.h, ROWST)
        const hbvx_param_grad &g0 = a.io.g[0];
        bool rows = g0.dyn && ((uintptr_t)g0.dyn & 7) == 0 &&
                    (g0.dyn_b_stride & 1) == 0 && (g0.dyn_t_stride & 1) == 0 && ((d.n_param * d.M) & 1) == 0;
        for (int i = 1; i < d.n_param && rows; i++)
            rows = a.io.g[i].dyn == g0.dyn + (int64_t)i * d.M && a.io.g[i].dyn_b_stride == g0.dyn_b_stride &&
                   a.io.g[i].dyn_t_stride == g0.dyn_t_stride;
        if (rows) hipLaunchKernelGGL((k_bwd_chunk_sweep<MODEL, BETAET, DYN, GFULL, 0, true>), g2, dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_bwd_chunk_sweep<MODEL, BETAET, DYN, GFULL>), g2, dim3(64), 0, st, a);
        return hipGetLastError();   // every parameter dynamic: the static gradient is zero
    }
    hipLaunchKernelGGL((k_bwd_chunk_sweep<MODEL, BETAET, DYN, GFULL>), g2, dim3(64), 0, st, a);
    if (DYN == 3) return hipGetLastError();   // every parameter dynamic: the static gradient is zero
    hipLaunchKernelGGL(k_bwd_chunk_reduce, dim3((unsigned)((N + 255) / 256), d.n_param), dim3(256), 0, st,
                       a, d.n_param);
    return hipGetLastError();
}

template <int DYN, bool GFULL>
static hipError_t launch_chunked_v(const hbvx_desc *d, const ChunkArgs &a, hipStream_t st)
{
    if (d->model == HBVX_MODEL_HBV10 && d->n_param == 12) return launch_chunked_t<MODEL_HBV10, false, DYN, GFULL>(a, st);
    if (d->model == HBVX_MODEL_HBV10) return launch_chunked_t<MODEL_HBV10, true, DYN, GFULL>(a, st);
    if (d->model == HBVX_MODEL_HBV11P) return launch_chunked_t<MODEL_HBV11P, true, DYN, GFULL>(a, st);
    if (d->model == HBVX_MODEL_HOURLY) return launch_chunked_t<MODEL_HOURLY, true, DYN, GFULL>(a, st);
    return launch_chunked_t<MODEL_HBV20, true, DYN, GFULL>(a, st);
}

static hipError_t launch_chunked(const hbvx_desc *d, const hbvx_bwd_io *io, hipStream_t st)
{
    ChunkArgs a;
    a.d = *d;
    a.io = *io;
    a.lgMp = lg_members(d->M);
    a.C = chunk_days();
    a.nchunk = (d->T + a.C - 1) / a.C;
    a.per_xcd = chunk_per_xcd(d->B, a.lgMp);
    const int64_t N = (int64_t)d->B * d->M;
    a.phi = (float *)io->workspace;
    a.abnd = a.phi + (int64_t)a.nchunk * 30 * N;
    a.gpart = a.abnd + (int64_t)a.nchunk * 5 * N;
    const bool gfull = io->grad_flux != nullptr;
    a.nd = 0;
    a.dslot[0] = a.dslot[1] = a.dslot[2] = 0;
    const int ndyn = count_dyn(d);
    if (ndyn > 0 && ndyn <= CHUNK_FEW) {   // few dynamic parameters: slot-list kernels (muwts: a run-time flag in them)
        for (int i = 0; i < d->n_param; i++)
            if (d->p[i].dyn) a.dslot[a.nd++] = i;
        return gfull ? launch_chunked_v<1, true>(d, a, st) : launch_chunked_v<1, false>(d, a, st);
    }
    // "all" mode: every parameter dynamic, no dy_drop mask, no muwts, and all rows in ONE tensor with the reference's
    // column order (hbv_chunked.h::chunk_dma_dyn addresses them from parameter 0's pointer and strides)
    bool alldyn = ndyn == d->n_param && !d->muwts;
    for (int i = 0; i < d->n_param && alldyn; i++)
        alldyn = d->p[i].drop == nullptr && d->p[i].dyn == d->p[0].dyn + (int64_t)i * d->M &&
                 d->p[i].dyn_t_stride == d->p[0].dyn_t_stride && d->p[i].dyn_b_stride == d->p[0].dyn_b_stride;
    if (alldyn) return gfull ? launch_chunked_v<3, true>(d, a, st) : launch_chunked_v<3, false>(d, a, st);
    if (ndyn > 0) return gfull ? launch_chunked_v<2, true>(d, a, st) : launch_chunked_v<2, false>(d, a, st);
    return gfull ? launch_chunked_v<0, true>(d, a, st) : launch_chunked_v<0, false>(d, a, st);
}

bool hbvx_host::try_bwd_chunked(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream, int *rc)
{
    if (io->workspace && chunked_applicable(d) && io->workspace_bytes >= hbvx_backward_workspace_bytes(d)) {
        hipError_t e = launch_chunked(d, io, (hipStream_t)stream);
        note_dispatch(1, "chunked");
        *rc = e != hipSuccess ? hip_fail(e, "hbvx_backward (chunked) launch") : HBVX_OK;
        return true;
    }
    return false;
}
