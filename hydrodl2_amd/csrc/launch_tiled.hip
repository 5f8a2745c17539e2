// launch_tiled.hip -- host dispatch of the wave-specialised LDS-tiled kernels (hbv_tiled.h) and of the
// implicit scheme (hbv_adj_kernels.h), whose forward is the tiled stepper running the Newton solve.
#include "hbvx_host.h"
#include "hbv_tiled.h"
#define HBVX_CHUNK_NO_SHARED_KERNELS
#include "hbv_chunked.h"        // (first: hbv_adj_kernels.h uses its XCD-aware block map)
#include "hbv_adj_kernels.h"

using namespace hbvx;
using namespace hbvx_host;

namespace hbvx_host {
// launch_chunked.hip: the scan / reduce kernels of the time-parallel scheme (shared with the implicit one)
void launch_chunk_scan(const hbvx::ChunkArgs &a, hipStream_t st);
void launch_chunk_reduce(const hbvx::ChunkArgs &a, int n_param, hipStream_t st);
}

static bool geom_fwd(const hbvx_desc *d, const hbvx_fwd_out *o, TileGeom &g)
{
    g = TileGeom{};
    g.lgMp = lg_members(d->M);
    g.ND = count_dyn(d);
    g.NDm = g.ND + (d->muwts ? 1 : 0);
    const int NF = (d->model == HBVX_MODEL_HBV10) ? 11 : (d->model == HBVX_MODEL_HBVADJ ? 1 : 12);
    // Few workgroups (cfg2: 168 on 256 CUs): deep tiles, one workgroup owns its CU.  Many
    // workgroups (cfg5: 3125): shallow tiles so that several steppers share a CU (LDS decides how
    // many) -- measured at cfg5: Kt 16/8 -> 7.4 ms, Kt 4 -> 5.8 ms.
    const int wgs = (d->B + (64 >> g.lgMp) - 1) / (64 >> g.lgMp);
    const int ktmax = env_int("HBVX_KT", wgs >= 1024 ? 4 : 16);
    for (int Kt = 16; Kt >= 1; Kt >>= 1) {
        if (Kt > ktmax) continue;
        g.Kt = Kt;
        g.off_pin = Kt * 256;
        g.in_sz = Kt * (256 + g.NDm * 64);
        g.off_tout = Kt * 64 * (o->flux ? NF : 0);
        g.out_sz = g.off_tout + Kt * 64 * (o->traj ? 7 : 0);
        if (g.out_sz == 0) g.out_sz = 4;
        if (2 * (g.in_sz + g.out_sz) * 4 <= LDS_BUDGET) return true;
    }
    return false;
}

static bool geom_bwd(const hbvx_desc *d, const hbvx_bwd_io *io, TileGeom &g)
{
    g = TileGeom{};
    g.lgMp = lg_members(d->M);
    g.ND = count_dyn(d);
    g.NDm = g.ND + (d->muwts ? 1 : 0);
    const int NF = io->grad_flux ? ((d->model == HBVX_MODEL_HBV10) ? 11 : 12) : 4; // staged series
    const int bpw = 64 >> g.lgMp;
    const int ktmax = env_int("HBVX_KT", 16);
    for (int Kt = 16; Kt >= 1; Kt >>= 1) {
        if (Kt > ktmax) continue;
        g.Kt = Kt;
        g.off_pin = Kt * 256;
        g.off_tin = g.off_pin + Kt * g.NDm * 64;
        g.off_gin = g.off_tin + Kt * 7 * 64;
        g.in_sz = (g.off_gin + Kt * NF * bpw + 3) & ~3;
        g.off_xout = Kt * 64 * g.ND;
        g.off_mout = g.off_xout + Kt * 64 * (io->grad_x ? 3 : 0);
        g.out_sz = g.off_mout + Kt * 64 * (io->grad_muwts ? 1 : 0);
        if (g.out_sz == 0) g.out_sz = 4;
        if (2 * (g.in_sz + g.out_sz) * 4 <= LDS_BUDGET) return true;
    }
    return false;
}

template <typename Args, typename K>
static hipError_t launch_tiled_one(K kern, const Args &a, dim3 grid, int threads, size_t lds,
                                   hipStream_t st)
{
    hipError_t e = set_dynamic_lds(reinterpret_cast<const void *>(kern), (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds, st, a);
    return hipGetLastError();
}

// LIMIT256: the kernel's HBV 2.0 / hourly instances are compiled for 256 threads (hbv_tiled.h, bwd_tiled_threads)
#define LAUNCH_TILED_V(K, d, a, grid, lds, st, ...)                                                \
    ([&]() -> hipError_t {                                                                        \
        int nh = env_int("HBVX_NH", (grid).x >= 1024 ? 3 : 7);                                    \
        nh = nh < 1 ? 1 : (nh > 7 ? 7 : nh);                                                      \
        if (LIMIT256 && ((d)->model == HBVX_MODEL_HBV20 || (d)->model == HBVX_MODEL_HOURLY)) nh = nh > 3 ? 3 : nh; \
        const int threads = 64 * (1 + nh);                                                        \
        const int m = (d)->model;                                                                 \
        const bool be = (d)->n_param == 13;                                                       \
        if (m == HBVX_MODEL_HBV10 && !be) return launch_tiled_one(K<MODEL_HBV10, false, __VA_ARGS__>, a, grid, threads, lds, st); \
        if (m == HBVX_MODEL_HBV10) return launch_tiled_one(K<MODEL_HBV10, true, __VA_ARGS__>, a, grid, threads, lds, st);         \
        if (m == HBVX_MODEL_HBV11P) return launch_tiled_one(K<MODEL_HBV11P, true, __VA_ARGS__>, a, grid, threads, lds, st);       \
        if (m == HBVX_MODEL_HOURLY) return launch_tiled_one(K<MODEL_HOURLY, true, __VA_ARGS__>, a, grid, threads, lds, st);       \
        return launch_tiled_one(K<MODEL_HBV20, true, __VA_ARGS__>, a, grid, threads, lds, st);    \
    })()

bool hbvx_host::try_fwd_tiled(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream, int *rc)
{
        FwdTArgs ta;
        if (use_tiled(d) && geom_fwd(d, out, ta.g)) {
            ta.d = *d;
            ta.o = *out;
            const int bpw_t = 64 >> ta.g.lgMp;
            dim3 grid_t((d->B + bpw_t - 1) / bpw_t);
            const size_t lds = (size_t)2 * (ta.g.in_sz + ta.g.out_sz) * 4;
            const bool dyn = ta.g.NDm > 0;
            constexpr bool LIMIT256 = false;
            hipError_t e = dyn ? LAUNCH_TILED_V(k_fwd_tiled, d, ta, grid_t, lds, (hipStream_t)stream, true)
                               : LAUNCH_TILED_V(k_fwd_tiled, d, ta, grid_t, lds, (hipStream_t)stream, false);
            note_dispatch(0, "tiled");
            *rc = e != hipSuccess ? hip_fail(e, "hbvx_forward (tiled) launch") : HBVX_OK;
            return true;
        }
    return false;
}

bool hbvx_host::try_bwd_tiled(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream, int *rc)
{
        BwdTArgs ta;
        if (use_tiled(d) && geom_bwd(d, io, ta.g)) {
            ta.d = *d;
            ta.io = *io;
            const int bpw_t = 64 >> ta.g.lgMp;
            dim3 grid_t((d->B + bpw_t - 1) / bpw_t);
            const size_t lds = (size_t)2 * (ta.g.in_sz + ta.g.out_sz) * 4;
            const bool dyn = ta.g.NDm > 0, gfull = io->grad_flux != nullptr;
            hipStream_t st_ = (hipStream_t)stream;
            store_gate(io, st_);
            hipError_t e =
                [&]() -> hipError_t {
                    if (dyn) {
                        constexpr bool LIMIT256 = true;
                        return gfull ? LAUNCH_TILED_V(k_bwd_tiled, d, ta, grid_t, lds, st_, true, true)
                                     : LAUNCH_TILED_V(k_bwd_tiled, d, ta, grid_t, lds, st_, true, false);
                    }
                    constexpr bool LIMIT256 = false;
                    return gfull ? LAUNCH_TILED_V(k_bwd_tiled, d, ta, grid_t, lds, st_, false, true)
                                 : LAUNCH_TILED_V(k_bwd_tiled, d, ta, grid_t, lds, st_, false, false);
                }();
            note_dispatch(1, "tiled");
            *rc = e != hipSuccess ? hip_fail(e, "hbvx_backward (tiled) launch") : HBVX_OK;
            return true;
        }
    return false;
}

// ---------------------------------------------------------------------------
// implicit HBV ("HBV adjoint", hbv_adj.py)
// ---------------------------------------------------------------------------
static int check_adj(const hbvx_desc *d)
{
    int rc = check_desc(d);
    if (rc) return rc;
    if (d->model != HBVX_MODEL_HBVADJ) return fail(HBVX_E_UNSUPPORTED, "hbvx_adj_* needs model HBVADJ");
    if (!(d->adj_gtol >= 0.0f) || d->adj_max_iter < 0 || d->adj_max_iter > 64)
        return fail(HBVX_E_SHAPE, "bad Newton policy (adj_gtol / adj_max_iter)");
    return HBVX_OK;
}

static int adj_forward_dispatch(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream);

extern "C" int hbvx_adj_forward(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream)
{
    zero_taken() = false;
    if (out && out->zero_ptr && !out->zero_state) return fail(HBVX_E_NULL, "zero_ptr needs zero_state");
    return adj_forward_dispatch(d, out, stream);
}

static int adj_forward_dispatch(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream)
{
    int rc = check_adj(d);
    if (rc) return rc;
    if (!out || !out->state_out) return fail(HBVX_E_NULL, "state_out is NULL");
    if (out->flux && out->n_flux != 1) return fail(HBVX_E_SHAPE, "hbvx_adj_forward writes n_flux = 1");
    if (d->adj_stop < 0 || d->adj_stop > 2) return fail(HBVX_E_SHAPE, "adj_stop must be 0, 1 or 2");
    if (d->adj_stop == 2 && try_fwd_pipe(d, out, stream, &rc)) return rc;   // staged solve: three-wave pipeline
    {
        FwdTArgs ta;
        if (use_tiled(d) && geom_fwd(d, out, ta.g)) {
            ta.d = *d;
            ta.o = *out;
            const int bpw_t = 64 >> ta.g.lgMp;
            dim3 grid_t((d->B + bpw_t - 1) / bpw_t);
            const size_t lds = (size_t)2 * (ta.g.in_sz + ta.g.out_sz) * 4;
            const bool dyn = ta.g.NDm > 0, be = d->n_param == 13;
            int nh = env_int("HBVX_NH", 7);
            nh = nh < 1 ? 1 : (nh > 7 ? 7 : nh);
            const int threads = 64 * (1 + nh);
            hipStream_t st = (hipStream_t)stream;
            hipError_t e =
                be ? (dyn ? launch_tiled_one(k_fwd_tiled<MODEL_HBVADJ, true, true>, ta, grid_t, threads, lds, st)
                          : launch_tiled_one(k_fwd_tiled<MODEL_HBVADJ, true, false>, ta, grid_t, threads, lds, st))
                   : (dyn ? launch_tiled_one(k_fwd_tiled<MODEL_HBVADJ, false, true>, ta, grid_t, threads, lds, st)
                          : launch_tiled_one(k_fwd_tiled<MODEL_HBVADJ, false, false>, ta, grid_t, threads, lds, st));
            if (e != hipSuccess) return hip_fail(e, "hbvx_adj_forward (tiled) launch");
            return HBVX_OK;
        }
    }
    AdjFwdArgs a;
    a.d = *d;
    a.o = *out;
    a.lgMp = lg_members(d->M);
    const int bpw = 64 >> a.lgMp;
    dim3 grid((d->B + bpw - 1) / bpw);
    if (d->n_param == 13) hipLaunchKernelGGL(k_adj_fwd<true>, grid, dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_adj_fwd<false>, grid, dim3(64), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "hbvx_adj_forward launch");
    return HBVX_OK;
}

extern "C" int hbvx_adj_backward(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream)
{
    int rc = check_adj(d);
    if (rc) return rc;
    if (!io || !io->traj) return fail(HBVX_E_NULL, "traj is NULL");
    if (io->n_flux != 1) return fail(HBVX_E_SHAPE, "hbvx_adj_backward expects n_flux = 1");
    if (d->T == 0) return HBVX_OK;
    if (io->workspace && chunked_applicable(d) && io->workspace_bytes >= hbvx_backward_workspace_bytes(d)) {
        // time-parallel adjoint (hbv_adj_kernels.h + the scan / reduce kernels of hbv_chunked.h)
        ChunkArgs ca;
        ca.d = *d;
        ca.io = *io;
        ca.lgMp = lg_members(d->M);
        ca.C = chunk_days();
        ca.nchunk = (d->T + ca.C - 1) / ca.C;
        const int64_t N = (int64_t)d->B * d->M;
        ca.phi = (float *)io->workspace;
        ca.abnd = ca.phi + (int64_t)ca.nchunk * 30 * N;
        ca.gpart = ca.abnd + (int64_t)ca.nchunk * 5 * N;
        hipStream_t st = (hipStream_t)stream;
        ca.per_xcd = chunk_per_xcd(d->B, ca.lgMp);
        dim3 g2((unsigned)(8 * ca.per_xcd * ca.nchunk));      // XCD-aware 1-D block map (hbv_chunked.h::chunk_block)
        // at most ADJ_FEW dynamic parameters: the slot-list instances (hbv_adj_kernels.h, "few" mode)
        AdjFew few{};
        const bool is_few = count_dyn(d) <= ADJ_FEW;
        if (is_few)
            for (int i = 0; i < d->n_param; i++)
                if (d->p[i].dyn) few.slot[few.nd++] = i;
        if (d->n_param == 13) {
            if (is_few) hipLaunchKernelGGL((k_adj_chunk_phi<true, true>), g2, dim3(64), 0, st, *d, *io, ca.lgMp, ca.C, ca.phi, ca.per_xcd, few);
            else hipLaunchKernelGGL((k_adj_chunk_phi<true, false>), g2, dim3(64), 0, st, *d, *io, ca.lgMp, ca.C, ca.phi, ca.per_xcd, few);
        } else {
            if (is_few) hipLaunchKernelGGL((k_adj_chunk_phi<false, true>), g2, dim3(64), 0, st, *d, *io, ca.lgMp, ca.C, ca.phi, ca.per_xcd, few);
            else hipLaunchKernelGGL((k_adj_chunk_phi<false, false>), g2, dim3(64), 0, st, *d, *io, ca.lgMp, ca.C, ca.phi, ca.per_xcd, few);
        }
        launch_chunk_scan(ca, st);
        store_gate(io, st);      // phi and scan only read; the sweep stores the dynamic gradients
        if (d->n_param == 13) {
            if (is_few) hipLaunchKernelGGL((k_adj_chunk_sweep<true, true>), g2, dim3(64), 0, st, *d, *io, ca.lgMp, ca.C, ca.abnd, ca.gpart, ca.per_xcd, few);
            else hipLaunchKernelGGL((k_adj_chunk_sweep<true, false>), g2, dim3(64), 0, st, *d, *io, ca.lgMp, ca.C, ca.abnd, ca.gpart, ca.per_xcd, few);
        } else {
            if (is_few) hipLaunchKernelGGL((k_adj_chunk_sweep<false, true>), g2, dim3(64), 0, st, *d, *io, ca.lgMp, ca.C, ca.abnd, ca.gpart, ca.per_xcd, few);
            else hipLaunchKernelGGL((k_adj_chunk_sweep<false, false>), g2, dim3(64), 0, st, *d, *io, ca.lgMp, ca.C, ca.abnd, ca.gpart, ca.per_xcd, few);
        }
        launch_chunk_reduce(ca, d->n_param, st);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "hbvx_adj_backward (chunked) launch");
        return HBVX_OK;
    }
    AdjBwdArgs a;
    a.d = *d;
    a.io = *io;
    a.lgMp = lg_members(d->M);
    const int bpw = 64 >> a.lgMp;
    dim3 grid((d->B + bpw - 1) / bpw);
    store_gate(io, (hipStream_t)stream);
    if (d->n_param == 13) hipLaunchKernelGGL(k_adj_bwd<true>, grid, dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_adj_bwd<false>, grid, dim3(64), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "hbvx_adj_backward launch");
    return HBVX_OK;
}
