// launch_stream.hip -- host dispatch of the streaming forward / adjoint for large grids (hbv_stream.h).
#include "hbvx_host.h"
#include "hbv_stream.h"

using namespace hbvx;
using namespace hbvx_host;

bool hbvx_host::try_fwd_stream(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream, int *rc)
{
        // large grids: streaming one-wave kernel (hbv_stream.h)
        const int lg = lg_members(d->M);
        const int bpw_s = 64 >> lg;
        const int64_t wgs = ((int64_t)d->B + bpw_s - 1) / bpw_s;
        const int64_t N = (int64_t)d->B * d->M, lim = (int64_t)1 << 32;
        const int nd = count_dyn(d);
        const int nf = (d->model == HBVX_MODEL_HBV10) ? 11 : 12;
        bool ok = use_tiled(d) && env_int("HBVX_STREAM", 1) != 0 && wgs >= env_int("HBVX_STREAM_MIN", 512) && nd <= 3 && !d->muwts &&
                  out->flux && (out->traj != nullptr) == (out->aux != nullptr) && d->T > 0 &&
                  5 * (int64_t)(d->T + 1) * N * 4 < lim && (int64_t)nf * d->T * d->B * 4 < lim &&
                  ((int64_t)d->T * d->x_t_stride + (int64_t)d->B * d->x_b_stride) * 4 < lim;
        for (int i = 0; i < d->n_param && ok; i++)
            if (d->p[i].dyn)
                ok = ((int64_t)d->T * d->p[i].dyn_t_stride + (int64_t)d->B * d->p[i].dyn_b_stride) * 4 < lim;
        if (ok) {
            StreamArgs sa;
            sa.d = *d;
            sa.o = *out;
            sa.lgMp = lg;
            sa.nd = 0;
            sa.dslot[0] = sa.dslot[1] = sa.dslot[2] = 0;
            for (int i = 0; i < d->n_param; i++)
                if (d->p[i].dyn) sa.dslot[sa.nd++] = i;
            const bool tr = out->traj != nullptr, few = nd > 0;
            dim3 grid_s((unsigned)wgs);
            hipStream_t st = (hipStream_t)stream;
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
            hipError_t e = hipGetLastError();
            *rc = e != hipSuccess ? hip_fail(e, "hbvx_forward (stream) launch") : HBVX_OK;
            return true;
        }
    return false;
}

bool hbvx_host::try_bwd_stream(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream, int *rc)
{
        // large grids: single-pass streaming adjoint (hbv_stream.h), no workspace.  Measured cross-over
        // against the time-parallel kernels: ~1500 wavefronts (forward stream vs tiled: ~400)
        const int lg = lg_members(d->M);
        const int bpw_s = 64 >> lg;
        const int64_t wgs = ((int64_t)d->B + bpw_s - 1) / bpw_s;
        const int64_t N = (int64_t)d->B * d->M, lim = (int64_t)1 << 32;
        const int nd = count_dyn(d);
        const int nf = (d->model == HBVX_MODEL_HBV10) ? 11 : 12;
        bool ok = use_tiled(d) && env_int("HBVX_STREAM", 1) != 0 && wgs >= env_int("HBVX_STREAM_MIN_BWD", 2048) && nd <= 3 && !d->muwts &&
                  (io->grad_flux || io->grad_flux4) && 5 * (int64_t)(d->T + 1) * N * 4 < lim &&
                  (int64_t)nf * d->T * d->B * 4 < lim &&
                  ((int64_t)d->T * d->x_t_stride + (int64_t)d->B * d->x_b_stride) * 4 < lim;
        for (int i = 0; i < d->n_param && ok; i++)
            if (d->p[i].dyn) {
                ok = ((int64_t)d->T * d->p[i].dyn_t_stride + (int64_t)d->B * d->p[i].dyn_b_stride) * 4 < lim;
                if (ok && io->g[i].dyn)
                    ok = ((int64_t)d->T * io->g[i].dyn_t_stride + (int64_t)d->B * io->g[i].dyn_b_stride) * 4 < lim;
            }
        if (ok) {
            StreamBwdArgs sa;
            sa.d = *d;
            sa.io = *io;
            sa.lgMp = lg;
            sa.nd = 0;
            sa.dslot[0] = sa.dslot[1] = sa.dslot[2] = 0;
            for (int i = 0; i < d->n_param; i++)
                if (d->p[i].dyn) sa.dslot[sa.nd++] = i;
            const bool few = nd > 0, gfull = io->grad_flux != nullptr;
            dim3 grid_s((unsigned)wgs);
            hipStream_t st = (hipStream_t)stream;
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
            hipError_t e = hipGetLastError();
            *rc = e != hipSuccess ? hip_fail(e, "hbvx_backward (stream) launch") : HBVX_OK;
            return true;
        }
    return false;
}
