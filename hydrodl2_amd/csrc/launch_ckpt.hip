// launch_ckpt.hip -- adjoint from K-day checkpoints (hbv_ckpt.h).
#include "hbvx_host.h"
#include "hbv_ckpt.h"

using namespace hbvx;
using namespace hbvx_host;

bool hbvx_host::try_bwd_ckpt(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream, int *rc)
{
    const int K = HBVX_TRAJ_CKPT_DAYS(io->traj_layout);
    if (K != 4 && K != 8 && K != 16) {
        *rc = fail(HBVX_E_SHAPE, "checkpoint interval must be 4, 8 or 16");
        return true;
    }
    if (!io->grad_flux && !io->grad_flux4 && !io->grad_state_out) {
        *rc = fail(HBVX_E_NULL, "checkpoints: no incoming gradient");
        return true;
    }
    CkptBwdArgs a;
    a.d = *d;
    a.io = *io;
    a.lgMp = lg_members(d->M);
    a.K = K;
    const int bpw = 64 >> a.lgMp;
    const dim3 grid((d->B + bpw - 1) / bpw);
    const size_t lds = (size_t)K * 7 * 64 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const int m = d->model;
    if (m == HBVX_MODEL_HBV10 && d->n_param == 12) hipLaunchKernelGGL((k_bwd_ckpt<MODEL_HBV10, false>), grid, dim3(64), lds, st, a);
    else if (m == HBVX_MODEL_HBV10) hipLaunchKernelGGL((k_bwd_ckpt<MODEL_HBV10, true>), grid, dim3(64), lds, st, a);
    else if (m == HBVX_MODEL_HBV11P) hipLaunchKernelGGL((k_bwd_ckpt<MODEL_HBV11P, true>), grid, dim3(64), lds, st, a);
    else if (m == HBVX_MODEL_HOURLY) hipLaunchKernelGGL((k_bwd_ckpt<MODEL_HOURLY, true>), grid, dim3(64), lds, st, a);
    else hipLaunchKernelGGL((k_bwd_ckpt<MODEL_HBV20, true>), grid, dim3(64), lds, st, a);
    hipError_t e = hipGetLastError();
    *rc = e != hipSuccess ? hip_fail(e, "hbvx_backward (checkpoints) launch") : HBVX_OK;
    return true;
}
