// launch_ckpt.hip -- adjoint from K-day checkpoints (hbv_ckpt.h).
//
// With caller scratch (hbvx_ckpt_workspace_bytes) the adjoint runs BLOCK-WISE: the record is cut into
// blocks of `Tb` days; for each block, last first, (1) k_ckpt_remat re-materialises the block's
// trajectory from its checkpoints -- every K-day segment in parallel, one forward step per day --
// into scratch rows, (2) the incoming gradient series of the block are copied next to it, and (3)
// the ordinary adjoint (hbvx_backward on a Tb-day window: streaming or time-parallel kernels) runs on
// it, seeded with the adjoint of the block after it.  Memory is O(Tb), the kernels are the fast ones.
// Without scratch: k_bwd_ckpt, one serial wave per 64 lanes with its segment in LDS.
#include "hbvx_host.h"
#include "hbv_ckpt.h"

using namespace hbvx;
using namespace hbvx_host;

namespace {

// Days per block.  The block's scratch is 5 (Tb + 1) N floats of re-materialised trajectory (20 bytes per lane-day;
// 28 in HBVX_SAVE_POW builds) plus the inner adjoint's own workspace: 16 GB for 512 days of 1.6 M lanes -- not
// "lean".  So the block is sized by BYTES: at most HBVX_CKPT_SCRATCH_MB (default 2048) of trajectory scratch, at
// most HBVX_CKPT_BLOCK (default 512) days, a multiple of K.  Short blocks cost launches (4 kernels per block), so
// the block is at least 8 K days while the byte budget allows it; when it does not (1.6 M lanes at K = 16: 67
// days) the budget wins down to a floor of 2 K days -- the bound include/hbvx.h states is kept.
int block_days(const hbvx_desc *d, int K)
{
    int tb = env_int("HBVX_CKPT_BLOCK", 512);
    const uint64_t N = (uint64_t)d->B * d->M;
    const uint64_t budget = (uint64_t)env_int("HBVX_CKPT_SCRATCH_MB", 2048) << 20;
    const uint64_t by_bytes = budget / ((SAVE_POW ? 28 : 20) * (N ? N : 1));
    if (tb < 8 * K) tb = 8 * K;
    if ((uint64_t)tb > by_bytes) tb = (int)by_bytes;
    if (tb < 2 * K) tb = 2 * K;
    tb = (tb / K) * K;
    const int tfull = ((d->T + K - 1) / K) * K;
    return tb > tfull ? tfull : tb;
}

// out[s][t][b] = in[s][t0 + t][b] for t < tb: the block's window of S gradient series [S,T,B] -> [S,tb,B].
// (A strided device copy: hipMemcpy2DAsync would do, but its pitch here is T*B*4 bytes -- 2.9 GB for 100 000
// basins x 7 300 days, beyond 2^31 -- and its width 200 MB: outside what 2-D copies are exercised at.)
__global__ void __launch_bounds__(256) k_ckpt_window(const float *__restrict__ in, float *__restrict__ out, int64_t row,
                                                     int64_t T_B, int64_t t0_B)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= row) return;
    const int s = blockIdx.y;
    out[(int64_t)s * row + i] = in[(int64_t)s * T_B + t0_B + i];
}

hipError_t copy_window(const float *in, float *out, int S, int tb, int T, int B, int t0, hipStream_t st)
{
    const int64_t row = (int64_t)tb * B;
    hipLaunchKernelGGL(k_ckpt_window, dim3((unsigned)((row + 255) / 256), (unsigned)S), dim3(256), 0, st, in, out, row,
                       (int64_t)T * B, (int64_t)t0 * B);
    return hipGetLastError();
}

struct Plan {
    int Tb;
    uint64_t traj, aux, gf, g4, carry, inner, total;   // byte sizes; total = sum, 256-byte aligned parts
};

uint64_t al(uint64_t v) { return (v + 255) & ~(uint64_t)255; }

Plan plan(const hbvx_desc *d, int K)
{
    Plan P{};
    P.Tb = block_days(d, K);
    const uint64_t N = (uint64_t)d->B * d->M, nf = d->model == HBVX_MODEL_HBV10 ? 11 : 12;
    P.traj = al(5 * (uint64_t)(P.Tb + 1) * N * 4);
    P.aux = SAVE_POW ? al(2 * (uint64_t)P.Tb * N * 4) : 0;   // the block's saved powers (HBVX_SAVE_POW builds only)
    P.gf = al(nf * (uint64_t)P.Tb * d->B * 4);
    P.g4 = al(4 * (uint64_t)P.Tb * d->B * 4);
    P.carry = al(5 * N * 4);
    hbvx_desc sub = *d;
    sub.T = P.Tb;
    P.inner = al(hbvx_backward_workspace_bytes(&sub));
    P.total = P.traj + P.aux + P.gf + P.g4 + 2 * P.carry + P.inner;
    return P;
}

template <typename K>
void launch_remat(K kern, const RematArgs &a, dim3 grid, hipStream_t st) { hipLaunchKernelGGL(kern, grid, dim3(64), 0, st, a); }

} // namespace

// include/hbvx.h
extern "C" uint64_t hbvx_ckpt_workspace_bytes(const hbvx_desc *d, int32_t K)
{
    if (!d || d->T <= 0 || d->B <= 0 || d->M <= 0 || (K != 4 && K != 8 && K != 16)) return 0;
    if (!check_desc(d) && stream_ckpt_applicable(d, K)) return 0;     // the segment stays in LDS: no scratch
    return plan(d, K).total;
}

bool hbvx_host::try_bwd_ckpt(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream, int *rc)
{
    const int K = HBVX_TRAJ_CKPT_DAYS(io->traj_layout);
    if (K != 4 && K != 8 && K != 16) {
        *rc = fail(HBVX_E_SHAPE, "checkpoint interval must be 4, 8 or 16");
        return true;
    }
    hipStream_t st = (hipStream_t)stream;
    const int lg = lg_members(d->M);
    const int bpw = 64 >> lg;
    const unsigned W = (unsigned)((d->B + bpw - 1) / bpw);
    const int m = d->model;
    const Plan P = plan(d, K);
    if (io->workspace && io->workspace_bytes >= P.total && env_int("HBVX_CKPT_BLOCKWISE", 1) != 0) {
        char *w = (char *)io->workspace;
        float *s_traj = (float *)w; w += P.traj;
        float *s_aux = (float *)w; w += P.aux;
        float *s_gf = (float *)w; w += P.gf;
        float *s_g4 = (float *)w; w += P.g4;
        float *carry[2] = {(float *)w, (float *)(w + P.carry)};
        w += 2 * P.carry;
        void *inner = P.inner ? (void *)w : nullptr;
        const int nf = io->n_flux, T = d->T, B = d->B;
        const int nblk = (T + P.Tb - 1) / P.Tb;
        int flip = 0;
        for (int blk = nblk - 1; blk >= 0; blk--) {
            const int t0 = blk * P.Tb, tb = T - t0 < P.Tb ? T - t0 : P.Tb;
            RematArgs ra;
            ra.d = *d;
            ra.ckpt = io->traj;
            ra.traj = s_traj;
            ra.aux = s_aux;
            ra.lgMp = lg; ra.K = K; ra.t0 = t0; ra.tb = tb;
            const dim3 grid(W, (unsigned)((tb + K - 1) / K));
            if (m == HBVX_MODEL_HBV10 && d->n_param == 12) launch_remat(k_ckpt_remat<MODEL_HBV10, false>, ra, grid, st);
            else if (m == HBVX_MODEL_HBV10) launch_remat(k_ckpt_remat<MODEL_HBV10, true>, ra, grid, st);
            else if (m == HBVX_MODEL_HBV11P) launch_remat(k_ckpt_remat<MODEL_HBV11P, true>, ra, grid, st);
            else if (m == HBVX_MODEL_HOURLY) launch_remat(k_ckpt_remat<MODEL_HOURLY, true>, ra, grid, st);
            else launch_remat(k_ckpt_remat<MODEL_HBV20, true>, ra, grid, st);
            hipError_t e = hipGetLastError();
            // the block's window of the gradient series, compacted to [series, tb, B]
            if (e == hipSuccess && io->grad_flux) e = copy_window(io->grad_flux, s_gf, nf, tb, T, B, t0, st);
            if (e == hipSuccess && io->grad_flux4) e = copy_window(io->grad_flux4, s_g4, 4, tb, T, B, t0, st);
            if (e != hipSuccess) {
                *rc = hip_fail(e, "hbvx_backward (checkpoints) block setup");
                return true;
            }
            hbvx_desc sd = *d;
            sd.T = tb;
            sd.x = d->x + (int64_t)t0 * d->x_t_stride;
            if (d->muwts) sd.muwts = d->muwts + (int64_t)t0 * d->mu_t_stride;
            sd.state_in = nullptr;
            hbvx_bwd_io si = *io;
            si.traj = s_traj;
            si.aux = s_aux;
            si.traj_layout = HBVX_TRAJ_ROWS;
            si.grad_flux = io->grad_flux ? s_gf : nullptr;
            si.grad_flux4 = io->grad_flux4 ? s_g4 : nullptr;
            si.grad_state_out = blk == nblk - 1 ? io->grad_state_out : carry[flip];
            si.grad_state_in = blk == 0 ? io->grad_state_in : carry[flip ^ 1];
            if (io->grad_x) si.grad_x = io->grad_x + (int64_t)t0 * d->x_t_stride;
            if (io->grad_muwts) si.grad_muwts = io->grad_muwts + (int64_t)t0 * B * d->M;
            for (int i = 0; i < d->n_param; i++) {
                if (d->p[i].dyn) sd.p[i].dyn = d->p[i].dyn + (int64_t)t0 * d->p[i].dyn_t_stride;
                if (io->g[i].dyn) si.g[i].dyn = io->g[i].dyn + (int64_t)t0 * io->g[i].dyn_t_stride;
            }
            si.workspace = inner;
            si.workspace_bytes = P.inner;
            const int r = hbvx_backward(&sd, &si, stream);
            if (r) {
                *rc = r;
                return true;
            }
            flip ^= 1;
        }
        {   // the blocks' own adjoint family (the last inner call left it) behind the block driver's name
            char both[48];
            snprintf(both, sizeof both, "ckpt-block:%s", hbvx_last_dispatch(1));
            note_dispatch(1, both);
        }
        *rc = HBVX_OK;
        return true;
    }
    // no scratch: serial wave per 64 lanes, segment in LDS
    if (!io->grad_flux && !io->grad_flux4 && !io->grad_state_out) {
        *rc = fail(HBVX_E_NULL, "checkpoints: no incoming gradient");
        return true;
    }
    CkptBwdArgs a;
    a.d = *d;
    a.io = *io;
    a.lgMp = lg;
    a.K = K;
    const dim3 grid(W);
    const size_t lds = (size_t)K * 7 * 64 * sizeof(float);
    store_gate(io, st);   // the one kernel of this path stores the dynamic-parameter / forcing gradients
    if (m == HBVX_MODEL_HBV10 && d->n_param == 12) hipLaunchKernelGGL((k_bwd_ckpt<MODEL_HBV10, false>), grid, dim3(64), lds, st, a);
    else if (m == HBVX_MODEL_HBV10) hipLaunchKernelGGL((k_bwd_ckpt<MODEL_HBV10, true>), grid, dim3(64), lds, st, a);
    else if (m == HBVX_MODEL_HBV11P) hipLaunchKernelGGL((k_bwd_ckpt<MODEL_HBV11P, true>), grid, dim3(64), lds, st, a);
    else if (m == HBVX_MODEL_HOURLY) hipLaunchKernelGGL((k_bwd_ckpt<MODEL_HOURLY, true>), grid, dim3(64), lds, st, a);
    else hipLaunchKernelGGL((k_bwd_ckpt<MODEL_HBV20, true>), grid, dim3(64), lds, st, a);
    hipError_t e = hipGetLastError();
    note_dispatch(1, "ckpt-lds");
    *rc = e != hipSuccess ? hip_fail(e, "hbvx_backward (checkpoints) launch") : HBVX_OK;
    return true;
}
