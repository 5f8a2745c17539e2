// launch_pipe.hip -- host dispatch of the pipelined forward (hbv_pipe.h).
#include "hbvx_host.h"
#include "hbv_pipe.h"

using namespace hbvx;
using namespace hbvx_host;

template <typename Args, typename K>
static hipError_t launch_tiled_one(K kern, const Args &a, dim3 grid, int threads, size_t lds,
                                   hipStream_t st)
{
    hipError_t e = set_dynamic_lds(reinterpret_cast<const void *>(kern), (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(threads), lds, st, a);
    return hipGetLastError();
}

bool hbvx_host::try_fwd_pipe(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream, int *rc)
{
        // HBV 1.0 / 1.1p / 2.0, at most PIPE_MAXDYN dynamic parameters, flux requested: pipelined
        // forward (hbv_pipe.h; three stages for HBV 1.0, two for the capillary models)
        const char *fv = getenv("HBVX_FWD");
        const bool adj = d->model == HBVX_MODEL_HBVADJ;   // implicit scheme, staged solve (hbvx_adj_forward only)
        // staged rows per day: the dynamic parameters' and, behind them, the learned ensemble weights `muwts`
        // (hbv_pipe.h: has_mu); "nd" below counts both
        const bool mu = d->muwts != nullptr && !adj;
        const int nd = count_dyn(d) + (mu ? 1 : 0);
        const bool many = nd > PIPE_FEWDYN;          // 4-day tiles, several staged rows per filler wave
        const int Kt = many ? PIPE_KT_MANY : (nd > 0 ? PIPE_KT : PIPE_KT_STATIC);
        const bool cap = d->model != HBVX_MODEL_HBV10 && !adj;
        const int nfl = adj ? 1 : (cap ? 12 : 11);
        // per-lane and per-tile byte offsets are 32-bit in the pipelined kernel
        bool off32 = ((int64_t)d->B * d->x_b_stride + (int64_t)Kt * d->x_t_stride) * 4 < (int64_t)1 << 31;
        for (int i = 0; i < d->n_param; i++)
            if (d->p[i].dyn)
                off32 = off32 && ((int64_t)d->B * d->p[i].dyn_b_stride + (int64_t)Kt * d->p[i].dyn_t_stride) * 4 <
                                     (int64_t)1 << 31;
        if (mu) off32 = off32 && ((int64_t)d->B * d->mu_b_stride + (int64_t)Kt * d->mu_t_stride) * 4 < (int64_t)1 << 31 &&
                        d->mu_t_stride >= 0;
        const int64_t wgs_p = ((int64_t)d->B + (64 >> lg_members(d->M)) - 1) / (64 >> lg_members(d->M));
        const bool pmodel = d->model == HBVX_MODEL_HBV10 || d->model == HBVX_MODEL_HBV11P ||
                            d->model == HBVX_MODEL_HBV20 || d->model == HBVX_MODEL_HOURLY ||
                            (adj && d->adj_stop == 2 && !many);
        const size_t lds = (size_t)PipeLds(Kt, nd > 0 ? (many ? nd : PIPE_FEWDYN) : 0, cap).total * 4;
        // HBVX_TRAJ_CKPT: the row drainers keep every K-th day only (no powers)
        const int ckpt_k = (out->traj && HBVX_TRAJ_KIND(out->traj_layout) == HBVX_TRAJ_CKPT)
                               ? HBVX_TRAJ_CKPT_DAYS(out->traj_layout) : 0;
        const bool ckpt_fits = !ckpt_k || ((int64_t)((d->T + ckpt_k - 1) / ckpt_k) * 5 * d->B * d->M * 4 < (int64_t)1 << 32);
        if (ckpt_fits && use_tiled(d) && !(fv && !strcmp(fv, "tiled")) && pmodel && off32 &&
            nd <= PIPE_MAXDYN && (int)lds <= LDS_BUDGET && wgs_p < 4096 && !(adj && d->muwts) && (out->flux || adj) && d->T >= 4 * Kt &&
            (adj ? !ckpt_k : (ckpt_k ? true : aux_matches_traj(out))) &&
            (int64_t)d->B * d->M * 4 * Kt < (int64_t)1 << 31 &&
            (int64_t)nfl * d->T * d->B * 4 < (int64_t)1 << 31) {
            PipeArgs pa;
            pa.d = *d;
            pa.o = *out;
            pa.lgMp = lg_members(d->M);
            pa.Kt = Kt;
            pa.ckptK = ckpt_k;
            const int bpw_p = 64 >> pa.lgMp;
            dim3 grid_p((d->B + bpw_p - 1) / bpw_p);
            // hbvx_fwd_out.zero_ptr: surplus workgroups of THIS launch write its zeros while the recurrence runs -- when the
            // chain leaves at least an eighth of the chip idle (config 2: 168 workgroups, one per CU, 88 CUs free).
            // HBVX_PIPE_FILL_WGS: 0 = never (everything is left to hbvx_zero_rest), n = that many fill workgroups;
            // HBVX_PIPE_FILL_WAVES: waves of each that write (tools: the fill's share of HBM)
            pa.fill_ptr = nullptr;
            pa.fill_state = nullptr;
            pa.fill_n16 = 0;
            pa.nwg = (int)grid_p.x;
            pa.fill_waves = env_int("HBVX_PIPE_FILL_WAVES", 16);
            if (out->zero_ptr && out->zero_state && out->zero_bytes >= 16 && ((uintptr_t)out->zero_ptr & 15) == 0) {
                const int n_cu = device_cu_count();
                const int spare = n_cu - (int)grid_p.x;
                const int nfill = env_int("HBVX_PIPE_FILL_WGS", spare >= n_cu / 8 ? spare : 0);
                if (nfill > 0) {
                    pa.fill_ptr = out->zero_ptr;
                    pa.fill_state = (unsigned *)out->zero_state;
                    pa.fill_n16 = out->zero_bytes / 16;     // a ragged tail (< 16 bytes) belongs to hbvx_zero_rest
                    grid_p.x += (unsigned)nfill;
                    zero_taken() = true;
                }
            }
            int pthreads = env_int("HBVX_PIPE_THREADS", 1024); // 3 steppers + filler + drainers (hbv_pipe.h)
            pthreads = pthreads < 512 ? 512 : (pthreads > 1024 ? 1024 : (pthreads / 64) * 64);
            if (nd > 0) pthreads = 1024;   // the dynamic-parameter roles assume all 16 waves
            const bool be = d->n_param == 13, tr = out->traj != nullptr, dy = nd > 0;
            // who stores the trajectory (hbv_pipe.h, TRAJ): the steppers themselves where the reducers are the busy
            // helpers -- 8 or more basins per wave (ensembles of at most 8 members); HBVX_PIPE_DIRECT=0|1 pins it (tools)
            const int dflag = env_int("HBVX_PIPE_DIRECT", -1);
            const bool direct = !SAVE_POW && (dflag >= 0 ? dflag != 0 : bpw_p >= 8);
            hipStream_t st = (hipStream_t)stream;
            hipError_t e;
            // compile-time dynamic sets (hbv_pipe.h, SC): {BETA, BETAET} and {BETA, K0, BETAET}
            unsigned dmask = 0;
            for (int i = 0; i < d->n_param; i++) dmask |= d->p[i].dyn ? (1u << i) : 0u;
            const int sc = dmask == ((1u << P_BETA) | (1u << P_BETAET)) ? 1
                         : dmask == ((1u << P_BETA) | (1u << P_K0) | (1u << P_BETAET)) ? 2 : 0;
#define PIPE_GO(MODEL, BE, TR, DY, MANY, SC) e = launch_tiled_one(k_fwd_pipe<MODEL, BE, TR, DY, MANY, SC>, pa, grid_p, pthreads, lds, st)
#define PIPE_GO3(MODEL, BE, TR, S1, S2)                                                            \
    do {                                                                                           \
        if (many) PIPE_GO(MODEL, BE, TR, true, true, 0);                                           \
        else if (dy && sc == 1 && S1) PIPE_GO(MODEL, BE, TR, true, false, (S1 ? 1 : 0));           \
        else if (dy && sc == 2 && S2) PIPE_GO(MODEL, BE, TR, true, false, (S2 ? 2 : 0));           \
        else if (dy) PIPE_GO(MODEL, BE, TR, true, false, 0);                                       \
        else PIPE_GO(MODEL, BE, TR, false, false, 0);                                              \
    } while (0)
#define PIPE_GO4(MODEL, BE, S1, S2)                                                                \
    do {                                                                                           \
        if (tr && direct) PIPE_GO3(MODEL, BE, 2, S1, S2);                                          \
        else if (tr) PIPE_GO3(MODEL, BE, 1, S1, S2);                                               \
        else PIPE_GO3(MODEL, BE, 0, S1, S2);                                                       \
    } while (0)
            if (adj) {   // at most PIPE_FEWDYN dynamic parameters (pmodel), descriptor flags
#define PIPE_GO_ADJ(BE)                                                                            \
    do {                                                                                           \
        if (tr) { if (dy) PIPE_GO(MODEL_HBVADJ, BE, 1, true, false, 0); else PIPE_GO(MODEL_HBVADJ, BE, 1, false, false, 0); } \
        else { if (dy) PIPE_GO(MODEL_HBVADJ, BE, 0, true, false, 0); else PIPE_GO(MODEL_HBVADJ, BE, 0, false, false, 0); } \
    } while (0)
                if (be) PIPE_GO_ADJ(true);
                else PIPE_GO_ADJ(false);
#undef PIPE_GO_ADJ
            } else if (d->model == HBVX_MODEL_HBV11P) PIPE_GO4(MODEL_HBV11P, true, true, false);
            else if (d->model == HBVX_MODEL_HBV20) PIPE_GO4(MODEL_HBV20, true, false, true);
            else if (d->model == HBVX_MODEL_HOURLY) PIPE_GO4(MODEL_HOURLY, true, false, true);
            else if (be) PIPE_GO4(MODEL_HBV10, true, true, true);
            else PIPE_GO4(MODEL_HBV10, false, false, false);
#undef PIPE_GO4
#undef PIPE_GO3
#undef PIPE_GO
            note_dispatch(0, "pipe");
            *rc = e != hipSuccess ? hip_fail(e, "hbvx_forward (pipelined) launch") : HBVX_OK;
            return true;
        }
    return false;
}

#ifdef PIPE_PROBE
extern "C" int hbvx_debug_pipe_blocks(unsigned long long *out, int n)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(hbvx::g_pipe_blocks), (size_t)(n > 4096 ? 4096 : n) * sizeof(unsigned long long));
    return e == hipSuccess ? 0 : -1;
}
extern "C" int hbvx_debug_pipe_probe(unsigned long long *out32)
{
    hipError_t e = hipMemcpyFromSymbol(out32, HIP_SYMBOL(hbvx::g_pipe_probe), 32 * sizeof(unsigned long long));
    return e == hipSuccess ? 0 : -1;
}
#endif
