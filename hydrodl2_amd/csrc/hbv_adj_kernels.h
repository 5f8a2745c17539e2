// hbv_adj_kernels.h -- implicit HBV ("HBV adjoint") forward / backward kernels.
// One wave per workgroup, lane = (basin, member) as everywhere else; per day a modified Newton
// solve in registers (hbv_adj_step.h).  First-cut launch geometry (round 1): correctness and
// parity against the oracle; the wave-specialised tiling of hbv_tiled.h is the next step here.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/hbvx.h"
#include "hbv_adj_step.h"

namespace hbvx {

struct AdjFwdArgs {
    hbvx_desc d;
    hbvx_fwd_out o;
    int lgMp;
};
struct AdjBwdArgs {
    hbvx_desc d;
    hbvx_bwd_io io;
    int lgMp;
};

struct AdjLane {
    int jm, b, j;
    bool active, leader;
    int64_t n;
};

__device__ __forceinline__ AdjLane adj_lane(const hbvx_desc &d, int lgMp, int bx = -1)
{
    AdjLane L;
    const int lane = threadIdx.x & 63;
    const int Mp = 1 << lgMp;
    L.jm = lane & (Mp - 1);
    int b = (bx < 0 ? (int)blockIdx.x : bx) * (64 >> lgMp) + (lane >> lgMp);
    L.active = (b < d.B) && (L.jm < d.M);
    L.leader = (b < d.B) && (L.jm == 0);
    L.b = b < d.B ? b : d.B - 1;
    L.j = L.jm < d.M ? L.jm : d.M - 1;
    L.n = (int64_t)L.b * d.M + L.j;
    return L;
}

__device__ __forceinline__ float adj_ens_sum(float v, int lgMp)
{
    for (int s = 0; s < lgMp; s++) v += __shfl_xor(v, 1 << s, 64);
    return v;
}

// per-lane parameter fetch for day t: unit value u[i] and physical p[i]
template <int NP>
__device__ __forceinline__ void adj_params(const hbvx_desc &d, const AdjLane &L, int t, bool raw,
                                           const float *usta, const bool *use_dyn, float *u, float *p)
{
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float uv = usta[i];
        if (s.dyn) {
            float v = s.dyn[(int64_t)t * s.dyn_t_stride + (int64_t)L.b * s.dyn_b_stride + L.j];
            v = raw ? sigmoid_dyn_(v) : v;
            uv = use_dyn[i] ? v : uv;
        }
        u[i] = uv;
        p[i] = descale_(uv, s.lo, s.hi);
    }
#pragma unroll
    for (int i = NP; i < NPARAM_MAX; i++) p[i] = 0.0f;
}

// "Few" mode of the time-parallel kernels (at most ADJ_FEW dynamic parameters, the usual case: config 4 has one): the
// static parameters are de-scaled once per lane, a day only touches the rows of the wave-uniform slot list -- like
// hbv_chunked.h's DYN == 1.  The generic form above walks all 13 slots every day behind per-slot flags, pointers and
// strides: 100+ scalar registers, 177 lane-spill operations per day in the sweep.  u[k] (k < nd): the unit value of
// dynamic parameter k that was used (for sigmoid').
#define ADJ_FEW 3
struct AdjFew {
    int nd, slot[ADJ_FEW];
};
template <int NP>
__device__ __forceinline__ void adj_params_few(const hbvx_desc &d, const AdjLane &L, int t, bool raw, const AdjFew &F,
                                               const float *psta, const bool *use_k, float *uk, float *p)
{
#pragma unroll
    for (int i = 0; i < NP; i++) p[i] = psta[i];
#pragma unroll
    for (int i = NP; i < NPARAM_MAX; i++) p[i] = 0.0f;
#pragma unroll
    for (int k = 0; k < ADJ_FEW; k++)
        if (k < F.nd) {
            const hbvx_param_src &s = d.p[F.slot[k]];
            float v = s.dyn[(int64_t)t * s.dyn_t_stride + (int64_t)L.b * s.dyn_b_stride + L.j];
            v = raw ? sigmoid_dyn_(v) : v;
            uk[k] = v;
            const float pv = descale_(v, s.lo, s.hi);
            const float cur = slot_get<NP>(p, F.slot[k]);
            slot_set<NP>(p, F.slot[k], use_k[k] ? pv : cur);
        }
}

template <bool BETAET>
__global__ void __launch_bounds__(64) k_adj_fwd(const AdjFwdArgs A)
{
    constexpr int NP = BETAET ? 13 : 12;
    const hbvx_desc &d = A.d;
    const hbvx_fwd_out &o = A.o;
    const AdjLane L = adj_lane(d, A.lgMp);
    const int T = d.T;
    const int64_t N = (int64_t)d.B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const float invM = 1.0f / (float)d.M;
    float usta[NP];
    bool use_dyn[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        usta[i] = raw ? sigmoid_(v) : v;
        use_dyn[i] = s.dyn && !(s.drop && s.drop[L.n]); // per-lane drop (hbv_adj.py:182-189)
    }
    float x[5];
#pragma unroll
    for (int k = 0; k < 5; k++) x[k] = d.state_in ? d.state_in[k * N + L.n] : 0.0f; // :254
    const float *xb = d.x + (int64_t)L.b * d.x_b_stride;
    for (int t = 0; t < T; t++) {
        float u[NP], p[NPARAM_MAX], xn[5];
        adj_params<NP>(d, L, t, raw, usta, use_dyn, u, p);
        AdjStep<BETAET> s;
        const float *xr = xb + (int64_t)t * d.x_t_stride;
        s.P = xr[d.ch_prcp]; s.Tf = xr[d.ch_tmean]; s.PET = xr[d.ch_pet];
        if (o.traj && L.active) {
#pragma unroll
            for (int k = 0; k < 5; k++) o.traj[((int64_t)k * (T + 1) + t) * N + L.n] = x[k];
        }
        float Qs = 0.0f;
        if (d.adj_stop == 2) AdjStaged<BETAET>::day(p, s.P, s.Tf, s.PET, x, d.adj_gtol, d.adj_max_iter, xn, Qs);
        else adj_newton<BETAET>(s, p, x, 1.0f, d.adj_gtol, d.adj_max_iter, xn, d.adj_stop != 0);
#pragma unroll
        for (int k = 0; k < 5; k++) x[k] = xn[k];
        if (o.flux) {
            // hbv_adj.py:309-317: Q = q0+q1+q2 at the solved state, mean over members
            const float SUZ = fmaxf(x[3], 0.0f), SLZ = fmaxf(x[4], 0.0f);
            const float q0 = p[P_K0] * fmaxf(SUZ - p[P_UZL], 0.0f);
            const float q1 = p[P_K1] * SUZ, q2 = p[P_K2] * SLZ;
            float Q = (q0 + q1) + q2;
            Q = adj_ens_sum(L.active ? Q : 0.0f, A.lgMp) * invM;
            if (L.leader) o.flux[(int64_t)t * d.B + L.b] = Q;
        }
    }
    if (L.active) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            o.state_out[k * N + L.n] = x[k];
            if (o.traj) o.traj[((int64_t)k * (T + 1) + T) * N + L.n] = x[k];
        }
    }
}

template <bool BETAET>
__global__ void __launch_bounds__(64) k_adj_bwd(const AdjBwdArgs A)
{
    constexpr int NP = BETAET ? 13 : 12;
    const hbvx_desc &d = A.d;
    const hbvx_bwd_io &io = A.io;
    const AdjLane L = adj_lane(d, A.lgMp);
    const int T = d.T;
    const int64_t N = (int64_t)d.B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const float invM = 1.0f / (float)d.M;
    float usta[NP], gsta[NP];
    bool use_dyn[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        usta[i] = raw ? sigmoid_(v) : v;
        use_dyn[i] = s.dyn && !(s.drop && s.drop[L.n]);
        gsta[i] = 0.0f;
    }
    const float *xb = d.x + (int64_t)L.b * d.x_b_stride;
    float a[5];
#pragma unroll
    for (int k = 0; k < 5; k++) a[k] = io.grad_state_out ? io.grad_state_out[k * N + L.n] : 0.0f;
    for (int t = T - 1; t >= 0; t--) {
        float u[NP], p[NPARAM_MAX], x[5], gp[NPARAM_MAX];
        adj_params<NP>(d, L, t, raw, usta, use_dyn, u, p);
        AdjStep<BETAET> s;
        const float *xr = xb + (int64_t)t * d.x_t_stride;
        s.P = xr[d.ch_prcp]; s.Tf = xr[d.ch_tmean]; s.PET = xr[d.ch_pet];
#pragma unroll
        for (int k = 0; k < 5; k++) x[k] = io.traj[((int64_t)k * (T + 1) + (t + 1)) * N + L.n];
        const int64_t gi = (int64_t)t * d.B + L.b;
        float gQ = io.grad_flux ? io.grad_flux[gi] : 0.0f;
        if (io.grad_flux4) gQ += io.grad_flux4[gi];
        gQ *= invM;
#pragma unroll
        for (int i = 0; i < NPARAM_MAX; i++) gp[i] = 0.0f;
        adj_backstep<BETAET>(s, p, x, 1.0f, gQ, a, gp);
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const float gu = gp[i] * (d.p[i].hi - d.p[i].lo);
            if (d.p[i].dyn) {
                const float gr = raw ? gu * (u[i] * (1.0f - u[i])) : gu;
                if (io.g[i].dyn && L.active)
                    io.g[i].dyn[(int64_t)t * io.g[i].dyn_t_stride + (int64_t)L.b * io.g[i].dyn_b_stride + L.j] =
                        use_dyn[i] ? gr : 0.0f;
                gsta[i] += use_dyn[i] ? 0.0f : gu;
            } else {
                gsta[i] += gu;
            }
        }
    }
    if (L.active) {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            if (!io.g[i].sta) continue;
            const float gr = raw ? gsta[i] * (usta[i] * (1.0f - usta[i])) : gsta[i];
            float *dst = io.g[i].sta + (int64_t)L.b * io.g[i].sta_b_stride + L.j;
            *dst += gr;
        }
        if (io.grad_state_in) {
#pragma unroll
            for (int k = 0; k < 5; k++) io.grad_state_in[k * N + L.n] = a[k];
        }
    }
}

} // namespace hbvx

// ---------------------------------------------------------------------------
// time-parallel adjoint of the implicit scheme (same structure as hbv_chunked.h):
//   a_t = lam/dt with (dG/dx)^T lam = a_{t+1} + gQ dQ/dx   -- linear in a.
// B1 only needs the transpose solves (5 unit vectors + the offset driven by gQ); B3 is the full
// adj_backstep seeded with the true incoming adjoint; B2 / B4 are shared with hbv_chunked.h.
// ---------------------------------------------------------------------------
namespace hbvx {

struct ChunkArgs; // hbv_chunked.h

// What a day of the time-parallel kernels reads, loaded ONE DAY AHEAD of its use (the adjoint runs t1-1 .. t0): the
// kernels are a wave per (64 lanes, chunk) walking 64 days, and with the loads at the top of the day that consumes
// them every day paid a full HBM round trip; the dynamic-parameter rows (any subset) stay in the day itself.
struct AdjRaw {
    float f[3], x[5], gQ, g4;   // gQ + g4 (the routing adjoint's share) is formed where it is used: adj_gq
};
__device__ __forceinline__ void adj_issue(const hbvx_desc &d, const hbvx_bwd_io &io, const AdjLane &L, int t, int64_t N,
                                          AdjRaw &R)
{
    const float *xr = d.x + (int64_t)t * d.x_t_stride + (int64_t)L.b * d.x_b_stride;
    R.f[0] = xr[d.ch_prcp]; R.f[1] = xr[d.ch_tmean]; R.f[2] = xr[d.ch_pet];
#pragma unroll
    for (int k = 0; k < 5; k++) R.x[k] = io.traj[((int64_t)k * (d.T + 1) + (t + 1)) * N + L.n];
    const int64_t gi = (int64_t)t * d.B + L.b;
    // loads only: an add here makes the compiler wait for all of the day's loads where they were just issued
    // (hbv_chunked.h::chunk_issue, profiles/r04_ab_chunk_prefetch.txt)
    R.gQ = io.grad_flux ? io.grad_flux[gi] : 0.0f;
    R.g4 = io.grad_flux4 ? io.grad_flux4[gi] : 0.0f;
}
__device__ __forceinline__ float adj_gq(const hbvx_bwd_io &io, const AdjRaw &R)
{
    return io.grad_flux4 ? R.gQ + R.g4 : R.gQ;
}

template <bool BETAET, bool FEW = false>
__global__ void __launch_bounds__(64) k_adj_chunk_phi(const hbvx_desc d, const hbvx_bwd_io io, int lgMp,
                                                      int C, float *phi_ws, int per_xcd, const AdjFew F)
{
    constexpr int NP = BETAET ? 13 : 12;
    ChunkBlock blk;                          // XCD-aware block map (hbv_chunked.h)
    if (!chunk_block(d, lgMp, per_xcd, blk)) return;
    const AdjLane L = adj_lane(d, lgMp, blk.bx);
    const int chunk = blk.chunk;
    const int T = d.T, t0 = chunk * C, t1 = min(T, t0 + C);
    const int64_t N = (int64_t)d.B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const float invM = 1.0f / (float)d.M;
    float usta[NP];                 // FEW: the static PHYSICAL values
    bool use_dyn[FEW ? ADJ_FEW : NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        usta[i] = raw ? sigmoid_(v) : v;
        if (FEW) usta[i] = descale_(usta[i], s.lo, s.hi);
        else use_dyn[i] = s.dyn && !(s.drop && s.drop[L.n]);
    }
    if (FEW) {
#pragma unroll
        for (int k = 0; k < ADJ_FEW; k++) {
            use_dyn[k] = false;
            if (k < F.nd) {
                const hbvx_param_src &s = d.p[F.slot[k]];
                use_dyn[k] = !(s.drop && s.drop[L.n]);
            }
        }
    }
    float Phi[5][5], phi[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        phi[k] = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; i++) Phi[k][i] = (i == k) ? 1.0f : 0.0f;
    }
    AdjRaw Rn;
    adj_issue(d, io, L, t1 - 1, N, Rn);
    for (int t = t1 - 1; t >= t0; t--) {
        const AdjRaw Rc = Rn;
        if (t > t0) adj_issue(d, io, L, t - 1, N, Rn);     // next day's loads in flight
        float u[NP], p[NPARAM_MAX], x[5];
        if (FEW) adj_params_few<NP>(d, L, t, raw, F, usta, use_dyn, u, p);
        else adj_params<NP>(d, L, t, raw, usta, use_dyn, u, p);
        AdjStep<BETAET> s;
        s.P = Rc.f[0]; s.Tf = Rc.f[1]; s.PET = Rc.f[2];
#pragma unroll
        for (int k = 0; k < 5; k++) x[k] = Rc.x[k];
        const float gQ = adj_gq(io, Rc) * invM;
        s.template eval<true>(x, p);
        float lam[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            s.solve_t(1.0f, Phi[k], lam);
#pragma unroll
            for (int i = 0; i < 5; i++) Phi[k][i] = lam[i];
        }
        float rhs[5] = {phi[0], phi[1], phi[2], phi[3] + gQ * (p[P_K0] * s.mq0 + p[P_K1]) * s.c3,
                        phi[4] + gQ * p[P_K2] * s.c4};
        s.solve_t(1.0f, rhs, lam);
#pragma unroll
        for (int i = 0; i < 5; i++) phi[i] = lam[i];
    }
    if (L.active) {
        float *dst = phi_ws + ((int64_t)chunk * 30) * N + L.n;
#pragma unroll
        for (int k = 0; k < 5; k++)
#pragma unroll
            for (int i = 0; i < 5; i++) dst[(int64_t)(k * 5 + i) * N] = Phi[k][i];
#pragma unroll
        for (int i = 0; i < 5; i++) dst[(int64_t)(25 + i) * N] = phi[i];
    }
}

template <bool BETAET, bool FEW = false>
__global__ void __launch_bounds__(64) k_adj_chunk_sweep(const hbvx_desc d, const hbvx_bwd_io io, int lgMp,
                                                        int C, const float *abnd, float *gpart, int per_xcd, const AdjFew F)
{
    constexpr int NP = BETAET ? 13 : 12;
    ChunkBlock blk;
    if (!chunk_block(d, lgMp, per_xcd, blk)) return;
    const AdjLane L = adj_lane(d, lgMp, blk.bx);
    const int chunk = blk.chunk;
    const int T = d.T, t0 = chunk * C, t1 = min(T, t0 + C);
    const int64_t N = (int64_t)d.B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const float invM = 1.0f / (float)d.M;
    float usta[NP], gsta[NP];       // FEW: usta holds the static PHYSICAL values
    bool use_dyn[FEW ? ADJ_FEW : NP];
    float gused[ADJ_FEW] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        usta[i] = raw ? sigmoid_(v) : v;
        if (FEW) usta[i] = descale_(usta[i], s.lo, s.hi);
        else use_dyn[i] = s.dyn && !(s.drop && s.drop[L.n]);
        gsta[i] = 0.0f;
    }
    if (FEW) {
#pragma unroll
        for (int k = 0; k < ADJ_FEW; k++) {
            use_dyn[k] = false;
            if (k < F.nd) {
                const hbvx_param_src &s = d.p[F.slot[k]];
                use_dyn[k] = !(s.drop && s.drop[L.n]);
            }
        }
    }
    float a[5];
#pragma unroll
    for (int k = 0; k < 5; k++) a[k] = abnd[((int64_t)chunk * 5 + k) * N + L.n];
    // (no one-day-ahead prefetch here: its nine registers take this kernel from 165 to 171 VGPRs, i.e. from three waves
    // per SIMD to two -- measured 3.53 instead of 3.15 ms for the adjoint of config 4; round 4, slot-list instance, where
    // the nine fit three waves (154 -> 163): still slower, 2.34 -> 2.51 ms)
    for (int t = t1 - 1; t >= t0; t--) {
        AdjRaw Rc;
        adj_issue(d, io, L, t, N, Rc);
        float u[NP], p[NPARAM_MAX], x[5], gp[NPARAM_MAX];
        if (FEW) adj_params_few<NP>(d, L, t, raw, F, usta, use_dyn, u, p);
        else adj_params<NP>(d, L, t, raw, usta, use_dyn, u, p);
        AdjStep<BETAET> s;
        s.P = Rc.f[0]; s.Tf = Rc.f[1]; s.PET = Rc.f[2];
#pragma unroll
        for (int k = 0; k < 5; k++) x[k] = Rc.x[k];
        const float gQ = adj_gq(io, Rc) * invM;
#pragma unroll
        for (int i = 0; i < NPARAM_MAX; i++) gp[i] = 0.0f;
        adj_backstep<BETAET>(s, p, x, 1.0f, gQ, a, gp);
        if (FEW) {
#pragma unroll
            for (int i = 0; i < NP; i++) gsta[i] += gp[i] * (d.p[i].hi - d.p[i].lo);
#pragma unroll
            for (int k = 0; k < ADJ_FEW; k++)
                if (k < F.nd) {
                    const int sl = F.slot[k];
                    const float gu = slot_get<NP>(gp, sl) * (d.p[sl].hi - d.p[sl].lo);
                    const float gr = raw ? gu * (u[k] * (1.0f - u[k])) : gu;
                    if (io.g[sl].dyn && L.active)
                        io.g[sl].dyn[(int64_t)t * io.g[sl].dyn_t_stride + (int64_t)L.b * io.g[sl].dyn_b_stride + L.j] =
                            use_dyn[k] ? gr : 0.0f;
                    gused[k] += use_dyn[k] ? gu : 0.0f;   // goes to the dynamic rows, not to the static one
                }
            continue;
        }
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const float gu = gp[i] * (d.p[i].hi - d.p[i].lo);
            if (d.p[i].dyn) {
                const float gr = raw ? gu * (u[i] * (1.0f - u[i])) : gu;
                if (io.g[i].dyn && L.active)
                    io.g[i].dyn[(int64_t)t * io.g[i].dyn_t_stride + (int64_t)L.b * io.g[i].dyn_b_stride + L.j] =
                        use_dyn[i] ? gr : 0.0f;
                gsta[i] += use_dyn[i] ? 0.0f : gu;
            } else {
                gsta[i] += gu;
            }
        }
    }
    if (FEW) {
        // the same daily terms were added to gsta and gused in the same order: exact cancellation
#pragma unroll
        for (int k = 0; k < ADJ_FEW; k++)
            if (k < F.nd) slot_set<NP>(gsta, F.slot[k], slot_get<NP>(gsta, F.slot[k]) - gused[k]);
    }
    if (L.active) {
#pragma unroll
        for (int i = 0; i < NP; i++) gpart[((int64_t)chunk * NP + i) * N + L.n] = gsta[i];
    }
}

} // namespace hbvx
