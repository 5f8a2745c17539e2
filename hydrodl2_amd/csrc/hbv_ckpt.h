// hbv_ckpt.h -- adjoint from K-day checkpoints (HBVX_TRAJ_CKPT): the memory-lean form of the path.
//
// The default build of the path saves the whole state trajectory (28 bytes per lane-day) because on
// this chip the time loops are bound by dependent-instruction latency, not by HBM, and every value
// the adjoint does not have to recompute shortens its chain.  That costs memory: 100 000 basins x
// 16 members x 7 300 days would need 327 GB.  Here the forward keeps only the five storages entering
// every K-th day (20 / K bytes per lane-day) and the adjoint re-materialises a segment at a time:
//
//   for segment s = last .. first:
//       storages <- checkpoint s
//       for t in segment, forward:   step (both powers recomputed), storages + powers -> LDS [K][7][64]
//       for t in segment, backward:  the ordinary adjoint day from the LDS copy (same Step<> code, so
//                                    every branch predicate matches the forward's), static-parameter
//                                    gradients accumulate in registers, dynamic ones are written
//
// One wavefront per 64 lanes, any model / dynamic set / ensemble weights (the parameter handling of
// the generic one-wave kernels of hbvx.hip).  K <= 16 (LDS: K x 1 792 bytes per wave).
// Cost against the saved-trajectory adjoint: one extra forward step per day (two more pow calls).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/hbvx.h"
#include "hbv_step.h"
#include "hbv_tiled.h"

namespace hbvx {

#define CKPT_MAXK 16

struct CkptBwdArgs {
    hbvx_desc d;
    hbvx_bwd_io io;
    int lgMp;
    int K;
};

template <int MODEL, bool BETAET>
__global__ void __launch_bounds__(64) k_bwd_ckpt(const CkptBwdArgs A)
{
    constexpr int NP = NParamT<MODEL, BETAET>::value;
    extern __shared__ float seg_lds[];   // [K][7][64]
    const hbvx_desc &d = A.d;
    const hbvx_bwd_io &io = A.io;
    const int lgMp = A.lgMp, K = A.K;
    const LaneT L = lane_t(d, lgMp);
    const int T = d.T;
    const int64_t N = (int64_t)d.B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const float nz = d.nearzero;
    const float ac = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.ac[L.b] : 0.0f;
    const float elev = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.elev[L.b] : 0.0f;
    const int nf = io.n_flux;
    const bool leader = L.active && L.jm == 0;
    float *const my = seg_lds + L.lane;

    float psta[NP], usta[NP], gsta[NP];
    const float *dynp[NP];
    float *gdyn[NP];
    bool use_dyn[NP];
    unsigned dmask = 0;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        usta[i] = raw ? sigmoid_(v) : v;
        psta[i] = descale_(usta[i], s.lo, s.hi);
        dynp[i] = s.dyn ? s.dyn + (int64_t)L.b * s.dyn_b_stride + L.j : s.sta;
        gdyn[i] = io.g[i].dyn ? io.g[i].dyn + (int64_t)L.b * io.g[i].dyn_b_stride + L.j : nullptr;
        use_dyn[i] = s.dyn && !(s.drop && s.drop[L.b]);
        gsta[i] = 0.0f;
        if (s.dyn) dmask |= 1u << i;
    }
    const float *xb = d.x + (int64_t)L.b * d.x_b_stride;
    const float *mu = d.muwts ? d.muwts + (int64_t)L.b * d.mu_b_stride + L.j : nullptr;
    const float invM = 1.0f / (float)d.M;
    float a[5];
#pragma unroll
    for (int k = 0; k < 5; k++) a[k] = io.grad_state_out ? io.grad_state_out[k * N + L.n] : 0.0f;

    // raw inputs of a day (forcings, raw dynamic values): loaded one day ahead of their use so that no
    // day waits for HBM
    struct Raw {
        float f[3], dv[NP];
        float gf[HBVX_MAX_FLUX];   // incoming flux-series gradients (adjoint sweep only)
        float g4[4];               // the routing adjoint's share, added where it is used (hbv_chunked.h::chunk_issue)
    };
    const int64_t fs = (int64_t)T * d.B;
    auto fetch_grad = [&](int t, Raw &R) {
        const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
        const int64_t go = (int64_t)tc * d.B + L.b;
#pragma unroll
        for (int k = 0; k < HBVX_MAX_FLUX; k++) {
            R.gf[k] = (io.grad_flux && k < nf) ? io.grad_flux[k * fs + go] : 0.0f;
            if (k < 4) R.g4[k] = io.grad_flux4 ? io.grad_flux4[k * fs + go] : 0.0f;
        }
    };
    auto fetch = [&](int t, Raw &R) {
        const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
        const float *xr = xb + (int64_t)tc * d.x_t_stride;
        R.f[0] = xr[d.ch_prcp]; R.f[1] = xr[d.ch_tmean]; R.f[2] = xr[d.ch_pet];
#pragma unroll
        for (int i = 0; i < NP; i++)
            R.dv[i] = ((dmask >> i) & 1) ? dynp[i][(int64_t)tc * d.p[i].dyn_t_stride] : 0.0f;
    };
    // parameters of a day -> p[], unit values -> ud[]
    auto params_of = [&](const Raw &R, float *p, float *ud) {
#pragma unroll
        for (int i = 0; i < NPARAM_MAX; i++) p[i] = 0.0f;
#pragma unroll
        for (int i = 0; i < NP; i++) {
            p[i] = psta[i];
            ud[i] = usta[i];
            if ((dmask >> i) & 1) {
                const float v = raw ? sigmoid_dyn_(R.dv[i]) : R.dv[i];
                if (use_dyn[i]) {
                    ud[i] = v;
                    p[i] = descale_(v, d.p[i].lo, d.p[i].hi);
                }
            }
        }
    };

    const int nseg = (T + K - 1) / K;
    for (int seg = nseg - 1; seg >= 0; seg--) {
        const int t0 = seg * K, t1 = min(T, t0 + K);
        float st[5];
#pragma unroll
        for (int k = 0; k < 5; k++) st[k] = io.traj[((int64_t)seg * 5 + k) * N + L.n];
        // forward over the segment: storages entering each day and the two powers -> LDS
        Raw nxt;
        fetch(t0, nxt);
        for (int t = t0; t < t1; t++) {
            Step<MODEL, BETAET> s;
            const Raw cur = nxt;
            fetch(t + 1 < t1 ? t + 1 : t1 - 1, nxt);   // after the last day: the first day of the sweep below
            if (t + 1 >= t1) fetch_grad(t1 - 1, nxt);
            s.P = cur.f[0]; s.Tf = cur.f[1]; s.PET = cur.f[2];
            float p[NPARAM_MAX], ud[NP];
            params_of(cur, p, ud);
            s.SP = st[0]; s.MW = st[1]; s.SM = st[2]; s.SUZ = st[3]; s.SLZ = st[4];
            s.template fwd<false, true>(p, nz, ac, elev, 0.0f, 0.0f);
            float *row = my + (t - t0) * 7 * 64;
#pragma unroll
            for (int k = 0; k < 5; k++) row[k * 64] = st[k];
            row[5 * 64] = s.sw0;
            row[6 * 64] = s.ef0;
            st[0] = s.SP3; st[1] = s.MW3; st[2] = s.SM4; st[3] = s.SUZ4; st[4] = s.SLZ2;
        }
        // adjoint over the segment, last day first
        for (int t = t1 - 1; t >= t0; t--) {
            Step<MODEL, BETAET> s;
            const Raw cur = nxt;
            fetch(t - 1 >= t0 ? t - 1 : t0, nxt);
            fetch_grad(t - 1 >= t0 ? t - 1 : t0, nxt);
            s.P = cur.f[0]; s.Tf = cur.f[1]; s.PET = cur.f[2];
            float p[NPARAM_MAX], ud[NP];
            params_of(cur, p, ud);
            const float *row = my + (t - t0) * 7 * 64;
            s.SP = row[0]; s.MW = row[64]; s.SM = row[128]; s.SUZ = row[192]; s.SLZ = row[256];
            s.template fwd<true>(p, nz, ac, elev, row[5 * 64], row[6 * 64]);

            FluxGrad g;
            auto GF = [&](int k) -> float { return (k < 4 && io.grad_flux4) ? cur.gf[k] + cur.g4[k < 4 ? k : 0] : cur.gf[k]; };
            const float gq = GF(HBVX_F_QSIM);
            const float wq = mu ? mu[(int64_t)t * d.mu_t_stride] : invM;
            g.gQ = gq * wq;
            g.gQ0 = GF(HBVX_F_Q0) * invM;
            g.gQ1 = GF(HBVX_F_Q1) * invM;
            g.gQ2 = GF(HBVX_F_Q2) * invM;
            g.gET = GF(HBVX_F_AET) * invM;
            g.gSWE = GF(HBVX_F_SWE) * invM;
            g.grech = GF(HBVX_F_RECHARGE) * invM;
            g.gexc = GF(HBVX_F_EXCS) * invM;
            g.gef = GF(HBVX_F_EVAPFACTOR) * invM;
            g.gtosoil = GF(HBVX_F_TOSOIL) * invM;
            g.gPERC = GF(HBVX_F_PERC) * invM;
            g.gcap = (nf > HBVX_F_CAPILLARY) ? GF(HBVX_F_CAPILLARY) * invM : 0.0f;
            if (io.grad_muwts && L.active) io.grad_muwts[((int64_t)t * d.B + L.b) * d.M + L.j] = gq * s.Q;

            float gp[NPARAM_MAX], gx[3];
#pragma unroll
            for (int i = 0; i < NPARAM_MAX; i++) gp[i] = 0.0f;
            s.bwd(p, nz, g, a, gp, gx);
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const float gu = gp[i] * (d.p[i].hi - d.p[i].lo);
                if ((dmask >> i) & 1) {
                    const float gr = raw ? gu * (ud[i] * (1.0f - ud[i])) : gu;
                    if (gdyn[i] && L.active)
                        gdyn[i][(int64_t)t * io.g[i].dyn_t_stride] = use_dyn[i] ? gr : 0.0f;
                    gsta[i] += use_dyn[i] ? 0.0f : gu;
                } else {
                    gsta[i] += gu;
                }
            }
            if (io.grad_x) {
                const float act = L.active ? 1.0f : 0.0f;
                float gs[3] = {gx[0] * act, gx[1] * act, gx[2] * act};
                for (int sft = 0; sft < lgMp; sft++) {
#pragma unroll
                    for (int c = 0; c < 3; c++) gs[c] += __shfl_xor(gs[c], 1 << sft, 64);
                }
                if (leader) {
                    float *gr = io.grad_x + (int64_t)t * d.x_t_stride + (int64_t)L.b * d.x_b_stride;
                    gr[d.ch_prcp] = gs[0]; gr[d.ch_tmean] = gs[1]; gr[d.ch_pet] = gs[2];
                }
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

// ---------------------------------------------------------------------------------------------
// Re-materialisation for the block-wise adjoint (launch_ckpt.hip): one wavefront per (64 lanes,
// K-day segment) of a block of days [t0, t0 + tb) steps forward from its checkpoint and writes the
// block's trajectory rows [5, tb + 1, N] and powers [2, tb, N] (HBVX_TRAJ_ROWS, local day index) into
// scratch.  All segments are independent: thousands of waves, no serial chain longer than K days.
// The regular adjoint kernels then run on the block as on a tb-day record.
// ---------------------------------------------------------------------------------------------
struct RematArgs {
    hbvx_desc d;          // the FULL problem (T, strides, parameter sources)
    const float *ckpt;    // [ceil(T/K), 5, N]
    float *traj, *aux;    // scratch rows of the block
    int lgMp, K, t0, tb;
};

template <int MODEL, bool BETAET>
__global__ void __launch_bounds__(64) k_ckpt_remat(const RematArgs A)
{
    constexpr int NP = NParamT<MODEL, BETAET>::value;
    const hbvx_desc &d = A.d;
    const LaneT L = lane_t(d, A.lgMp);
    const int K = A.K, tb = A.tb;
    const int64_t N = (int64_t)d.B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const float nz = d.nearzero;
    const float ac = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.ac[L.b] : 0.0f;
    const float elev = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.elev[L.b] : 0.0f;
    const int l0 = blockIdx.y * K;                     // first local day of this segment
    const int l1 = min(tb, l0 + K);
    const int seg = (A.t0 + l0) / K;                   // t0 is a multiple of K

    float p[NPARAM_MAX], psta[NP];
    const float *dynp[NP];
    bool use_dyn[NP];
    unsigned dmask = 0;
#pragma unroll
    for (int i = 0; i < NPARAM_MAX; i++) p[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        v = raw ? sigmoid_(v) : v;
        psta[i] = descale_(v, s.lo, s.hi);
        dynp[i] = s.dyn ? s.dyn + (int64_t)L.b * s.dyn_b_stride + L.j : s.sta;
        use_dyn[i] = s.dyn && !(s.drop && s.drop[L.b]);
        if (s.dyn) dmask |= 1u << i;
    }
    const float *xb = d.x + (int64_t)L.b * d.x_b_stride;
    float st[5];
#pragma unroll
    for (int k = 0; k < 5; k++) st[k] = A.ckpt[((int64_t)seg * 5 + k) * N + L.n];

    float nf[3], nd[NP];
    auto fetch = [&](int t) {
        const float *xr = xb + (int64_t)t * d.x_t_stride;
        nf[0] = xr[d.ch_prcp]; nf[1] = xr[d.ch_tmean]; nf[2] = xr[d.ch_pet];
#pragma unroll
        for (int i = 0; i < NP; i++) nd[i] = ((dmask >> i) & 1) ? dynp[i][(int64_t)t * d.p[i].dyn_t_stride] : 0.0f;
    };
    fetch(A.t0 + l0);
    for (int l = l0; l < l1; l++) {
        Step<MODEL, BETAET> s;
        s.P = nf[0]; s.Tf = nf[1]; s.PET = nf[2];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            p[i] = psta[i];
            if ((dmask >> i) & 1) {
                const float v = raw ? sigmoid_dyn_(nd[i]) : nd[i];
                if (use_dyn[i]) p[i] = descale_(v, d.p[i].lo, d.p[i].hi);
            }
        }
        if (l + 1 < l1) fetch(A.t0 + l + 1);
        s.SP = st[0]; s.MW = st[1]; s.SM = st[2]; s.SUZ = st[3]; s.SLZ = st[4];
        s.template fwd<false, true>(p, nz, ac, elev, 0.0f, 0.0f);
        if (L.active) {
#pragma unroll
            for (int k = 0; k < 5; k++) A.traj[((int64_t)k * (tb + 1) + l) * N + L.n] = st[k];
            if (SAVE_POW) {
                A.aux[(int64_t)l * N + L.n] = s.sw0;
                A.aux[((int64_t)tb + l) * N + L.n] = s.ef0;
            }
        }
        st[0] = s.SP3; st[1] = s.MW3; st[2] = s.SM4; st[3] = s.SUZ4; st[4] = s.SLZ2;
    }
}

} // namespace hbvx
