// hbv_stream.h -- forward for LARGE grids (thousands of wavefronts of state, e.g. 12 500+ basins x 16).
//
// With 3 000+ independent wavefronts the time loop is throughput-bound, not latency-bound: the
// right shape is the plain one -- one wavefront per 64 lanes, no LDS, no helper waves, as many
// wavefronts per SIMD as the registers allow -- provided the wave never stalls on its own memory
// traffic.  On CDNA loads and stores retire through one in-order counter (vmcnt), so a wave that
// consumes a load issued one day ago also waits for every store issued before that load.  The ring
// below keeps the inputs of STREAM_D days in registers: by the time a day's loads are consumed the
// stores that precede them are STREAM_D days old.  All memory operations use buffer addressing
// (descriptor + scalar day offset + constant per-lane offset, out-of-range lanes dropped), so the
// loop has no vector address arithmetic, no EXEC masking and no branches around memory operations.
// Ensemble means: DPP butterflies inside the wave with the same add tree as the tiled kernels
// (quad, quad pair, row halves ...), so results are bit-identical to them.
//
// Supports all explicit variants with static parameters or up to three dynamic ones (slot list,
// as in the time-parallel adjoint); muwts and larger dynamic sets stay on the tiled kernel.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/hbvx.h"
#include "hbv_step.h"
#include "hbv_tiled.h"

namespace hbvx {

#define STREAM_D 4
#ifndef STREAM_EXP
#define STREAM_EXP 0   // dev experiments: 1 no flux stores, 2 no ensemble reduction, 4 no trajectory stores
#endif

struct StreamArgs {
    hbvx_desc d;
    hbvx_fwd_out o;
    int lgMp;
    int nd;
    int dslot[6];      // first generation: up to 3; hbv_stream2.h run-time lists: up to STREAM2_LIST_MAX
    int per_xcd;   // hbv_stream2.h: basin groups per XCD (grid = 8 * per_xcd)
};

template <int CTRL>
__device__ __forceinline__ float dpp_(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// sums over the Mp lanes of a basin for NV values at once (every lane of the basin gets the sums);
// the wave-uniform tests on lgMp are outside the loops over the values: 6 scalar branches per call
// instead of 6 per value
template <int NV>
__device__ __forceinline__ void ens_sum_dpp(float *v, int lgMp)
{
    if (lgMp >= 1) {
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] += dpp_<0xB1>(v[k]);   // quad_perm [1,0,3,2]
    }
    if (lgMp >= 2) {
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] += dpp_<0x4E>(v[k]);   // quad_perm [2,3,0,1]
    }
    if (lgMp >= 3) {
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] += dpp_<0x141>(v[k]);  // row_half_mirror: the other quad of the 8
    }
    if (lgMp >= 4) {
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] += dpp_<0x140>(v[k]);  // row_mirror: the other half of the 16
    }
    if (lgMp >= 5) {
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] += __shfl_xor(v[k], 16, 64);
    }
    if (lgMp >= 6) {
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] += __shfl_xor(v[k], 32, 64);
    }
}

template <int MODEL, bool BETAET, bool TRAJ, bool FEW>
__global__ void __launch_bounds__(64) k_fwd_stream(const StreamArgs A)
{
    constexpr int NP = NParamT<MODEL, BETAET>::value;
    constexpr int NF = MODEL == MODEL_HBV10 ? 11 : 12;
    constexpr int D = STREAM_D;
    const hbvx_desc &d = A.d;
    const hbvx_fwd_out &o = A.o;
    const int lgMp = A.lgMp;
    const LaneT L = lane_t(d, lgMp);
    const int T = d.T, B = d.B;
    const int64_t N = (int64_t)B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const float nz = d.nearzero, invM = 1.0f / (float)d.M;
    const float ac = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.ac[L.b] : 0.0f;
    const float elev = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.elev[L.b] : 0.0f;
    const bool leader = L.active && L.jm == 0;

    float p[NPARAM_MAX];
#pragma unroll
    for (int i = 0; i < NPARAM_MAX; i++) p[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        v = raw ? sigmoid_(v) : v;
        p[i] = descale_(v, s.lo, s.hi);
    }

    auto rsrc = [&](const void *base) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, -1, 0x00020000);
    };
    auto bload = [&](__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
    };
    auto bstore = [&](__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so, float v) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, vo, so, 0);
    };
    const unsigned OOB = 0xFFFFFFFFu;

    // inputs
    const auto rx = rsrc(d.x);
    const unsigned xvo = (unsigned)(L.b * d.x_b_stride * 4);
    const unsigned xts = (unsigned)(d.x_t_stride * 4);
    const unsigned xcp = d.ch_prcp * 4, xct = d.ch_tmean * 4, xce = d.ch_pet * 4;
    __amdgpu_buffer_rsrc_t rd[3];
    unsigned dvo[3], dts[3];
    float dlo[3], dhi[3], dsta[3];
    bool duse[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        rd[k] = rx; dvo[k] = 0; dts[k] = 0; dlo[k] = dhi[k] = dsta[k] = 0.0f; duse[k] = false;
        if (FEW && k < A.nd) {
            const hbvx_param_src &ps = d.p[A.dslot[k]];
            rd[k] = rsrc(ps.dyn);
            dvo[k] = (unsigned)((L.b * ps.dyn_b_stride + L.j) * 4);
            dts[k] = (unsigned)(ps.dyn_t_stride * 4);
            dlo[k] = ps.lo; dhi[k] = ps.hi;
            dsta[k] = p[A.dslot[k]];
            duse[k] = !(ps.drop && ps.drop[L.b]);
        }
    }
    // outputs
    const auto rtraj = rsrc(o.traj), raux = rsrc(SAVE_POW ? o.aux : o.traj), rflux = rsrc(o.flux);
    unsigned tvo[5], avo[2];
#pragma unroll
    for (int k = 0; k < 5; k++) tvo[k] = (TRAJ && L.active) ? (unsigned)((k * (int64_t)(T + 1) * N + L.n) * 4) : OOB;
#pragma unroll
    for (int k = 0; k < 2; k++) avo[k] = (TRAJ && L.active) ? (unsigned)((k * (int64_t)T * N + L.n) * 4) : OOB;
    const unsigned row4 = (unsigned)(N * 4);
    const unsigned fvo = leader ? (unsigned)(L.b * 4) : OOB;
    const unsigned fT = (unsigned)((int64_t)T * B * 4), fB = (unsigned)(B * 4);

    float st[5];
#pragma unroll
    for (int k = 0; k < 5; k++) st[k] = d.state_in ? d.state_in[k * N + L.n] : 0.001f;

    float fx[D], fy[D], fz[D], dv[D][3];
    auto issue = [&](int t, int j) {
        const unsigned tc = (unsigned)min(t, T - 1);
        const unsigned so = tc * xts;
        fx[j] = bload(rx, xvo, so + xcp); fy[j] = bload(rx, xvo, so + xct); fz[j] = bload(rx, xvo, so + xce);
        if (FEW) {
#pragma unroll
            for (int k = 0; k < 3; k++) dv[j][k] = bload(rd[k], dvo[k], tc * dts[k]);
        }
    };
    auto day = [&](int t, int j) {
        Step<MODEL, BETAET> s;
        s.P = fx[j]; s.Tf = fy[j]; s.PET = fz[j];
        if (FEW) {
#pragma unroll
            for (int k = 0; k < 3; k++)
                if (k < A.nd) {
                    const float u = raw ? sigmoid_dyn_(dv[j][k]) : dv[j][k];
                    // wave-uniform index: stays in VGPRs (s_set_gpr_idx / v_movrel; checked: no scratch)
                    p[A.dslot[k]] = duse[k] ? descale_(u, dlo[k], dhi[k]) : dsta[k];
                }
        }
        s.SP = st[0]; s.MW = st[1]; s.SM = st[2]; s.SUZ = st[3]; s.SLZ = st[4];
        s.template fwd<false, true>(p, nz, ac, elev, 0.0f, 0.0f);
        const unsigned so = (unsigned)t * row4;
        if (TRAJ && !(STREAM_EXP & 4)) {
#pragma unroll
            for (int k = 0; k < 5; k++) bstore(rtraj, tvo[k], so, st[k]);
            if (SAVE_POW) {
                bstore(raux, avo[0], so, s.sw0);
                bstore(raux, avo[1], so, s.ef0);
            }
        }
        st[0] = s.SP3; st[1] = s.MW3; st[2] = s.SM4; st[3] = s.SUZ4; st[4] = s.SLZ2;
        const float act = L.active ? 1.0f : 0.0f;
        float f[HBVX_MAX_FLUX];
        f[HBVX_F_QSIM] = s.Q; f[HBVX_F_Q0] = s.Q0; f[HBVX_F_Q1] = s.Q1; f[HBVX_F_Q2] = s.Q2;
        f[HBVX_F_AET] = s.ET; f[HBVX_F_SWE] = s.SP3; f[HBVX_F_RECHARGE] = s.rech; f[HBVX_F_EXCS] = s.exc;
        f[HBVX_F_EVAPFACTOR] = s.ef; f[HBVX_F_TOSOIL] = s.tosoil; f[HBVX_F_PERC] = s.PERC;
        f[HBVX_F_CAPILLARY] = s.cap;
#pragma unroll
        for (int k = 0; k < NF; k++) f[k] *= act;
        if (!(STREAM_EXP & 2)) ens_sum_dpp<NF>(f, lgMp);
        unsigned fso = (unsigned)t * fB;
        if (STREAM_EXP & 1) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < NF; k++) acc += f[k];
            if (acc == 123.456f) bstore(rflux, fvo, fso, acc);
        } else {
#pragma unroll
            for (int k = 0; k < NF; k++) {
                bstore(rflux, fvo, fso, f[k] * invM);
                fso += fT;
            }
        }
    };

#pragma unroll
    for (int j = 0; j < D; j++) issue(j, j);
    // Full groups are straight-line code: with a branch around a day the compiler's wait-counter
    // bookkeeping turns conservative (vmcnt(0) at the top of every day = wait for yesterday's
    // stores), which defeats the ring.
    int t0 = 0;
    for (; t0 + D <= T; t0 += D) {
#pragma unroll
        for (int j = 0; j < D; j++) {
            day(t0 + j, j);
            issue(t0 + j + D, j);
        }
    }
#pragma unroll
    for (int j = 0; j < D; j++)
        if (t0 + j < T) day(t0 + j, j);
    if (L.active) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            o.state_out[k * N + L.n] = st[k];
            if (TRAJ) o.traj[((int64_t)k * (T + 1) + T) * N + L.n] = st[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Adjoint for large grids: the plain serial sweep (ONE pass over the trajectory, where the
// time-parallel scheme needs two), one wavefront per 64 lanes, same register ring / buffer
// addressing.  With thousands of wavefronts there is no need to parallelise over time.
// Per-lane static-parameter gradients accumulate in registers over the whole record and go straight
// into the gradient row (each lane owns its element: deterministic, no partials, no workspace).
// ---------------------------------------------------------------------------------------------
#define STREAM_DB 2

struct StreamBwdArgs {
    hbvx_desc d;
    hbvx_bwd_io io;
    int lgMp;
    int nd;
    int dslot[6];      // first generation: up to 3; hbv_stream2.h run-time lists: up to STREAM2_LIST_MAX
    int per_xcd;
};

template <int MODEL, bool BETAET, bool FEW, bool GFULL>
__global__ void __launch_bounds__(64) k_bwd_stream(const StreamBwdArgs A)
{
    constexpr int NP = NParamT<MODEL, BETAET>::value;
    constexpr int NF = MODEL == MODEL_HBV10 ? 11 : 12;
    constexpr int NG = GFULL ? NF : 4;      // flux series that carry gradient
    constexpr int D = STREAM_DB;
    const hbvx_desc &d = A.d;
    const hbvx_bwd_io &io = A.io;
    const int lgMp = A.lgMp;
    const LaneT L = lane_t(d, lgMp);
    const int T = d.T, B = d.B;
    const int64_t N = (int64_t)B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const float nz = d.nearzero, invM = 1.0f / (float)d.M;
    const float ac = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.ac[L.b] : 0.0f;
    const float elev = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.elev[L.b] : 0.0f;
    const bool leader = L.active && L.jm == 0;
    const unsigned OOB = 0xFFFFFFFFu;

    float p[NPARAM_MAX], usta[NP], gsta[NP];
#pragma unroll
    for (int i = 0; i < NPARAM_MAX; i++) p[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        const float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        usta[i] = raw ? sigmoid_(v) : v;
        p[i] = descale_(usta[i], s.lo, s.hi);
        gsta[i] = 0.0f;
    }

    auto rsrc = [&](const void *base) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, -1, 0x00020000);
    };
    auto bload = [&](__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
    };
    auto bstore = [&](__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so, float v) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, vo, so, 0);
    };

    const auto rx = rsrc(d.x), rtraj = rsrc(io.traj), raux = rsrc(SAVE_POW ? io.aux : io.traj);
    const auto rgf = rsrc(io.grad_flux ? io.grad_flux : io.grad_flux4);
    const auto rg4 = rsrc(io.grad_flux4 ? io.grad_flux4 : io.grad_flux);
    const bool has_gf = io.grad_flux != nullptr, has_g4 = io.grad_flux4 != nullptr;
    const unsigned xvo = (unsigned)(L.b * d.x_b_stride * 4), xts = (unsigned)(d.x_t_stride * 4);
    const unsigned xcp = d.ch_prcp * 4, xct = d.ch_tmean * 4, xce = d.ch_pet * 4;
    unsigned tvo[5], avo[2];
#pragma unroll
    for (int k = 0; k < 5; k++) tvo[k] = (unsigned)((k * (int64_t)(T + 1) * N + L.n) * 4);
#pragma unroll
    for (int k = 0; k < 2; k++) avo[k] = (unsigned)((k * (int64_t)T * N + L.n) * 4);
    const unsigned row4 = (unsigned)(N * 4);
    const unsigned gvo = (unsigned)(L.b * 4);
    const unsigned fT = (unsigned)((int64_t)T * B * 4), fB = (unsigned)(B * 4);

    __amdgpu_buffer_rsrc_t rd[3], rgd[3];
    unsigned dvo[3], dts[3], gdvo[3], gdts[3];
    float dlo[3], dhi[3], dsta[3], gused[3];
    bool duse[3], dgrad[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        rd[k] = rgd[k] = rx; dvo[k] = dts[k] = gdts[k] = 0; gdvo[k] = OOB;
        dlo[k] = dhi[k] = dsta[k] = gused[k] = 0.0f; duse[k] = dgrad[k] = false;
        if (FEW && k < A.nd) {
            const int sl = A.dslot[k];
            const hbvx_param_src &ps = d.p[sl];
            rd[k] = rsrc(ps.dyn);
            dvo[k] = (unsigned)((L.b * ps.dyn_b_stride + L.j) * 4);
            dts[k] = (unsigned)(ps.dyn_t_stride * 4);
            dlo[k] = ps.lo; dhi[k] = ps.hi;
            dsta[k] = p[sl];
            duse[k] = !(ps.drop && ps.drop[L.b]);
            dgrad[k] = io.g[sl].dyn != nullptr;
            if (dgrad[k]) {
                rgd[k] = rsrc(io.g[sl].dyn);
                gdvo[k] = L.active ? (unsigned)((L.b * io.g[sl].dyn_b_stride + L.j) * 4) : OOB;
                gdts[k] = (unsigned)(io.g[sl].dyn_t_stride * 4);
            }
        }
    }
    const auto rgx = rsrc(io.grad_x ? io.grad_x : const_cast<float *>(d.x));
    const bool has_gx = io.grad_x != nullptr;
    const unsigned gxvo = leader ? xvo : OOB;

    float a[5];
#pragma unroll
    for (int k = 0; k < 5; k++) a[k] = io.grad_state_out ? io.grad_state_out[k * N + L.n] : 0.0f;

    float fx[D][3], st[D][5], ax[D][2], gf[D][NG], g4[D][4], dv[D][3];   // g4: added where it is used (GF), see hbv_stream2.h
    auto issue = [&](int t, int j) {
        const unsigned tc = (unsigned)max(t, 0);
        const unsigned so = tc * xts, sr = tc * row4, sg = tc * fB;
        fx[j][0] = bload(rx, xvo, so + xcp); fx[j][1] = bload(rx, xvo, so + xct); fx[j][2] = bload(rx, xvo, so + xce);
#pragma unroll
        for (int k = 0; k < 5; k++) st[j][k] = bload(rtraj, tvo[k], sr);
        if (SAVE_POW) { ax[j][0] = bload(raux, avo[0], sr); ax[j][1] = bload(raux, avo[1], sr); }
        else ax[j][0] = ax[j][1] = 0.0f;
#pragma unroll
        for (int k = 0; k < NG; k++) {
            gf[j][k] = 0.0f;
            if (GFULL) { if (has_gf) gf[j][k] = bload(rgf, gvo, sg + (unsigned)k * fT); }
            if (k < 4) g4[j][k] = has_g4 ? bload(rg4, gvo, sg + (unsigned)k * fT) : 0.0f;
        }
        if (FEW) {
#pragma unroll
            for (int k = 0; k < 3; k++) dv[j][k] = bload(rd[k], dvo[k], tc * dts[k]);
        }
    };
    auto day = [&](int t, int j) {
        Step<MODEL, BETAET> s;
        s.P = fx[j][0]; s.Tf = fx[j][1]; s.PET = fx[j][2];
        s.SP = st[j][0]; s.MW = st[j][1]; s.SM = st[j][2]; s.SUZ = st[j][3]; s.SLZ = st[j][4];
        float ud[3] = {0.0f, 0.0f, 0.0f};
        if (FEW) {
#pragma unroll
            for (int k = 0; k < 3; k++)
                if (k < A.nd) {
                    ud[k] = raw ? sigmoid_dyn_(dv[j][k]) : dv[j][k];
                    p[A.dslot[k]] = duse[k] ? descale_(ud[k], dlo[k], dhi[k]) : dsta[k];
                }
        }
        s.template fwd<SAVE_POW>(p, nz, ac, elev, ax[j][0], ax[j][1]);
        FluxGrad g;
        auto GF = [&](int k) -> float {
            if (k >= NG) return 0.0f;
            const float v = (k < 4 && has_g4) ? gf[j][k] + g4[j][k < 4 ? k : 0] : gf[j][k];
            return v * invM;
        };
        g.gQ = GF(HBVX_F_QSIM); g.gQ0 = GF(HBVX_F_Q0); g.gQ1 = GF(HBVX_F_Q1); g.gQ2 = GF(HBVX_F_Q2);
        g.gET = GF(HBVX_F_AET); g.gSWE = GF(HBVX_F_SWE); g.grech = GF(HBVX_F_RECHARGE);
        g.gexc = GF(HBVX_F_EXCS); g.gef = GF(HBVX_F_EVAPFACTOR); g.gtosoil = GF(HBVX_F_TOSOIL);
        g.gPERC = GF(HBVX_F_PERC); g.gcap = (NF > HBVX_F_CAPILLARY) ? GF(HBVX_F_CAPILLARY) : 0.0f;
        float gp[NPARAM_MAX], gx[3];
#pragma unroll
        for (int i = 0; i < NPARAM_MAX; i++) gp[i] = 0.0f;
        s.bwd(p, nz, g, a, gp, gx);
#pragma unroll
        for (int i = 0; i < NP; i++) gsta[i] += gp[i] * (d.p[i].hi - d.p[i].lo);
        if (FEW) {
#pragma unroll
            for (int k = 0; k < 3; k++)
                if (k < A.nd) {
                    const int sl = A.dslot[k];
                    const float gu = gp[sl] * (d.p[sl].hi - d.p[sl].lo);
                    const float gr = raw ? gu * (ud[k] * (1.0f - ud[k])) : gu;
                    if (dgrad[k]) bstore(rgd[k], gdvo[k], (unsigned)t * gdts[k], duse[k] ? gr : 0.0f);
                    gused[k] += duse[k] ? gu : 0.0f;
                }
        }
        if (has_gx) {
            const float act = L.active ? 1.0f : 0.0f;
            float gs[3] = {gx[0] * act, gx[1] * act, gx[2] * act};
            ens_sum_dpp<3>(gs, lgMp);
            const unsigned so = (unsigned)t * xts;
            bstore(rgx, gxvo, so + xcp, gs[0]); bstore(rgx, gxvo, so + xct, gs[1]); bstore(rgx, gxvo, so + xce, gs[2]);
        }
    };

#pragma unroll
    for (int j = 0; j < D; j++) issue(T - 1 - j, j);
    // (measured: unlike the forward, the adjoint is faster with the day under a branch -- it has
    // few stores, and the conservative waits cost less than the longer live ranges of the
    // straight-line schedule: 3.4 vs 3.9 ms at config 5)
    for (int t0 = T - 1; t0 >= 0; t0 -= D) {
#pragma unroll
        for (int j = 0; j < D; j++) {
            const int t = t0 - j;
            if (t >= 0) day(t, j);
            issue(t - D, j);
        }
    }
    if (FEW) {
        // the same daily terms went into gsta and gused in the same order: exact cancellation
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (k < A.nd) gsta[A.dslot[k]] = gsta[A.dslot[k]] - gused[k];
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
