// hbv_stream2_ckpt.h -- streaming adjoint from K-day checkpoints, the segment kept ON CHIP.
//
// north_star's form of the backward pass ("re-materialises states from checkpoints and accumulates the parameter
// gradients in registers") for the large-grid regime, where the streaming kernels sit at the memory pipe and 40 of
// the 76 bytes a lane-day moves are the saved trajectory (20 written by the forward, 20 read back here).  The
// forward (k_fwd_stream2, TRJ = 3) keeps only the five storages entering every K-th day: 20 / K bytes per lane-day.
// This kernel walks the record backwards one K-day segment at a time, one wavefront per 64 lanes, no scratch in HBM,
// no second kernel:
//
//   for segment s = last .. first:
//       storages <- checkpoint s                                            (5 loads per lane and SEGMENT)
//       R: for t in segment, forward:   storages entering day t and the day's de-scaled dynamic values -> LDS,
//                                       then the plain forward step (not for the segment's last day: the sweep
//                                       below recomputes each day's forward from its entering storages anyway)
//       B: for t in segment, backward:  the ordinary adjoint day (Step::fwd + Step::bwd: the forward's own
//                                       instruction sequence, so every clamp predicate sees the same bits) from the
//                                       LDS copy; dynamic-parameter gradients are stored, static ones accumulate in
//                                       LDS words the lane owns
//
// Bytes per lane-day against the saved-trajectory pair (M = 16 members, three dynamic parameters, loss on the runoff):
// the forward writes 20 / K instead of 20; this kernel reads 20 / K instead of 20, the dynamic rows ONCE (phase R
// stashes the unit values for phase B: 4 ND bytes of LDS per lane-day instead of a second trip to HBM) and the
// per-basin rows -- forcing, gradient series -- twice (12 / M + 16 / M bytes: nothing).  What it costs: K - 1 extra
// forward steps per K days on a wave whose day is ~2/3 adjoint arithmetic, and LDS: K (5 + ND) + NP rows of 256
// bytes per wave (K = 4, ND = 3, NP = 16: 12 KB -- twelve waves per CU, the three per SIMD the registers allow;
// K = 8: 20 KB, seven waves per CU).
//
// Loads run one ITEM ahead across the phase boundaries (R_t0 .. R_t1-1, B_t1-1 .. B_t0, checkpoint + R of the
// segment before): no segment starts with an exposed memory round trip.
#pragma once

#include "hbv_stream2.h"

namespace hbvx {

// LDS floats per wave (dynamic shared memory of the launch)
template <int NP, int ND>
__host__ __device__ constexpr int s2c_lds_floats(int K)
{
    return (K * (5 + ND) + NP) * 64;
}

template <int MODEL, bool BETAET, int SC, bool GFULL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3)))
k_bwd_stream2_ckpt(const StreamBwdArgs A, const int K)
{
    constexpr int NP = NParamT<MODEL, BETAET>::value;
    constexpr int NF = MODEL == MODEL_HBV10 ? 11 : 12;
    constexpr int NG = GFULL ? NF : 4;
    constexpr int ND = StreamDyn<SC>::nd;
    constexpr int NDA = ND > 0 ? ND : 1;
    extern __shared__ float s2c_lds[];
    const hbvx_desc &d = A.d;
    const hbvx_bwd_io &io = A.io;
    const int lgMp = A.lgMp;
    const S2Lane L = s2_lane(d, lgMp, A.per_xcd);
    if (!L.valid) return;
    const int T = d.T, B = d.B;
    const int64_t N = (int64_t)B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const float nz = d.nearzero, invM = 1.0f / (float)d.M;
    const float ac = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.ac[L.b] : 0.0f;
    const float elev = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.elev[L.b] : 0.0f;
    const bool leader = L.active && L.jm == 0;
    const unsigned OOB = 0xFFFFFFFFu;
    const int ln = threadIdx.x & 63;

    // LDS rows of this wave (each lane owns word `ln` of every row: no conflicts, no synchronisation)
    float *const seg_st = s2c_lds + ln;                    // [K][5] storages entering the segment's days
    float *const seg_ud = seg_st + K * 5 * 64;             // [K][ND] unit values of the dynamic parameters
    float *const acc = seg_ud + K * ND * 64;               // [NP] sums over days of dL/d(physical value)

    float p[NPARAM_MAX];
#pragma unroll
    for (int i = 0; i < NPARAM_MAX; i++) p[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        v = raw ? sigmoid_(v) : v;
        p[i] = descale_(v, s.lo, s.hi);
        acc[i * 64] = 0.0f;
    }

    const auto rx = S2Buf::rsrc(d.x);
    const auto rgf = S2Buf::rsrc(io.grad_flux ? io.grad_flux : io.grad_flux4);
    const auto rg4 = S2Buf::rsrc(io.grad_flux4 ? io.grad_flux4 : io.grad_flux);
    const bool has_gf = io.grad_flux != nullptr, has_g4 = io.grad_flux4 != nullptr;
    const unsigned xvo = (unsigned)(L.b * d.x_b_stride * 4), xts = (unsigned)(d.x_t_stride * 4);
    const unsigned cvo = (unsigned)(L.n * 4), row4 = (unsigned)(N * 4);
    const unsigned gvo = (unsigned)(L.b * 4);
    const unsigned fT = (unsigned)((int64_t)T * B * 4), fB = (unsigned)(B * 4);

    // dynamic rows and their gradient rows: descriptors rebased per day (hbv_stream2.h)
    const float *dbase[NDA];
    float *gdbase[NDA];
    int64_t dts[NDA], gdts[NDA];
    unsigned dvo[NDA], gdvo[NDA];
    float dlo[NDA], dsc[NDA], dsta[NDA];
    bool duse[NDA], dmasked[NDA];
    const int nd = SC >= 3 ? A.nd : ND;                 // SC >= 3: a run-time list of nd <= 3 slots (hbv_stream2.h)
    int dsl[NDA];
    const int cp = d.ch_prcp, ct = d.ch_tmean, ce = d.ch_pet;
    const bool ident = cp == 0 && ct == 1 && ce == 2;   // forcing channels already in (prcp, tmean, pet) order
#pragma unroll
    for (int k = 0; k < ND; k++) {
        const int sl = SC >= 3 ? A.dslot[k < nd ? k : 0] : stream_slot<SC>(k);
        dsl[k] = sl;
        const hbvx_param_src &ps = d.p[sl];
        dbase[k] = ps.dyn ? ps.dyn : d.x;
        dvo[k] = (unsigned)((L.b * ps.dyn_b_stride + L.j) * 4);
        dts[k] = ps.dyn_t_stride;
        dlo[k] = ps.lo;
        dsc[k] = ps.hi - ps.lo;
        dsta[k] = SC >= 3 ? s2_get<NP>(p, sl) : p[stream_slot<SC>(k)];
        duse[k] = !(ps.drop && ps.drop[L.b]);
        dmasked[k] = ps.drop != nullptr;
        const bool dg = io.g[sl].dyn != nullptr;
        gdbase[k] = dg ? io.g[sl].dyn : const_cast<float *>(d.x);
        gdvo[k] = (dg && L.active) ? (unsigned)((L.b * io.g[sl].dyn_b_stride + L.j) * 4) : OOB;
        gdts[k] = dg ? io.g[sl].dyn_t_stride : 0;
    }
    const auto rgx = S2Buf::rsrc(io.grad_x ? io.grad_x : const_cast<float *>(d.x));
    const bool has_gx = io.grad_x != nullptr;
    const unsigned gxvo = leader ? xvo : OOB;

    float a[5];
#pragma unroll
    for (int k = 0; k < 5; k++) a[k] = io.grad_state_out ? io.grad_state_out[k * N + L.n] : 0.0f;

    // phase R item: forcings + raw dynamic values of a day; phase B item: forcings + gradient series of a day
    struct RIn { s2_f3 x; float dv[NDA]; };
    struct BIn { s2_f3 x; float gf[NG], g4[4]; };
    auto issueR = [&](int t, RIn &I) {
        I.x = S2Buf::ld3(rx, xvo, (unsigned)t * xts);      // (the plan admits three adjacent channels only)
#pragma unroll
        for (int k = 0; k < ND; k++)
            if (SC < 3 || k < nd) I.dv[k] = S2Buf::ld(S2Buf::rsrc(dbase[k] + (int64_t)t * dts[k]), dvo[k], 0);
    };
    auto issueB = [&](int t, BIn &I) {
        I.x = S2Buf::ld3(rx, xvo, (unsigned)t * xts);
        const unsigned sg = (unsigned)t * fB;
#pragma unroll
        for (int k = 0; k < NG; k++) {
            // only loads here -- the two sources are summed where the day uses them (hbv_chunked.h::chunk_issue)
            I.gf[k] = (GFULL && has_gf) ? S2Buf::ld(rgf, gvo, sg + (unsigned)k * fT) : 0.0f;
            if (k < 4) I.g4[k] = has_g4 ? S2Buf::ld(rg4, gvo, sg + (unsigned)k * fT) : 0.0f;
        }
    };
    float ck[5];
    auto issueC = [&](int seg) {
        const auto rck = S2Buf::rsrc(io.traj + (int64_t)seg * 5 * N);      // this checkpoint's five rows
#pragma unroll
        for (int k = 0; k < 5; k++) ck[k] = S2Buf::ld(rck, cvo, (unsigned)k * row4);
    };

    auto day = [&](int t, int l, const BIn &I) {
        Step<MODEL, BETAET> s;
        if (ident) { s.P = I.x.x; s.Tf = I.x.y; s.PET = I.x.z; }
        else { s.P = s2_pick(I.x, cp); s.Tf = s2_pick(I.x, ct); s.PET = s2_pick(I.x, ce); }
        const float *row = seg_st + l * 5 * 64;
        s.SP = row[0]; s.MW = row[64]; s.SM = row[128]; s.SUZ = row[192]; s.SLZ = row[256];
        float ud[NDA];
#pragma unroll
        for (int k = 0; k < ND; k++) {
            ud[k] = 0.0f;
            if (SC >= 3 && k >= nd) continue;
            ud[k] = seg_ud[(l * ND + k) * 64];
            const float pv = duse[k] ? ud[k] * dsc[k] + dlo[k] : dsta[k];
            if (SC >= 3) s2_put<NP>(p, dsl[k], pv);
            else p[stream_slot<SC>(k)] = pv;
        }
        s.template fwd<false>(p, nz, ac, elev, 0.0f, 0.0f);
        FluxGrad g;
        auto GF = [&](int k) -> float {
            if (k >= NG) return 0.0f;
            const float v = (k < 4 && has_g4) ? I.gf[k] + I.g4[k < 4 ? k : 0] : I.gf[k];
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
        if constexpr (SC >= 3) {
#pragma unroll
            for (int k = 0; k < ND; k++) {
                if (k >= nd) continue;
                const float gpk = s2_get<NP>(gp, dsl[k]);
                const float gu = gpk * dsc[k];
                const float gr = raw ? gu * (ud[k] * (1.0f - ud[k])) : gu;
                S2Buf::st(S2Buf::rsrc(gdbase[k] + (int64_t)t * gdts[k]), gdvo[k], 0, duse[k] ? gr : 0.0f);
                if (dmasked[k]) s2_put<NP>(gp, dsl[k], duse[k] ? 0.0f : gpk);     // (hbv_stream2.h::k_bwd_stream2)
            }
#pragma unroll
            for (int i = 0; i < NP; i++) acc[i * 64] += gp[i];
        } else {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            bool dyn_slot = false;
            int kd = 0;
#pragma unroll
            for (int k = 0; k < ND; k++)
                if (stream_slot<SC>(k) == i) { dyn_slot = true; kd = k; }
            if (!dyn_slot) {
                acc[i * 64] += gp[i];
            } else {
                const float gu = gp[i] * dsc[kd];
                const float gr = raw ? gu * (ud[kd] * (1.0f - ud[kd])) : gu;
                S2Buf::st(S2Buf::rsrc(gdbase[kd] + (int64_t)t * gdts[kd]), gdvo[kd], 0, duse[kd] ? gr : 0.0f);
                acc[i * 64] += duse[kd] ? 0.0f : gp[i];
            }
        }
        }
        if (has_gx) {
            const float act = L.active ? 1.0f : 0.0f;
            float gs[3] = {gx[0] * act, gx[1] * act, gx[2] * act};
            ens_sum_dpp<3>(gs, lgMp);
            const unsigned so = (unsigned)t * xts;
            S2Buf::st(rgx, gxvo, so + cp * 4, gs[0]);
            S2Buf::st(rgx, gxvo, so + ct * 4, gs[1]);
            S2Buf::st(rgx, gxvo, so + ce * 4, gs[2]);
        }
    };

    const int nseg = (T + K - 1) / K;
    RIn rn;
    BIn bn;
    issueC(nseg - 1);
    issueR((nseg - 1) * K, rn);
    for (int seg = nseg - 1; seg >= 0; seg--) {
        const int t0 = seg * K, t1 = min(T, t0 + K);
        float st[5];
#pragma unroll
        for (int k = 0; k < 5; k++) st[k] = ck[k];
        // R: storages entering every day of the segment -> LDS
#pragma unroll 1
        for (int t = t0; t < t1; t++) {
            const RIn rc = rn;
            if (t + 1 < t1) issueR(t + 1, rn);
            else issueB(t1 - 1, bn);
            const int l = t - t0;
            float *row = seg_st + l * 5 * 64;
#pragma unroll
            for (int k = 0; k < 5; k++) row[k * 64] = st[k];
#pragma unroll
            for (int k = 0; k < ND; k++) {
                if (SC >= 3 && k >= nd) continue;
                const float u = raw ? sigmoid_dyn_(rc.dv[k]) : rc.dv[k];
                seg_ud[(l * ND + k) * 64] = u;
                const float pv = duse[k] ? u * dsc[k] + dlo[k] : dsta[k];
                if (SC >= 3) s2_put<NP>(p, dsl[k], pv);
                else p[stream_slot<SC>(k)] = pv;
            }
            if (t + 1 < t1) {
                Step<MODEL, BETAET> s;
                if (ident) { s.P = rc.x.x; s.Tf = rc.x.y; s.PET = rc.x.z; }
                else { s.P = s2_pick(rc.x, cp); s.Tf = s2_pick(rc.x, ct); s.PET = s2_pick(rc.x, ce); }
                s.SP = st[0]; s.MW = st[1]; s.SM = st[2]; s.SUZ = st[3]; s.SLZ = st[4];
                s.template fwd<false, true>(p, nz, ac, elev, 0.0f, 0.0f);
                st[0] = s.SP3; st[1] = s.MW3; st[2] = s.SM4; st[3] = s.SUZ4; st[4] = s.SLZ2;
            }
        }
        // B: the adjoint over the segment, last day first
#pragma unroll 1
        for (int t = t1 - 1; t >= t0; t--) {
            const BIn bc = bn;
            if (t > t0) {
                issueB(t - 1, bn);
            } else if (seg > 0) {
                issueC(seg - 1);
                issueR((seg - 1) * K, rn);
            }
            day(t, t - t0, bc);
        }
    }
    if (L.active) {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            if (!io.g[i].sta) continue;
            const hbvx_param_src &s = d.p[i];
            if (SC >= 3 && s.dyn && !s.drop) continue;     // a dynamic slot without a mask has no static share
            float gr = acc[i * 64] * (s.hi - s.lo);
            if (raw) {
                const float u = sigmoid_(s.sta[(int64_t)L.b * s.sta_b_stride + L.j]);
                gr = gr * (u * (1.0f - u));
            }
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
