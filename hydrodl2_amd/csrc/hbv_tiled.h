// hbv_tiled.h -- wave-specialised, LDS-tiled recurrence kernels (the fast path).
//
// One workgroup advances 64 lanes (64/Mp basins x Mp members) through all T days:
//
//   wave 0      "stepper": runs the serial recurrence.  It touches only LDS and
//               VALU -- no global loads or stores, hence never waits on vmcnt.
//               (On CDNA loads and stores share the in-order vmcnt counter: a
//               stepper that stored its own outputs would wait for HBM write
//               acknowledgements on every prefetch; measured 4.9 us/step in the
//               one-wave kernel, hbvx.hip k_fwd.)
//   waves 1..NH "helpers": all HBM traffic.  They stage the next tile of Kt days
//               (forcings; dynamic parameters with sigmoid / dy_drop blend /
//               de-scaling already applied; in the adjoint also the saved
//               trajectory and the incoming flux gradients) into LDS, and drain
//               the previous tile (ensemble means -> flux series, trajectory,
//               dynamic-parameter gradients through sigmoid').
//
// Tiles are double-buffered; one raw s_barrier per tile (s_waitcnt lgkmcnt(0) +
// s_barrier, never __syncthreads(): its release fence would drain the helpers'
// outstanding global stores at every tile).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/hbvx.h"
#include "hbv_step.h"
#include "hbv_adj_step.h"

namespace hbvx {

struct TileGeom {
    int Kt;       // days per tile
    int ND;       // dynamic parameter slots
    int NDm;      // ND + 1 if muwts is staged too
    int in_sz;    // floats per input buffer
    int out_sz;   // floats per output buffer
    int off_pin;  // input buffer: physical dynamic parameters [Kt][NDm][64]
    int off_tin;  // input buffer (adjoint): trajectory+aux [Kt][7][64]
    int off_gin;  // input buffer (adjoint): flux gradients [Kt][NF][bpw]
    int off_tout; // output buffer (forward): trajectory+aux [Kt][7][64]
    int off_xout; // output buffer (adjoint): forcing gradients [Kt][3][64]
    int off_mout; // output buffer (adjoint): muwts gradients [Kt][64]
    int lgMp;
};

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int MODEL, bool BETAET>
struct NParamT {
    static constexpr int value = (MODEL == MODEL_HBV10 || MODEL == MODEL_HBVADJ) ? (BETAET ? 13 : 12)
                               : MODEL == MODEL_HBV11P ? 14 : (MODEL == MODEL_HOURLY ? 19 : 16);
};

struct LaneT {
    int lane, jm, bl, b, j;
    bool active;
    int64_t n;
};

__device__ __forceinline__ LaneT lane_t(const hbvx_desc &d, int lgMp)
{
    LaneT L;
    L.lane = threadIdx.x & 63;
    const int Mp = 1 << lgMp;
    L.jm = L.lane & (Mp - 1);
    L.bl = L.lane >> lgMp;
    int b = blockIdx.x * (64 >> lgMp) + L.bl;
    L.active = (b < d.B) && (L.jm < d.M);
    L.b = b < d.B ? b : d.B - 1;
    L.j = L.jm < d.M ? L.jm : d.M - 1;
    L.n = (int64_t)L.b * d.M + L.j;
    return L;
}

// Ensemble sum of one basin whose per-member values sit in LDS at src[0..Mp): LPI = max(Mp/4, 1)
// adjacent lanes cooperate; lane `sub` reads members 4*sub..4*sub+3 with one ds_read_b128 (masked to
// m < M), then an xor butterfly over the LPI lanes.  Fixed summation tree: independent of tile
// alignment and of where the basin sits.  All 64 lanes must call it (cross-lane shuffles).
__device__ __forceinline__ float ens_sum_lds(const float *src, int sub, int M, int lgMp)
{
    float s;
    if (lgMp >= 2) {
        const float4 v = *reinterpret_cast<const float4 *>(src + 4 * sub);
        const int m0 = 4 * sub;
        const float a = ((m0 < M) ? v.x : 0.0f) + ((m0 + 1 < M) ? v.y : 0.0f);
        const float b = ((m0 + 2 < M) ? v.z : 0.0f) + ((m0 + 3 < M) ? v.w : 0.0f);
        s = a + b;
        for (int k = 0; k < lgMp - 2; k++) s += __shfl_xor(s, 1 << k, 64);
    } else {
        s = src[0];
        if (M > 1) s += src[1];
    }
    return s;
}

// Ensemble sum by ONE lane, same fixed tree as ens_sum_lds: quads (x0+x1)+(x2+x3), then pairwise
// over the quads.  `rot` (0..3) only matters for 4 quads (Mp = 16): the lane reads its quads in
// the order rot, rot+1, ... so that the 16-lane groups of one ds_read_b128 spread over all banks
// (rows of a tile are 64 floats apart, i.e. on the same banks).
__device__ __forceinline__ float ens_sum_lane(const float *src, int M, int lgMp, int rot)
{
    if (lgMp < 2) {
        float s = src[0];
        if (M > 1) s += src[1];
        return s;
    }
    const bool full = (M == (1 << lgMp));
    auto quad = [&](int qi) {
        const float4 v = *reinterpret_cast<const float4 *>(src + 4 * qi);
        if (full) return (v.x + v.y) + (v.z + v.w);
        const int m0 = 4 * qi;
        const float a = ((m0 < M) ? v.x : 0.0f) + ((m0 + 1 < M) ? v.y : 0.0f);
        const float b = ((m0 + 2 < M) ? v.z : 0.0f) + ((m0 + 3 < M) ? v.w : 0.0f);
        return a + b;
    };
    if (lgMp == 4) {
        const float t0 = quad(rot & 3), t1 = quad((rot + 1) & 3), t2 = quad((rot + 2) & 3),
                    t3 = quad((rot + 3) & 3);
        const bool odd = rot & 1;  // rot odd: (t3,t0,t1,t2) are quads (0,1,2,3) up to the swap of halves
        return (t0 + (odd ? t3 : t1)) + (t2 + (odd ? t1 : t3));
    }
    const int nq = 1 << (lgMp - 2);
    float q[16];
#pragma unroll
    for (int i = 0; i < 16; i++) q[i] = (i < nq) ? quad(i) : 0.0f;
#pragma unroll
    for (int k = 1; k < 16; k <<= 1)
        if (k < nq) {
#pragma unroll
            for (int i = 0; i < 16; i += 2 * k)
                if (i < nq) q[i] += q[i + k];
        }
    return q[0];
}

// The same sum for full 16-member ensembles (Mp = M = 16: every BASELINE configuration), as straight code: the four
// 16-byte reads leave together and are waited for once.  ens_sum_lane's general form branches per quad on wave-uniform
// flags, which makes the compiler wait for every read before it issues the next (4 x ~120 cycles of LDS latency per
// item: the reducer waves of the pipelined forward were busy 270-315 of its 330-360 cycles per day with that).
// Same quads, same order, same tree: bit-identical.
__device__ __forceinline__ float ens_sum_lane16(const float *src, int rot)
{
    const float4 a = *reinterpret_cast<const float4 *>(src + 4 * (rot & 3));
    const float4 b = *reinterpret_cast<const float4 *>(src + 4 * ((rot + 1) & 3));
    const float4 c = *reinterpret_cast<const float4 *>(src + 4 * ((rot + 2) & 3));
    const float4 e = *reinterpret_cast<const float4 *>(src + 4 * ((rot + 3) & 3));
    const float t0 = (a.x + a.y) + (a.z + a.w), t1 = (b.x + b.y) + (b.z + b.w);
    const float t2 = (c.x + c.y) + (c.z + c.w), t3 = (e.x + e.y) + (e.z + e.w);
    const bool odd = rot & 1;
    return (t0 + (odd ? t3 : t1)) + (t2 + (odd ? t1 : t3));
}

// Ensembles of at most 8 members (Mp = 4 or 8: one or two quads) as straight code -- masks by select, no wave-uniform
// branch around the reads (ens_sum_lane's general form walks sixteen guarded quads: with 16 basins per wave the reducers of
// the pipelined forward have 24 passes per tile and that walk was most of each).  Same quads, same tree: bit-identical.
__device__ __forceinline__ float ens_sum_lane_small(const float *src, int M, int lgMp)
{
    auto quad = [&](int qi) {
        const float4 v = *reinterpret_cast<const float4 *>(src + 4 * qi);
        const int m0 = 4 * qi;
        const float a = ((m0 < M) ? v.x : 0.0f) + ((m0 + 1 < M) ? v.y : 0.0f);
        const float b = ((m0 + 2 < M) ? v.z : 0.0f) + ((m0 + 3 < M) ? v.w : 0.0f);
        return a + b;
    };
    const float q0 = quad(0);
    if (lgMp == 2) return q0;
    return q0 + quad(1);
}

// Reduce `items` (= nt * NFS * bpw) basin-series of an output tile and hand each sum to `emit`.
// Item e -> (bl = e % bpw, ks = (e / bpw) % NFS, tt = e / (bpw * NFS)); values of item e live at
// buf[(tt * NSER + ks) * 64 + bl * Mp ...].  One lane per item, 64 items per wave and pass: per
// item Mp/4 ds_read_b128 and Mp-1 adds, no cross-lane traffic.  `w`/`nw`: this helper wave's
// index / number of waves.
template <int NSER, int NFS, typename Emit>
__device__ __forceinline__ void ens_reduce_pass(const float *buf, int items, int base, int lane, int M,
                                                int lgMp, Emit emit)
{
    const int e = base + lane;
    const bool valid = e < items;
    const int ec = valid ? e : items - 1;
    const int bl = ec & ((64 >> lgMp) - 1);
    const int r = ec >> (6 - lgMp);
    const int ks = r % NFS, tt = r / NFS;
    const float *src = buf + (tt * NSER + ks) * 64 + (bl << lgMp);
    const float v = (lgMp == 4 && M == 16) ? ens_sum_lane16(src, r)
                  : ((lgMp == 2 || lgMp == 3) ? ens_sum_lane_small(src, M, lgMp) : ens_sum_lane(src, M, lgMp, r));
    if (valid) emit(tt, ks, bl, v);
}

template <int NSER, int NFS, typename Emit>
__device__ __forceinline__ void ens_reduce_tile(const float *buf, int items, int lane, int w, int nw,
                                                int M, int lgMp, Emit emit)
{
    for (int base = w * 64; base < items; base += nw * 64)
        ens_reduce_pass<NSER, NFS>(buf, items, base, lane, M, lgMp, emit);
}

// index of dynamic slot i among the dynamic slots (wave-uniform)
__device__ __forceinline__ int dyn_index(unsigned dmask, int i)
{
    return __builtin_popcount(dmask & ((1u << i) - 1u));
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
struct FwdTArgs {
    hbvx_desc d;
    hbvx_fwd_out o;
    TileGeom g;
};

// DYN = false: all parameters static and no muwts -- the specialisation the all-static
// configuration runs; every dynamic-parameter scalar drops out of the stepper loop.
template <int MODEL, bool BETAET, bool DYN>
__global__ void __launch_bounds__(512) k_fwd_tiled(const FwdTArgs A)
{
    constexpr int NP = NParamT<MODEL, BETAET>::value;
    constexpr bool IMPLICIT = (MODEL == MODEL_HBVADJ); // hbv_adj.py: Newton solve per day, flux = Q only
    constexpr int NF = IMPLICIT ? 1 : ((MODEL == MODEL_HBV10) ? 11 : 12);
    extern __shared__ __align__(16) float lds[];
    const hbvx_desc &d = A.d;
    const hbvx_fwd_out &o = A.o;
    const TileGeom &G = A.g;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NH = (blockDim.x >> 6) - 1;
    const LaneT L = lane_t(d, G.lgMp);
    const int lane = L.lane;
    const int T = d.T, Kt = G.Kt;
    const int nT = (T + Kt - 1) / Kt;
    const int64_t N = (int64_t)d.B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const bool has_flux = o.flux != nullptr, has_traj = o.traj != nullptr, has_aux = SAVE_POW && o.aux != nullptr;
    const bool has_mu = DYN && d.muwts != nullptr;

    unsigned dmask = 0;
    if (DYN) {
#pragma unroll
        for (int i = 0; i < NP; i++)
            if (d.p[i].dyn) dmask |= 1u << i;
    }

    float psta[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        v = raw ? sigmoid_(v) : v;
        psta[i] = descale_(v, s.lo, s.hi);
    }

    if (wave == 0) {
        // ------------------------------ stepper ------------------------------
        __builtin_amdgcn_s_setprio(3);
        const float nz = d.nearzero;
        const float ac = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.ac[L.b] : 0.0f;
        const float elev = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.elev[L.b] : 0.0f;
        float p[NPARAM_MAX];
#pragma unroll
        for (int i = 0; i < NPARAM_MAX; i++) p[i] = i < NP ? psta[i < NP ? i : 0] : 0.0f;
        float st[5];
#pragma unroll
        for (int k = 0; k < 5; k++)
            st[k] = d.state_in ? d.state_in[k * N + L.n] : (IMPLICIT ? 0.0f : 0.001f); // hbv_adj.py:254

        lds_barrier(); // tile 0 staged
        for (int kt = 0; kt < nT; kt++) {
            const float *in = lds + (kt & 1) * G.in_sz;
            float *out = lds + 2 * G.in_sz + (kt & 1) * G.out_sz;
            const int nt = min(Kt, T - kt * Kt);
            const float4 *in4 = reinterpret_cast<const float4 *>(in);
            const float *pin = in + G.off_pin;
            // register prefetch of step 0 of the tile
            float4 nf = in4[lane];
            float nd[NP];
            float nmu = 1.0f;
            // Branch-free parameter pick-up: a static slot reads row 0 of the day (always inside the
            // tile) and keeps its value through a select on the wave-uniform mask; the prefetch of
            // day tt+1 runs unconditionally (past the last day it reads rows nobody consumes, still
            // inside this workgroup's LDS).  With 14 dynamic parameters the scalar branches these
            // replace were ~30 per day on the stepper wave.
#pragma unroll
            for (int i = 0; i < NP; i++)
                nd[i] = DYN ? pin[(((dmask >> i) & 1) ? dyn_index(dmask, i) * 64 : 0) + lane] : 0.0f;
            if (has_mu) nmu = pin[G.ND * 64 + lane];
            for (int tt = 0; tt < nt; tt++) {
                const float fP = nf.x, fT = nf.y, fE = nf.z;
                const float wq = nmu;
                if (DYN) {
#pragma unroll
                    for (int i = 0; i < NP; i++) p[i] = ((dmask >> i) & 1) ? nd[i] : p[i];
                }
                { // LDS -> register prefetch of the next day
                    nf = in4[(tt + 1) * 64 + lane];
                    if (DYN) {
                        const float *pr = pin + (tt + 1) * G.NDm * 64 + lane;
#pragma unroll
                        for (int i = 0; i < NP; i++) nd[i] = pr[((dmask >> i) & 1) ? dyn_index(dmask, i) * 64 : 0];
                        if (has_mu) nmu = pr[G.ND * 64];
                    }
                }
                if constexpr (IMPLICIT) {
                    AdjStep<BETAET> s;
                    s.P = fP; s.Tf = fT; s.PET = fE;
                    if (has_traj) {
                        float *to = out + G.off_tout + tt * 7 * 64 + lane;
#pragma unroll
                        for (int k = 0; k < 5; k++) to[k * 64] = st[k];
                    }
                    float xn[5];
                    float Qs = 0.0f;
                    if (d.adj_stop == 2) AdjStaged<BETAET>::day(p, fP, fT, fE, st, d.adj_gtol, d.adj_max_iter, xn, Qs);
                    else adj_newton<BETAET>(s, p, st, 1.0f, d.adj_gtol, d.adj_max_iter, xn, d.adj_stop != 0);
#pragma unroll
                    for (int k = 0; k < 5; k++) st[k] = xn[k];
                    if (has_flux) { // hbv_adj.py:309-313,431: Q at the solved storages
                        const float SUZ = fmaxf(st[3], 0.0f), SLZ = fmaxf(st[4], 0.0f);
                        const float q0 = p[P_K0] * fmaxf(SUZ - p[P_UZL], 0.0f);
                        out[tt * NF * 64 + lane] = (q0 + p[P_K1] * SUZ) + p[P_K2] * SLZ;
                    }
                } else {
                    Step<MODEL, BETAET> s;
                    s.P = fP; s.Tf = fT; s.PET = fE;
                    s.SP = st[0]; s.MW = st[1]; s.SM = st[2]; s.SUZ = st[3]; s.SLZ = st[4];
                    s.template fwd<false, true>(p, nz, ac, elev, 0.f, 0.f);
                    if (has_traj) {
                        float *to = out + G.off_tout + tt * 7 * 64 + lane;
#pragma unroll
                        for (int k = 0; k < 5; k++) to[k * 64] = st[k];
                        if (has_aux) { to[5 * 64] = s.sw0; to[6 * 64] = s.ef0; }
                    }
                    st[0] = s.SP3; st[1] = s.MW3; st[2] = s.SM4; st[3] = s.SUZ4; st[4] = s.SLZ2;
                    if (has_flux) {
                        float *fo = out + tt * NF * 64 + lane;
                        fo[HBVX_F_QSIM * 64] = has_mu ? s.Q * wq : s.Q;
                        if constexpr (NF > 1) {
                            fo[HBVX_F_Q0 * 64] = s.Q0;
                            fo[HBVX_F_Q1 * 64] = s.Q1;
                            fo[HBVX_F_Q2 * 64] = s.Q2;
                            fo[HBVX_F_AET * 64] = s.ET;
                            fo[HBVX_F_SWE * 64] = s.SP3;
                            fo[HBVX_F_RECHARGE * 64] = s.rech;
                            fo[HBVX_F_EXCS * 64] = s.exc;
                            fo[HBVX_F_EVAPFACTOR * 64] = s.ef;
                            fo[HBVX_F_TOSOIL * 64] = s.tosoil;
                            fo[HBVX_F_PERC * 64] = s.PERC;
                            if (NF > HBVX_F_CAPILLARY) fo[HBVX_F_CAPILLARY * 64] = s.cap;
                        }
                    }
                }
            }
            lds_barrier();
        }
        if (L.active) {
#pragma unroll
            for (int k = 0; k < 5; k++) {
                o.state_out[k * N + L.n] = st[k];
                if (has_traj) o.traj[((int64_t)k * (T + 1) + T) * N + L.n] = st[k];
            }
        }
    } else {
        // ------------------------------ helpers ------------------------------
        const int w = wave - 1;
        const int lgMp = G.lgMp, bpw = 64 >> lgMp;
        const int b0 = blockIdx.x * bpw;
        const float invM = 1.0f / (float)d.M;
        const float *xb = d.x + (int64_t)L.b * d.x_b_stride;
        const float *mu = has_mu ? d.muwts + (int64_t)L.b * d.mu_b_stride + L.j : nullptr;
        const float *dynp[NP];
        bool use_dyn[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const hbvx_param_src &s = d.p[i];
            dynp[i] = s.dyn ? s.dyn + (int64_t)L.b * s.dyn_b_stride + L.j : s.sta;
            use_dyn[i] = s.dyn && !(s.drop && s.drop[IMPLICIT ? L.n : (int64_t)L.b]); // hbv_adj.py:182-189
        }

        auto fill = [&](int kt) {
            float *in = lds + (kt & 1) * G.in_sz;
            float4 *in4 = reinterpret_cast<float4 *>(in);
            float *pin = in + G.off_pin;
            const int t0 = kt * Kt, nt = min(Kt, T - t0);
            for (int tt = w; tt < nt; tt += NH) {
                const int t = t0 + tt;
                const float *xr = xb + (int64_t)t * d.x_t_stride;
                float4 f;
                f.x = xr[d.ch_prcp]; f.y = xr[d.ch_tmean]; f.z = xr[d.ch_pet]; f.w = 0.0f;
                float rv[NP];
#pragma unroll
                for (int i = 0; i < NP; i++)
                    rv[i] = ((dmask >> i) & 1) ? dynp[i][(int64_t)t * d.p[i].dyn_t_stride] : 0.0f;
                float muv = has_mu ? mu[(int64_t)t * d.mu_t_stride] : 0.0f;
                in4[tt * 64 + lane] = f;
                float *pr = pin + tt * G.NDm * 64 + lane;
#pragma unroll
                for (int i = 0; i < NP; i++)
                    if ((dmask >> i) & 1) {
                        float v = raw ? sigmoid_dyn_(rv[i]) : rv[i];
                        float pv = descale_(v, d.p[i].lo, d.p[i].hi);
                        pr[dyn_index(dmask, i) * 64] = use_dyn[i] ? pv : psta[i];
                    }
                if (has_mu) pr[G.ND * 64] = muv;
            }
        };

        auto drain = [&](int kt) {
            const float *out = lds + 2 * G.in_sz + (kt & 1) * G.out_sz;
            const int t0 = kt * Kt, nt = min(Kt, T - t0);
            if (has_traj) {
                for (int tt = w; tt < nt; tt += NH) {
                    const int t = t0 + tt;
                    const float *to = out + G.off_tout + tt * 7 * 64 + lane;
                    float v[7];
#pragma unroll
                    for (int k = 0; k < 7; k++) v[k] = (k < 5 || has_aux) ? to[k * 64] : 0.0f;
                    if (L.active) {
#pragma unroll
                        for (int k = 0; k < 5; k++) o.traj[((int64_t)k * (T + 1) + t) * N + L.n] = v[k];
                        if (has_aux) {
                            o.aux[((int64_t)0 * T + t) * N + L.n] = v[5];
                            o.aux[((int64_t)1 * T + t) * N + L.n] = v[6];
                        }
                    }
                }
            }
            if (has_flux) {
                ens_reduce_tile<NF, NF>(out, nt * NF * bpw, lane, w, NH, d.M, lgMp,
                                        [&](int tt, int kk, int bl, float acc) {
                                            if (!(kk == HBVX_F_QSIM && has_mu)) acc = acc * invM;
                                            if (b0 + bl < d.B)
                                                o.flux[((int64_t)kk * T + (t0 + tt)) * d.B + b0 + bl] = acc;
                                        });
            }
        };

        fill(0);
        lds_barrier();
        for (int kt = 0; kt < nT; kt++) {
            if (kt + 1 < nT) fill(kt + 1);
            if (kt > 0) drain(kt - 1);
            lds_barrier();
        }
        if (nT > 0) drain(nT - 1);
    }
}

// ---------------------------------------------------------------------------
// adjoint
// ---------------------------------------------------------------------------
struct BwdTArgs {
    hbvx_desc d;
    hbvx_bwd_io io;
    TileGeom g;
};

// GFULL = false: only the runoff series (Qsim, Q0, Q1, Q2) carry gradient (loss on routed or
// un-routed streamflow): the other flux adjoints are compile-time zero and are neither staged
// nor added.
// Workgroup size the tiled adjoint is compiled for: 8 waves (two per SIMD, 256 VGPRs each), except
// the HBV 2.0 / hourly steps with dynamic parameters, whose stepper wave needs more than 256 registers
// (it spilled 7-31 values): 4 waves, one per SIMD, up to 512 registers -- stepper + three helpers.
template <int MODEL, bool DYN>
constexpr int bwd_tiled_threads()
{
    return ((MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) && DYN) ? 256 : 512;
}

template <int MODEL, bool BETAET, bool DYN, bool GFULL>
__global__ void __launch_bounds__((bwd_tiled_threads<MODEL, DYN>())) k_bwd_tiled(const BwdTArgs A)
{
    constexpr int NP = NParamT<MODEL, BETAET>::value;
    constexpr int NF = (MODEL == MODEL_HBV10) ? 11 : 12;
    extern __shared__ __align__(16) float lds[];
    const hbvx_desc &d = A.d;
    const hbvx_bwd_io &io = A.io;
    const TileGeom &G = A.g;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NH = (blockDim.x >> 6) - 1;
    const LaneT L = lane_t(d, G.lgMp);
    const int lane = L.lane;
    const int T = d.T, Kt = G.Kt;
    const int nT = (T + Kt - 1) / Kt;
    const int64_t N = (int64_t)d.B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    constexpr int NG = GFULL ? NF : 4; // staged gradient series per day
    const bool has_mu = DYN && d.muwts != nullptr;
    const bool has_gx = io.grad_x != nullptr, has_gmu = DYN && io.grad_muwts != nullptr;
    const int lgMp = G.lgMp, bpw = 64 >> lgMp;
    const float invM = 1.0f / (float)d.M;

    unsigned dmask = 0;
    if (DYN) {
#pragma unroll
        for (int i = 0; i < NP; i++)
            if (d.p[i].dyn) dmask |= 1u << i;
    }

    float psta[NP], usta[NP];
    bool use_dyn[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        usta[i] = raw ? sigmoid_(v) : v;
        psta[i] = descale_(usta[i], s.lo, s.hi);
        use_dyn[i] = s.dyn && !(s.drop && s.drop[L.b]);
    }

    if (wave == 0) {
        // ------------------------------ stepper ------------------------------
        __builtin_amdgcn_s_setprio(3);
        const float nz = d.nearzero;
        const float ac = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.ac[L.b] : 0.0f;
        const float elev = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.elev[L.b] : 0.0f;
        float p[NPARAM_MAX], gsta[NP];
#pragma unroll
        for (int i = 0; i < NPARAM_MAX; i++) p[i] = i < NP ? psta[i < NP ? i : 0] : 0.0f;
#pragma unroll
        for (int i = 0; i < NP; i++) gsta[i] = 0.0f;
        float a[5];
#pragma unroll
        for (int k = 0; k < 5; k++) a[k] = io.grad_state_out ? io.grad_state_out[k * N + L.n] : 0.0f;

        lds_barrier();
        for (int it = 0; it < nT; it++) {
            const int kt = nT - 1 - it;
            const float *in = lds + (it & 1) * G.in_sz;
            float *out = lds + 2 * G.in_sz + (it & 1) * G.out_sz;
            const int nt = min(Kt, T - kt * Kt);
            const float4 *in4 = reinterpret_cast<const float4 *>(in);
            const float *pin = in + G.off_pin;
            const float *tin = in + G.off_tin;
            const float *gin = in + G.off_gin;
            for (int tt = nt - 1; tt >= 0; tt--) {
                Step<MODEL, BETAET> s;
                const float4 f = in4[tt * 64 + lane];
                s.P = f.x; s.Tf = f.y; s.PET = f.z;
                const float *pr = pin + tt * G.NDm * 64 + lane;
#pragma unroll
                for (int i = 0; i < NP; i++)
                    if ((dmask >> i) & 1) p[i] = pr[dyn_index(dmask, i) * 64];
                const float *tr = tin + tt * 7 * 64 + lane;
                s.SP = tr[0]; s.MW = tr[64]; s.SM = tr[128]; s.SUZ = tr[192]; s.SLZ = tr[256];
                const float sw0 = SAVE_POW ? tr[320] : 0.0f, ef0 = SAVE_POW ? tr[384] : 0.0f;
                const float *gr = gin + tt * NG * bpw + L.bl;
                FluxGrad g;
                const float gq = gr[HBVX_F_QSIM * bpw];
                const float wq = has_mu ? pr[G.ND * 64] : invM;
                g.gQ = gq * wq;
                g.gQ0 = gr[HBVX_F_Q0 * bpw];
                g.gQ1 = gr[HBVX_F_Q1 * bpw];
                g.gQ2 = gr[HBVX_F_Q2 * bpw];
                g.gET = GFULL ? gr[HBVX_F_AET * bpw] : 0.0f;
                g.gSWE = GFULL ? gr[HBVX_F_SWE * bpw] : 0.0f;
                g.grech = GFULL ? gr[HBVX_F_RECHARGE * bpw] : 0.0f;
                g.gexc = GFULL ? gr[HBVX_F_EXCS * bpw] : 0.0f;
                g.gef = GFULL ? gr[HBVX_F_EVAPFACTOR * bpw] : 0.0f;
                g.gtosoil = GFULL ? gr[HBVX_F_TOSOIL * bpw] : 0.0f;
                g.gPERC = GFULL ? gr[HBVX_F_PERC * bpw] : 0.0f;
                g.gcap = (GFULL && NF > HBVX_F_CAPILLARY) ? gr[(NF - 1) * bpw] : 0.0f;
                s.template fwd<SAVE_POW>(p, nz, ac, elev, sw0, ef0);
                float gp[NPARAM_MAX], gx[3];
#pragma unroll
                for (int i = 0; i < NPARAM_MAX; i++) gp[i] = 0.0f;
                s.bwd(p, nz, g, a, gp, gx);
                float *go = out + tt * G.ND * 64 + lane;
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    const float gu = gp[i] * (d.p[i].hi - d.p[i].lo);
                    if ((dmask >> i) & 1) {
                        go[dyn_index(dmask, i) * 64] = gu;
                        gsta[i] += use_dyn[i] ? 0.0f : gu;
                    } else {
                        gsta[i] += gu;
                    }
                }
                if (has_gx) {
                    float *xo = out + G.off_xout + tt * 3 * 64 + lane;
                    xo[0] = gx[0]; xo[64] = gx[1]; xo[128] = gx[2];
                }
                if (has_gmu) out[G.off_mout + tt * 64 + lane] = gq * s.Q;
            }
            lds_barrier();
        }
        // The static row is row T-1 of the same tensor the helpers write the dynamic rows into: wait
        // until they have drained the last tile (with a single tile that is the one holding T-1, and
        // their zero / dynamic store would race with the += below).
        lds_barrier();
        if (L.active) {
#pragma unroll
            for (int i = 0; i < NP; i++) {
                if (!io.g[i].sta) continue;
                float gr = raw ? gsta[i] * (usta[i] * (1.0f - usta[i])) : gsta[i];
                float *dst = io.g[i].sta + (int64_t)L.b * io.g[i].sta_b_stride + L.j;
                *dst += gr;
            }
            if (io.grad_state_in) {
#pragma unroll
                for (int k = 0; k < 5; k++) io.grad_state_in[k * N + L.n] = a[k];
            }
        }
    } else {
        // ------------------------------ helpers ------------------------------
        const int w = wave - 1;
        const int hid = w * 64 + lane, nhid = NH * 64;
        const int b0 = blockIdx.x * bpw;
        const float *xb = d.x + (int64_t)L.b * d.x_b_stride;
        const float *mu = has_mu ? d.muwts + (int64_t)L.b * d.mu_b_stride + L.j : nullptr;
        const float *dynp[NP];
        float *gdyn[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const hbvx_param_src &s = d.p[i];
            dynp[i] = s.dyn ? s.dyn + (int64_t)L.b * s.dyn_b_stride + L.j : s.sta;
            gdyn[i] = io.g[i].dyn ? io.g[i].dyn + (int64_t)L.b * io.g[i].dyn_b_stride + L.j : nullptr;
        }

        auto fill = [&](int it) {
            const int kt = nT - 1 - it;
            float *in = lds + (it & 1) * G.in_sz;
            float4 *in4 = reinterpret_cast<float4 *>(in);
            float *pin = in + G.off_pin;
            float *tin = in + G.off_tin;
            float *gin = in + G.off_gin;
            const int t0 = kt * Kt, nt = min(Kt, T - t0);
            for (int tt = w; tt < nt; tt += NH) {
                const int t = t0 + tt;
                const float *xr = xb + (int64_t)t * d.x_t_stride;
                float4 f;
                f.x = xr[d.ch_prcp]; f.y = xr[d.ch_tmean]; f.z = xr[d.ch_pet]; f.w = 0.0f;
                float tv[7];
#pragma unroll
                for (int k = 0; k < 5; k++) tv[k] = io.traj[((int64_t)k * (T + 1) + t) * N + L.n];
                if (SAVE_POW) {
                    tv[5] = io.aux[((int64_t)0 * T + t) * N + L.n];
                    tv[6] = io.aux[((int64_t)1 * T + t) * N + L.n];
                } else {
                    tv[5] = tv[6] = 0.0f;
                }
                float rv[NP];
#pragma unroll
                for (int i = 0; i < NP; i++)
                    rv[i] = ((dmask >> i) & 1) ? dynp[i][(int64_t)t * d.p[i].dyn_t_stride] : 0.0f;
                float muv = has_mu ? mu[(int64_t)t * d.mu_t_stride] : 0.0f;
                in4[tt * 64 + lane] = f;
                float *tr = tin + tt * 7 * 64 + lane;
#pragma unroll
                for (int k = 0; k < (SAVE_POW ? 7 : 5); k++) tr[k * 64] = tv[k];
                float *pr = pin + tt * G.NDm * 64 + lane;
#pragma unroll
                for (int i = 0; i < NP; i++)
                    if ((dmask >> i) & 1) {
                        float v = raw ? sigmoid_dyn_(rv[i]) : rv[i];
                        float pv = descale_(v, d.p[i].lo, d.p[i].hi);
                        pr[dyn_index(dmask, i) * 64] = use_dyn[i] ? pv : psta[i];
                    }
                if (has_mu) pr[G.ND * 64] = muv;
            }
            // incoming flux-series gradients, compact [tt][k][bl], pre-scaled by 1/M
            // (mean backward) except Qsim, whose per-lane weight the stepper applies.
            const int items = nt * NG * bpw;
            for (int e = hid; e < items; e += nhid) {
                const int bl = e & (bpw - 1);
                const int r = e >> (6 - lgMp);
                const int kk = r % NG, tt = r / NG;
                const int bb = min(b0 + bl, d.B - 1);
                const int64_t gi = ((int64_t)kk * T + (t0 + tt)) * d.B + bb;
                float v = (GFULL && io.grad_flux) ? io.grad_flux[gi] : 0.0f;
                if (io.grad_flux4 && kk < 4) v += io.grad_flux4[gi];
                gin[(tt * NG + kk) * bpw + bl] = (kk == HBVX_F_QSIM) ? v : v * invM;
            }
        };

        auto drain = [&](int it) {
            const int kt = nT - 1 - it;
            const float *out = lds + 2 * G.in_sz + (it & 1) * G.out_sz;
            const int t0 = kt * Kt, nt = min(Kt, T - t0);
            if (dmask) {
                for (int tt = w; tt < nt; tt += NH) {
                    const int t = t0 + tt;
                    const float *go = out + tt * G.ND * 64 + lane;
                    float rv[NP];
#pragma unroll
                    for (int i = 0; i < NP; i++)
                        rv[i] = (((dmask >> i) & 1) && raw)
                                    ? dynp[i][(int64_t)t * d.p[i].dyn_t_stride] : 0.0f;
#pragma unroll
                    for (int i = 0; i < NP; i++)
                        if (((dmask >> i) & 1) && gdyn[i]) {
                            float gu = go[dyn_index(dmask, i) * 64];
                            if (raw) {
                                float u = sigmoid_dyn_(rv[i]);
                                gu = gu * (u * (1.0f - u));
                            }
                            if (L.active)
                                gdyn[i][(int64_t)t * io.g[i].dyn_t_stride] = use_dyn[i] ? gu : 0.0f;
                        }
                }
            }
            if (has_gx) {
                ens_reduce_tile<3, 3>(out + G.off_xout, nt * 3 * bpw, lane, w, NH, d.M, lgMp,
                                      [&](int tt, int c, int bl, float acc) {
                                          if (b0 + bl < d.B) {
                                              const int ch = c == 0 ? d.ch_prcp : (c == 1 ? d.ch_tmean : d.ch_pet);
                                              io.grad_x[(int64_t)(t0 + tt) * d.x_t_stride +
                                                        (int64_t)(b0 + bl) * d.x_b_stride + ch] = acc;
                                          }
                                      });
            }
            if (has_gmu) {
                for (int tt = w; tt < nt; tt += NH) {
                    float v = out[G.off_mout + tt * 64 + lane];
                    if (L.active) io.grad_muwts[((int64_t)(t0 + tt) * d.B + L.b) * d.M + L.j] = v;
                }
            }
        };

        fill(0);
        lds_barrier();
        for (int it = 0; it < nT; it++) {
            if (it + 1 < nT) fill(it + 1);
            if (it > 0) drain(it - 1);
            lds_barrier();
        }
        if (nT > 0) drain(nT - 1);
        // stores of the last drain have left this wave before the stepper adds the static gradient
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
    }
}

} // namespace hbvx
