// hbv_chunked.h -- time-parallel adjoint of the explicit HBV recurrence.
//
// The forward recurrence is nonlinear and strictly serial in time, but its adjoint
//     a_t = J_t^T a_{t+1} + c_t        (a_t = dL/d(storages entering day t); c_t from dL/d(flux_t))
// is LINEAR in a, and J_t, c_t depend only on the saved trajectory.  With the trajectory in HBM
// (hbvx_fwd_out.traj) the 7300-day chain therefore splits into independent chunks:
//
//   B1 k_bwd_chunk_phi    per (64 lanes, chunk of C days), all chunks in parallel: sweep the chunk
//                         backwards propagating the 5 unit adjoints and the offset, i.e. build
//                         a_{t0} = Phi a_{t1} + phi  (Phi 5x5, phi 5) for the chunk.
//   B2 k_bwd_chunk_scan   per lane: hop the adjoint across chunk boundaries, last chunk first
//                         (T/C steps of a 5x5 mat-vec); record the adjoint entering every chunk.
//   B3 k_bwd_chunk_sweep  per (64 lanes, chunk), all chunks in parallel: the ordinary adjoint
//                         sweep seeded with the chunk's true incoming adjoint: dynamic-parameter,
//                         forcing and muwts gradients are written, static-parameter gradients are
//                         summed per chunk.
//   B4 k_bwd_chunk_reduce per (parameter, lane): fixed-order sum of the chunk partials (deterministic),
//                         sigmoid' chain, accumulate into the static-row gradient.
//
// ~19 000 independent waves at cfg2 instead of 168 serial ones: the adjoint becomes throughput/HBM
// bound.  B1 and B3 are plain one-wave-per-block kernels with direct global loads -- there is now
// enough thread-level parallelism to hide memory latency without LDS staging.
// Floating point: a_t is obtained through composed maps, so gradients differ from the serial
// sweep by rounding (1e-6 relative); branch predicates come from the same recomputed forward values.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/hbvx.h"
#include "hbv_step.h"

namespace hbvx {

struct ChunkArgs {
    hbvx_desc d;
    hbvx_bwd_io io;
    int lgMp;
    int C;        // days per chunk
    int nchunk;
    float *phi;   // [nchunk][30][N]  Phi (5 propagated unit adjoints x 5) then phi (5)
    float *abnd;  // [nchunk][5][N]   adjoint entering each chunk from the future
    float *gpart; // [nchunk][NP][N]  per-chunk static-parameter gradient (unit space)
    int nd;       // DYN == 1: number of dynamic parameters (<= CHUNK_FEW) and their slots
    int dslot[3];
    int per_xcd;  // basin groups per XCD of the block map below (host: chunk_per_xcd)
};

// XCD-aware block -> (basin group, chunk) map of the two chunk-parallel kernels.  Workgroups are dealt round-robin
// over the 8 XCDs (block i runs on XCD i % 8: observed, not guaranteed -- it only matters for speed), each with its
// own L2.  A wave reads 48 bytes of a day's forcing row and 16 of each gradient-series row (4 basins): with the
// plain (basin group, chunk) grid the 168 waves that share a day's rows are spread over all eight L2s and every XCD
// pulls every line -- counters at config 2 (profiles/r04_pmc_fetch_size.csv): 2.2 GB of the adjoint's 5.5 GB of HBM
// reads were those 137 MB of basin-shared rows fetched 8 times per pass.  Here XCD c owns the contiguous run of basin
// groups [c * per, (c + 1) * per) for every chunk, walking basin groups first, so the waves of an XCD that are in
// flight together share days AND neighbouring basins.  Grid = 8 * per * nchunk blocks, 1-D; groups past the end
// return at once.
struct ChunkBlock {
    int bx, chunk;
};
__host__ __device__ inline int chunk_per_xcd(int B, int lgMp)
{
    const int bpw = 64 >> lgMp;
    return ((B + bpw - 1) / bpw + 7) / 8;
}
__device__ __forceinline__ bool chunk_block(const hbvx_desc &d, int lgMp, int per_xcd, ChunkBlock &b)
{
    const int i = blockIdx.x;
#if defined(HBVX_CHUNK_XCD) && HBVX_CHUNK_XCD == 0      // A/B builds: the plain map (consecutive blocks = consecutive basin groups)
    b.chunk = i / (8 * per_xcd);
    b.bx = i - b.chunk * (8 * per_xcd);
#else
    const int c = i & 7, k = i >> 3;
    b.chunk = k / per_xcd;
    b.bx = c * per_xcd + (k - b.chunk * per_xcd);
#endif
    return b.bx * (64 >> lgMp) < d.B;
}

// DYN template values: 0 all static; 1 "few": at most CHUNK_FEW dynamic parameters, no muwts -- only
// those rows are loaded / de-scaled / stored, addressed through the wave-uniform slot list (the
// generic mode walks all NP slots every day and needs 230+ VGPRs); 2 generic; 3 "all": every
// parameter dynamic, no dy_drop mask, no muwts (config 3) -- no static values, selects or static
// gradient accumulators at all, and the B4 reduction is skipped (the static gradient is zero).
#define CHUNK_FEW 3

// p[slot] = v / return p[slot] for a wave-uniform runtime slot without dynamic register indexing
template <int NP>
__device__ __forceinline__ void slot_set(float *p, int slot, float v)
{
#pragma unroll
    for (int i = 0; i < NP; i++)
        if (slot == i) p[i] = v;
}
template <int NP>
__device__ __forceinline__ float slot_get(const float *p, int slot)
{
    float v = 0.0f;
#pragma unroll
    for (int i = 0; i < NP; i++)
        if (slot == i) v = p[i];
    return v;
}

struct ChunkLane {
    int jm, b, j;
    bool active, leader;
    int64_t n;
};

__device__ __forceinline__ ChunkLane chunk_lane(const hbvx_desc &d, int lgMp, int bx)
{
    ChunkLane L;
    const int lane = threadIdx.x & 63;
    const int Mp = 1 << lgMp;
    L.jm = lane & (Mp - 1);
    int b = bx * (64 >> lgMp) + (lane >> lgMp);
    L.active = (b < d.B) && (L.jm < d.M);
    L.leader = (b < d.B) && (L.jm == 0);
    L.b = b < d.B ? b : d.B - 1;
    L.j = L.jm < d.M ? L.jm : d.M - 1;
    L.n = (int64_t)L.b * d.M + L.j;
    return L;
}

template <int MODEL, bool BETAET>
struct ChunkNP {
    static constexpr int value = MODEL == MODEL_HBV10 ? (BETAET ? 13 : 12)
                               : MODEL == MODEL_HBV11P ? 14 : (MODEL == MODEL_HOURLY ? 19 : 16);
};

// Raw per-day loads (issued one day ahead), then the recompute that turns them into a Step.
template <int NP>
struct ChunkRaw {
    float f[3];      // P, T, PET
    float st[5];     // storages entering the day
    float sw0, ef0;  // saved pow results (HBVX_SAVE_POW builds only)
    float gf[HBVX_MAX_FLUX];
    float g4[4];     // the routing adjoint's share of the four runoff gradients (grad_flux4), added by chunk_finish
    float dv[NP];    // raw dynamic-parameter values
    float mu;
};

template <int MODEL, bool BETAET, int NP>
struct ChunkDay {
    Step<MODEL, BETAET> s;
    float p[NPARAM_MAX];
    float ud[NP];   // unit value used (for sigmoid')
    FluxGrad g;
    float gq;       // raw dL/dQsim (before the member weight), for the muwts gradient
};

// DYN == 3 with LDSDV: the NP raw dynamic values of the next day travel HBM -> LDS by LDS-DMA
// (chunk_dma_dyn) instead of waiting in NP registers for a whole day; chunk_pull_dyn moves them into
// registers at the start of their day, where the de-scaling consumes them at once.  14-19 registers
// less at the kernels' pressure peak: config 3's sweep drops under the 168 of three waves per SIMD.
// DYN == 3 means more than "every parameter dynamic": all NP rows live in ONE tensor with the reference's column
// order (column i * M + j, hbv.py:201-208; the host checks it and falls back to the generic mode otherwise), so a
// day's values of a lane are NP words M apart behind one address.  (With a pointer and two 64-bit strides per
// parameter the kernels needed 14 x 6 scalar registers for addressing alone: 56 lane-spill reads, 70 scalar
// multiplications and their hazard no-ops per day in config 3's sweep, a third of its instruction stream.)
template <int NP>
__device__ __forceinline__ void chunk_dma_dyn(const hbvx_desc &d, const ChunkLane &L, int t, float *lds)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const hbvx_param_src &ps = d.p[0];
    const float *src = ps.dyn + (int64_t)t * ps.dyn_t_stride + (int64_t)L.b * ps.dyn_b_stride + L.j;
    const int M = d.M;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds + i * 64), 4, 0, 0);
        src += M;
    }
#endif
}
template <int NP>
__device__ __forceinline__ void chunk_pull_dyn(ChunkRaw<NP> &R, const float *lds)
{
#pragma unroll
    for (int i = 0; i < NP; i++) R.dv[i] = lds[i * 64 + (threadIdx.x & 63)];
}

// MU: the call carries learned ensemble weights (d.muwts) -- a template flag of the static and slot-list modes
// (as a run-time flag it cost those instances 3-12 registers and the delta-MG default its fourth wave per SIMD); the
// generic mode (DYN == 2) tests the pointer at run time, the "all" mode never has them.
template <int NP, int DYN, bool GFULL, bool LDSDV = false, bool MU = false>
__device__ __forceinline__ void chunk_issue(const hbvx_desc &d, const hbvx_bwd_io &io,
                                            const ChunkLane &L, int t, int nf, ChunkRaw<NP> &R,
                                            int nd, const int *dslot)
{
    const int T = d.T;
    const int64_t N = (int64_t)d.B * d.M;
    const float *xr = d.x + (int64_t)t * d.x_t_stride + (int64_t)L.b * d.x_b_stride;
    R.f[0] = xr[d.ch_prcp]; R.f[1] = xr[d.ch_tmean]; R.f[2] = xr[d.ch_pet];
    const float *tp = io.traj + (int64_t)t * N + L.n;
    const int64_t ks = (int64_t)(T + 1) * N;
#pragma unroll
    for (int k = 0; k < 5; k++) R.st[k] = tp[k * ks];
    if (SAVE_POW) {
        const float *ap = io.aux + (int64_t)t * N + L.n;
        R.sw0 = ap[0];
        R.ef0 = ap[(int64_t)T * N];
    } else {
        R.sw0 = R.ef0 = 0.0f;   // recomputed by chunk_finish (HBVX_SAVE_POW, hbv_step.h)
    }
    const int64_t fs = (int64_t)T * d.B, go = (int64_t)t * d.B + L.b;
    // loads only, no arithmetic on what they return: an add here (gf + g4) made the compiler wait for every load of
    // the day at the spot where they had just been issued -- a full memory round trip per day instead of a prefetch
    // that flies while the previous day is computed (profiles/r04_ab_chunk_prefetch.txt)
#pragma unroll
    for (int k = 0; k < HBVX_MAX_FLUX; k++) {
        R.gf[k] = 0.0f;
        if (GFULL && k < nf) R.gf[k] = io.grad_flux[k * fs + go];
        if (k < 4) R.g4[k] = io.grad_flux4 ? io.grad_flux4[k * fs + go] : 0.0f;
    }
    R.mu = 0.0f;
    if (DYN == 1) {
#pragma unroll
        for (int k = 0; k < CHUNK_FEW; k++)
            if (k < nd) {
                const hbvx_param_src &ps = d.p[dslot[k]];
                R.dv[k] = ps.dyn[(int64_t)t * ps.dyn_t_stride + (int64_t)L.b * ps.dyn_b_stride + L.j];
            }
    }
    if (DYN == 3 && !LDSDV) {
        const hbvx_param_src &ps = d.p[0];     // one tensor, column i * M + j (see chunk_dma_dyn)
        const float *src = ps.dyn + (int64_t)t * ps.dyn_t_stride + (int64_t)L.b * ps.dyn_b_stride + L.j;
#pragma unroll
        for (int i = 0; i < NP; i++) R.dv[i] = src[i * d.M];
    }
    if (DYN == 2) {
        // branch-free: a static slot re-reads its static value (never used: use_dyn is false)
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const hbvx_param_src &ps = d.p[i];
            const bool dy = ps.dyn != nullptr;
            const float *base = dy ? ps.dyn : ps.sta;
            const int64_t ts = dy ? ps.dyn_t_stride : 0, bs = dy ? ps.dyn_b_stride : ps.sta_b_stride;
            R.dv[i] = base[(int64_t)t * ts + (int64_t)L.b * bs + L.j];
        }
    }
    // learned ensemble weights: in every mode but "all" (round 5: with static or few dynamic parameters they used to
    // force the generic instances -- 2.55 ms against 1.31 for config 2's adjoint)
    if ((MU || DYN == 2) && d.muwts) R.mu = d.muwts[(int64_t)t * d.mu_t_stride + (int64_t)L.b * d.mu_b_stride + L.j];
}

template <int MODEL, bool BETAET, int NP, int DYN, bool GFULL, bool MU = false>
__device__ __forceinline__ void chunk_finish(const hbvx_desc &d, const ChunkRaw<NP> &R, bool raw, float nz,
                                             float ac, float elev, const float *usta, const float *psta,
                                             const bool *use_dyn, int nf, float invM,
                                             ChunkDay<MODEL, BETAET, NP> &D, int nd, const int *dslot, bool has4)
{
    float gf[HBVX_MAX_FLUX];   // (gradient of the flux series or 0) + the routing adjoint's share, as one sum
#pragma unroll
    for (int k = 0; k < HBVX_MAX_FLUX; k++) {
        gf[k] = R.gf[k];
        if (k < 4 && has4) gf[k] += R.g4[k];
    }
    D.s.P = R.f[0]; D.s.Tf = R.f[1]; D.s.PET = R.f[2];
    D.s.SP = R.st[0]; D.s.MW = R.st[1]; D.s.SM = R.st[2]; D.s.SUZ = R.st[3]; D.s.SLZ = R.st[4];
    if (DYN == 1) {
        // use_dyn[k] / D.ud[k] are indexed by the position k in the slot list
#pragma unroll
        for (int i = 0; i < NP; i++) D.p[i] = psta[i];
#pragma unroll
        for (int k = 0; k < CHUNK_FEW; k++)
            if (k < nd) {
                const hbvx_param_src &ps = d.p[dslot[k]];
                const float v = raw ? sigmoid_dyn_(R.dv[k]) : R.dv[k];
                D.ud[k] = v;
                const float pv = descale_(v, ps.lo, ps.hi);
                const float cur = slot_get<NP>(D.p, dslot[k]);
                slot_set<NP>(D.p, dslot[k], use_dyn[k] ? pv : cur);
            }
    }
    if (DYN == 3) {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            D.ud[i] = raw ? sigmoid_dyn_(R.dv[i]) : R.dv[i];
            D.p[i] = descale_(D.ud[i], d.p[i].lo, d.p[i].hi);
        }
    }
#pragma unroll
    for (int i = 0; i < NP; i++) {
        if (DYN == 1 || DYN == 3) break;
        D.ud[i] = usta[i];
        D.p[i] = psta[i];
        if (DYN == 2) {   // selects on the per-lane flag (false for static slots), no branches
            const float v = raw ? sigmoid_dyn_(R.dv[i]) : R.dv[i];
            D.ud[i] = use_dyn[i] ? v : usta[i];
            D.p[i] = use_dyn[i] ? descale_(v, d.p[i].lo, d.p[i].hi) : psta[i];
        }
    }
#pragma unroll
    for (int i = NP; i < NPARAM_MAX; i++) D.p[i] = 0.0f;
    D.s.template fwd<SAVE_POW>(D.p, nz, ac, elev, R.sw0, R.ef0);
    D.gq = gf[HBVX_F_QSIM];
    const float wq = ((MU || DYN == 2) && d.muwts) ? R.mu : invM;
    D.g.gQ = D.gq * wq;
    D.g.gQ0 = gf[HBVX_F_Q0] * invM;
    D.g.gQ1 = gf[HBVX_F_Q1] * invM;
    D.g.gQ2 = gf[HBVX_F_Q2] * invM;
    D.g.gET = GFULL ? gf[HBVX_F_AET] * invM : 0.0f;
    D.g.gSWE = GFULL ? gf[HBVX_F_SWE] * invM : 0.0f;
    D.g.grech = GFULL ? gf[HBVX_F_RECHARGE] * invM : 0.0f;
    D.g.gexc = GFULL ? gf[HBVX_F_EXCS] * invM : 0.0f;
    D.g.gef = GFULL ? gf[HBVX_F_EVAPFACTOR] * invM : 0.0f;
    D.g.gtosoil = GFULL ? gf[HBVX_F_TOSOIL] * invM : 0.0f;
    D.g.gPERC = GFULL ? gf[HBVX_F_PERC] * invM : 0.0f;
    D.g.gcap = (GFULL && nf > HBVX_F_CAPILLARY) ? gf[HBVX_F_CAPILLARY] * invM : 0.0f;
}

template <int NP, int DYN>
__device__ __forceinline__ void chunk_static(const hbvx_desc &d, const ChunkLane &L, bool raw,
                                             float *usta, float *psta, bool *use_dyn, int nd,
                                             const int *dslot)
{
#pragma unroll
    for (int i = 0; i < NP; i++) {
        if (DYN == 3) {   // never read
            usta[i] = psta[i] = 0.0f;
            use_dyn[i] = true;
            continue;
        }
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        usta[i] = raw ? sigmoid_(v) : v;
        psta[i] = descale_(usta[i], s.lo, s.hi);
        use_dyn[i] = DYN == 2 && s.dyn && !(s.drop && s.drop[L.b]);
    }
    if (DYN == 1) {
#pragma unroll
        for (int k = 0; k < CHUNK_FEW; k++) {
            use_dyn[k] = false;
            if (k < nd) {
                const hbvx_param_src &s = d.p[dslot[k]];
                use_dyn[k] = !(s.drop && s.drop[L.b]);
            }
        }
    }
}

// ---- B1 -------------------------------------------------------------------------------------
// SC: compile-time slot list of the "few" mode for the configurations users actually run
// (0 = runtime list; 1 = {parBETA, parBETAET}, the delta-MG default; 2 = {parBETA, parK0, parBETAET}):
// with constant slots the slot_get / slot_set chains fold into plain register accesses.
template <int SC>
struct SlotCombo {
    static constexpr int nd = SC == 1 ? 2 : (SC == 2 ? 3 : 0);
    static constexpr int s0 = P_BETA, s1 = SC == 1 ? P_BETAET : P_K0, s2 = P_BETAET;
};
#define CHUNK_SLOTS(A)                                                                             \
    const int nd_ = SC ? SlotCombo<SC>::nd : (A).nd;                                               \
    const int ds_[3] = {SC ? SlotCombo<SC>::s0 : (A).dslot[0], SC ? SlotCombo<SC>::s1 : (A).dslot[1], \
                        SC ? SlotCombo<SC>::s2 : (A).dslot[2]}

template <int MODEL, bool BETAET, int DYN, bool GFULL, int SC = 0, bool MU = false>
__global__ void __launch_bounds__(64) k_bwd_chunk_phi(const ChunkArgs A)
{
    CHUNK_SLOTS(A);
    constexpr int NP = ChunkNP<MODEL, BETAET>::value;
    const hbvx_desc &d = A.d;
    const hbvx_bwd_io &io = A.io;
    ChunkBlock blk;
    if (!chunk_block(d, A.lgMp, A.per_xcd, blk)) return;
    const ChunkLane L = chunk_lane(d, A.lgMp, blk.bx);
    const int chunk = blk.chunk;
    const int t0 = chunk * A.C, t1 = min(d.T, t0 + A.C);
    const int64_t N = (int64_t)d.B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const float nz = d.nearzero, invM = 1.0f / (float)d.M;
    const float ac = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.ac[L.b] : 0.0f;
    const float elev = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.elev[L.b] : 0.0f;
    float usta[NP], psta[NP];
    bool use_dyn[NP];
    chunk_static<NP, DYN>(d, L, raw, usta, psta, use_dyn, nd_, ds_);

    float Phi[5][5], phi[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        phi[k] = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; i++) Phi[k][i] = (i == k) ? 1.0f : 0.0f;
    }
    FluxGrad g0;
    g0.gQ = g0.gQ0 = g0.gQ1 = g0.gQ2 = g0.gET = g0.gSWE = g0.grech = g0.gexc = g0.gef = g0.gtosoil =
        g0.gPERC = g0.gcap = 0.0f;
    constexpr bool LDSDV = DYN == 3;
    __shared__ float dvbuf[LDSDV ? NP * 64 : 1];
    ChunkRaw<NP> Rn;
    if (LDSDV) chunk_dma_dyn<NP>(d, L, t1 - 1, dvbuf);   // before the day's other loads: see the loop
    chunk_issue<NP, DYN, GFULL, LDSDV, MU>(d, io, L, t1 - 1, io.n_flux, Rn, nd_, ds_);
    for (int t = t1 - 1; t >= t0; t--) {
        ChunkRaw<NP> Rc = Rn;
        if (LDSDV) {
            // EXPLICIT order, DMA -> LDS read: every vector-memory operation in flight here belongs to day t (its rows'
            // DMA and the loads of chunk_issue behind it), all of it is consumed below, so waiting for all of it costs
            // nothing -- and no longer rests on where the compiler happens to wait for the copy above (round 4: a build
            // without that copy read stale rows at config 3's full size).  tests/test_code_object.py checks the ISA.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            chunk_pull_dyn<NP>(Rc, dvbuf);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS reads done before the DMA rewrites the rows
        }
        if (t > t0) {
            // The DMA goes first (oldest of next day's vector-memory operations); next day's vmcnt(0) above covers it.
            if (LDSDV) chunk_dma_dyn<NP>(d, L, t - 1, dvbuf);
            chunk_issue<NP, DYN, GFULL, LDSDV, MU>(d, io, L, t - 1, io.n_flux, Rn, nd_, ds_); // next day's loads in flight
        }
        ChunkDay<MODEL, BETAET, NP> D;
        chunk_finish<MODEL, BETAET, NP, DYN, GFULL, MU>(d, Rc, raw, nz, ac, elev, usta, psta, use_dyn,
                                                    io.n_flux, invM, D, nd_, ds_, io.grad_flux4 != nullptr);
        float gp[NPARAM_MAX], gx[3];
#pragma unroll
        for (int i = 0; i < NPARAM_MAX; i++) gp[i] = 0.0f;
        if constexpr (MODEL == MODEL_HBV10) {
            // feed-forward day: block-triangular J^T, coefficients once, sparse application
            typedef Step<MODEL, BETAET> S;
            const typename S::JT c = D.s.jt_coef(D.p, nz);
            S::template jt_unit<0>(c, Phi[0]);
            S::template jt_unit<0>(c, Phi[1]);
            S::template jt_unit<1>(c, Phi[2]);
            S::template jt_unit<2>(c, Phi[3]);
            S::template jt_unit<2>(c, Phi[4]);
            if constexpr (!GFULL) {
                // loss on the runoff series only: the offset vector takes the same sparse map plus
                // the three source terms instead of a full adjoint step
                S::jt_affine(c, phi, D.g.gQ0 + D.g.gQ, D.g.gQ1 + D.g.gQ, D.g.gQ2 + D.g.gQ);
                continue;
            }
        } else if constexpr (MODEL == MODEL_HBV11P || MODEL == MODEL_HBV20) {
            // capillary rise couples soil and groundwater, but the snow block still stands alone:
            // the SNOWPACK / MELTWATER unit adjoints never leave it
            // the other three rows couple through the capillary exchange but stay sparse (Step::jt_cap):
            // coefficients once per day, three applications instead of three full adjoint steps
            typedef Step<MODEL, BETAET> S;
            const typename S::JTC c = D.s.jt_coef_cap(D.p, nz);
            S::template jt_unit<0>(c, Phi[0]);
            S::template jt_unit<0>(c, Phi[1]);
#pragma unroll
            for (int k = 2; k < 5; k++) S::jt_cap(c, Phi[k], 0.0f, 0.0f, 0.0f);
            if constexpr (!GFULL) {
                S::jt_cap(c, phi, D.g.gQ0 + D.g.gQ, D.g.gQ1 + D.g.gQ, D.g.gQ2 + D.g.gQ);
                continue;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) D.s.bwd(D.p, nz, g0, Phi[k], gp, gx);
        }
        D.s.bwd(D.p, nz, D.g, phi, gp, gx);
    }
    if (L.active) {
        float *dst = A.phi + ((int64_t)chunk * 30) * N + L.n;
#pragma unroll
        for (int k = 0; k < 5; k++)
#pragma unroll
            for (int i = 0; i < 5; i++) dst[(int64_t)(k * 5 + i) * N] = Phi[k][i];
#pragma unroll
        for (int i = 0; i < 5; i++) dst[(int64_t)(25 + i) * N] = phi[i];
    }
}

#ifndef HBVX_CHUNK_NO_SHARED_KERNELS   // non-template kernels: defined in launch_chunked.hip only
// ---- B2 -------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_bwd_chunk_scan(const ChunkArgs A)
{
    const hbvx_desc &d = A.d;
    const int64_t N = (int64_t)d.B * d.M;
    const int64_t n = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    float a[5];
#pragma unroll
    for (int k = 0; k < 5; k++) a[k] = A.io.grad_state_out ? A.io.grad_state_out[k * N + n] : 0.0f;
    // Register ring, SCAN_DEPTH chunks deep: a lane's hop across a chunk is 30 multiply-adds behind 30 loads, and with
    // one chunk in flight every hop waited a memory round trip (T / C = 115 of them at config 2: 0.1-0.2 ms of a 1.3 ms
    // adjoint call for 168 waves).  The loop is unrolled by the depth so that every buffer is a fixed set of registers.
    constexpr int SCAN_DEPTH = 4;
    float buf[SCAN_DEPTH][30];
    auto fetch = [&](int c, float *dst) __attribute__((always_inline)) {
        const float *src = A.phi + ((int64_t)c * 30) * N + n;
#pragma unroll
        for (int i = 0; i < 30; i++) dst[i] = src[(int64_t)i * N];
    };
#pragma unroll
    for (int j = 0; j < SCAN_DEPTH; j++)
        if (A.nchunk - 1 - j >= 0) fetch(A.nchunk - 1 - j, buf[j]);
    for (int c0 = A.nchunk - 1; c0 >= 0; c0 -= SCAN_DEPTH) {
#pragma unroll
        for (int j = 0; j < SCAN_DEPTH; j++) {
            const int c = c0 - j;
            if (c < 0) break;
            float cur[30];
#pragma unroll
            for (int i = 0; i < 30; i++) cur[i] = buf[j][i];
            if (c - SCAN_DEPTH >= 0) fetch(c - SCAN_DEPTH, buf[j]);   // its registers are free again
#pragma unroll
            for (int k = 0; k < 5; k++) A.abnd[((int64_t)c * 5 + k) * N + n] = a[k];
            float an[5];
#pragma unroll
            for (int i = 0; i < 5; i++) {
                float v = cur[25 + i];
#pragma unroll
                for (int k = 0; k < 5; k++) v += a[k] * cur[k * 5 + i];
                an[i] = v;
            }
#pragma unroll
            for (int i = 0; i < 5; i++) a[i] = an[i];
        }
    }
    if (A.io.grad_state_in) {
#pragma unroll
        for (int k = 0; k < 5; k++) A.io.grad_state_in[k * N + n] = a[k];
    }
}

#endif

// ---- B3 -------------------------------------------------------------------------------------
__device__ __forceinline__ float chunk_ens_sum(float v, int lgMp)
{
    for (int s = 0; s < lgMp; s++) v += __shfl_xor(v, 1 << s, 64);
    return v;
}

// ROWST (DYN == 3 only): the NP gradients of a lane-day are transposed through LDS and leave as whole
// rows -- a wave's (64/Mp) basins x NP x M floats are contiguous in the [T,B,ny] gradient tensor, so
// 8-byte stores of 512 contiguous bytes per instruction replace NP four-byte stores that each touch
// (64/Mp) separate 64-byte pieces of misaligned 904-byte rows (partial-line writes: profiles/
// r02_pmc_calibration.csv shows what those cost).  The host sets it when every parameter's gradient
// lives in one tensor with the reference's column order (column = i*M + j, hbv.py:201-208).
template <int MODEL, bool BETAET, int DYN, bool GFULL, int SC = 0, bool ROWST = false, bool MU = false>
__global__ void __launch_bounds__(64) k_bwd_chunk_sweep(const ChunkArgs A)
{
    CHUNK_SLOTS(A);
    constexpr int NP = ChunkNP<MODEL, BETAET>::value;
    __shared__ float rowbuf[ROWST ? 64 * NP : 1];
    const hbvx_desc &d = A.d;
    const hbvx_bwd_io &io = A.io;
    ChunkBlock blk;
    if (!chunk_block(d, A.lgMp, A.per_xcd, blk)) return;
    const ChunkLane L = chunk_lane(d, A.lgMp, blk.bx);
    const int chunk = blk.chunk;
    const int t0 = chunk * A.C, t1 = min(d.T, t0 + A.C);
    const int64_t N = (int64_t)d.B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const float nz = d.nearzero, invM = 1.0f / (float)d.M;
    const float ac = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.ac[L.b] : 0.0f;
    const float elev = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.elev[L.b] : 0.0f;
    float usta[NP], psta[NP], gsta[NP];
    bool use_dyn[NP];
    chunk_static<NP, DYN>(d, L, raw, usta, psta, use_dyn, nd_, ds_);
#pragma unroll
    for (int i = 0; i < NP; i++) gsta[i] = 0.0f;
    float gused[CHUNK_FEW] = {0.0f, 0.0f, 0.0f};
    float a[5];
#pragma unroll
    for (int k = 0; k < 5; k++) a[k] = A.abnd[((int64_t)chunk * 5 + k) * N + L.n];

    constexpr bool LDSDV = DYN == 3;
    __shared__ float dvbuf[LDSDV ? NP * 64 : 1];
    ChunkRaw<NP> Rn;
    if (LDSDV) chunk_dma_dyn<NP>(d, L, t1 - 1, dvbuf);   // before the day's other loads: see the loop
    chunk_issue<NP, DYN, GFULL, LDSDV, MU>(d, io, L, t1 - 1, io.n_flux, Rn, nd_, ds_);
    for (int t = t1 - 1; t >= t0; t--) {
        ChunkRaw<NP> Rc = Rn;
        if (LDSDV) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the rows' DMA has landed (k_bwd_chunk_phi)
            chunk_pull_dyn<NP>(Rc, dvbuf);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (t > t0) {
            if (LDSDV) chunk_dma_dyn<NP>(d, L, t - 1, dvbuf);
            chunk_issue<NP, DYN, GFULL, LDSDV, MU>(d, io, L, t - 1, io.n_flux, Rn, nd_, ds_);
        }
        ChunkDay<MODEL, BETAET, NP> D;
        chunk_finish<MODEL, BETAET, NP, DYN, GFULL, MU>(d, Rc, raw, nz, ac, elev, usta, psta, use_dyn,
                                                    io.n_flux, invM, D, nd_, ds_, io.grad_flux4 != nullptr);
        if ((MU || DYN == 2) && io.grad_muwts && L.active) io.grad_muwts[((int64_t)t * d.B + L.b) * d.M + L.j] = D.gq * D.s.Q;
        float gp[NPARAM_MAX], gx[3];
#pragma unroll
        for (int i = 0; i < NPARAM_MAX; i++) gp[i] = 0.0f;
        D.s.bwd(D.p, nz, D.g, a, gp, gx);
        if (DYN == 1) {
#pragma unroll
            for (int i = 0; i < NP; i++) gsta[i] += gp[i] * (d.p[i].hi - d.p[i].lo);
#pragma unroll
            for (int k = 0; k < CHUNK_FEW; k++)
                if (k < nd_) {
                    const int sl = ds_[k];
                    const float gu = slot_get<NP>(gp, sl) * (d.p[sl].hi - d.p[sl].lo);
                    const float gr = raw ? gu * (D.ud[k] * (1.0f - D.ud[k])) : gu;
                    if (io.g[sl].dyn && L.active)
                        io.g[sl].dyn[(int64_t)t * io.g[sl].dyn_t_stride + (int64_t)L.b * io.g[sl].dyn_b_stride + L.j] =
                            use_dyn[k] ? gr : 0.0f;
                    gused[k] += use_dyn[k] ? gu : 0.0f;   // goes to the dynamic rows, not to the static one
                }
        }
        if (DYN == 3 && !ROWST) {
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const float gu = gp[i] * (d.p[i].hi - d.p[i].lo);
                const float gr = raw ? gu * (D.ud[i] * (1.0f - D.ud[i])) : gu;
                if (io.g[i].dyn && L.active)
                    io.g[i].dyn[(int64_t)t * io.g[i].dyn_t_stride + (int64_t)L.b * io.g[i].dyn_b_stride + L.j] = gr;
            }
        }
        if (DYN == 3 && ROWST) {
            // LDS image of the wave's rows: [basin in wave][i*M + j]
            const int M = d.M, rowf = NP * M, bl = (threadIdx.x & 63) >> A.lgMp;
            if (L.jm < M) {
#pragma unroll
                for (int i = 0; i < NP; i++) {
                    const float gu = gp[i] * (d.p[i].hi - d.p[i].lo);
                    rowbuf[bl * rowf + i * M + L.jm] = raw ? gu * (D.ud[i] * (1.0f - D.ud[i])) : gu;
                }
            }
            // (one wave per workgroup: LDS traffic is ordered by the hardware, no barrier)
            const int b0 = blk.bx * (64 >> A.lgMp);
            const int nb = min(64 >> A.lgMp, d.B - b0);
            const int half = rowf >> 1, total2 = nb * half;                 // rowf is even (host)
            float *gbase = io.g[0].dyn + (int64_t)t * io.g[0].dyn_t_stride + (int64_t)b0 * io.g[0].dyn_b_stride;
            for (int k = threadIdx.x & 63; k < total2; k += 64) {
                const int rb = k / half, w2 = k - rb * half;
                const float2 v = *reinterpret_cast<const float2 *>(&rowbuf[rb * rowf + 2 * w2]);
                *reinterpret_cast<float2 *>(gbase + (int64_t)rb * io.g[0].dyn_b_stride + 2 * w2) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < NP; i++) {
            if (DYN == 1 || DYN == 3) break;
            const float gu = gp[i] * (d.p[i].hi - d.p[i].lo);
            if (DYN == 2) {
                // io.g[i].dyn is only set for dynamic slots; use_dyn is false for static ones
                const float gr = raw ? gu * (D.ud[i] * (1.0f - D.ud[i])) : gu;
                if (io.g[i].dyn && L.active)
                    io.g[i].dyn[(int64_t)t * io.g[i].dyn_t_stride + (int64_t)L.b * io.g[i].dyn_b_stride + L.j] =
                        use_dyn[i] ? gr : 0.0f;
                gsta[i] += use_dyn[i] ? 0.0f : gu;
            } else {
                gsta[i] += gu;
            }
        }
        if (io.grad_x) {
            const float act = L.active ? 1.0f : 0.0f;
            const float g0 = chunk_ens_sum(gx[0] * act, A.lgMp), g1 = chunk_ens_sum(gx[1] * act, A.lgMp),
                        g2 = chunk_ens_sum(gx[2] * act, A.lgMp);
            if (L.leader) {
                float *gr = io.grad_x + (int64_t)t * d.x_t_stride + (int64_t)L.b * d.x_b_stride;
                gr[d.ch_prcp] = g0; gr[d.ch_tmean] = g1; gr[d.ch_pet] = g2;
            }
        }
    }
    if (DYN == 1) {
        // the same daily terms were added to gsta and gused in the same order: exact cancellation
#pragma unroll
        for (int k = 0; k < CHUNK_FEW; k++)
            if (k < nd_) slot_set<NP>(gsta, ds_[k], slot_get<NP>(gsta, ds_[k]) - gused[k]);
    }
    if (DYN != 3 && L.active) {
#pragma unroll
        for (int i = 0; i < NP; i++) A.gpart[((int64_t)chunk * NP + i) * N + L.n] = gsta[i];
    }
}

#ifndef HBVX_CHUNK_NO_SHARED_KERNELS
// ---- B4 -------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bwd_chunk_reduce(const ChunkArgs A, int NP)
{
    const hbvx_desc &d = A.d;
    const hbvx_bwd_io &io = A.io;
    const int64_t N = (int64_t)d.B * d.M;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (n >= N || !io.g[i].sta) return;
    const float *src = A.gpart + (int64_t)i * N + n;
    const int64_t cs = (int64_t)NP * N;
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;
    int c = 0;
    for (; c + 3 < A.nchunk; c += 4) {
        acc0 += src[(int64_t)c * cs];
        acc1 += src[(int64_t)(c + 1) * cs];
        acc2 += src[(int64_t)(c + 2) * cs];
        acc3 += src[(int64_t)(c + 3) * cs];
    }
    for (; c < A.nchunk; c++) acc0 += src[(int64_t)c * cs];
    float gsum = (acc0 + acc1) + (acc2 + acc3);
    const int b = (int)(n / d.M), j = (int)(n % d.M);
    if (d.raw_sigmoid) {
        const float u = sigmoid_(d.p[i].sta[(int64_t)b * d.p[i].sta_b_stride + j]);
        gsum = gsum * (u * (1.0f - u));
    }
    float *dst = io.g[i].sta + (int64_t)b * io.g[i].sta_b_stride + j;
    *dst += gsum;
}

#endif

} // namespace hbvx
