// hbv_pipe.h -- pipelined forward for HBV 1.0 with static parameters.
//
// In HBV 1.0 one day is a feed-forward chain of three sub-recurrences
//     snow (SNOWPACK, MELTWATER) -> soil (SM) -> groundwater (SUZ, SLZ)
// (hbv.py:429-459 -> :462-480 -> :483-494; no storage feeds back into an earlier stage; HBV 1.1p
// and 2.0 break this with capillary rise SLZ -> SM).  Each sub-recurrence is serial in time, but
// stage s of tile k only needs stage s-1 of the same tile.  So three waves of one workgroup run the
// three stages one tile apart -- a software pipeline through LDS -- and the per-day latency of the
// workgroup becomes the LONGEST stage (soil: the pow and two divisions) instead of the sum.
//
//   wave 0  snow         tile it      reads forcings, writes RAIN/tosoil for the soil wave
//   wave 1  soil         tile it-1    reads RAIN/tosoil + PET, writes recharge/excess for wave 2
//   wave 2  groundwater  tile it-2
//   helpers (loads and stores share the in-order vmcnt counter, so no wave does both):
//     filler        forcings of tile it+3 -> registers, tile it+2 registers -> LDS (never waits
//                   for a load it has just issued)
//     row drainers  storages / pow results of tiles it-1..it-3 -> trajectory / aux rows
//     reducers      ensemble means of tiles it-1..it-3 -> flux series
//
// The stepper waves are bound by vector-ALU issue: the soil wave keeps its SIMD's VALU ~90 % busy
// (measured with the -DPIPE_PROBE build, tools/pipe_probe.py).  A workgroup's waves go to the four
// SIMDs cyclically, so waves w and w+4 share a SIMD: helpers on the soil wave's SIMD (5, 9, 13) stay
// idle, the row drainers sit on the SIMD without a stepper (7, 11, 15, next to the filler 3), the
// reducers next to the lighter snow / groundwater steppers (4, 8, 12 and 6, 10, 14).  Helper stores
// use buffer addressing: row base in SGPRs, constant per-lane offset, out-of-range lanes dropped by
// the hardware -- no 64-bit VALU address arithmetic, no EXEC masking.
//
// HBV 1.1p / 2.0: capillary rise feeds the lower zone back into the soil (SLZ -> SM), so soil and
// groundwater form ONE recurrence; the pipeline then has two stages (snow | soil + capillary +
// groundwater on wave 1, wave 2 idle) and the groundwater tile is drained one iteration earlier.
//
// One raw s_barrier per iteration; every interface is double-buffered (forcings: 4 slots, filled
// two tiles ahead and read by two stages).  The arithmetic is Step::fwd_snow / fwd_soil / fwd_gw of hbv_step.h, i.e.
// operation for operation what the tiled kernel computes; ensemble sums use the same member order,
// so both kernels give bit-identical results.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/hbvx.h"
#include "hbv_step.h"
#include "hbv_tiled.h"
#include "hbv_adj_step.h"

namespace hbvx {

#ifndef PIPE_KT
#define PIPE_KT 8       // days per tile with at most PIPE_FEWDYN dynamic parameters (host and device)
#endif
#ifndef PIPE_KT_STATIC
#define PIPE_KT_STATIC 10  // ... without any: no staged parameter rows in LDS, so longer tiles fit (140-145 KB of the 160) and
#endif                     // the workgroup's barrier comes every ten days: forward 0.98 -> 0.91 ms at config 2, six rounds each
                           // (profiles/r04_pipe_helpers_probe.txt; eleven days no longer fit)
#define PIPE_KT_MANY 4  // ... with more: the staged parameter rows need the LDS
#define PIPE_FEWDYN 3   // rows staged one per filler wave
#define PIPE_MAXDYN 18  // rows staged at most (shared by the three or four filler waves)
// Helper roles, switchable for A/B builds (tools/build_variant.sh; the defaults are what was measured best):
#ifndef PIPE_REBALANCE
#define PIPE_REBALANCE 1   // the forcing filler comes last in the row rotation also with <= 3 dynamic rows
#endif
#ifndef PIPE_Q1
#define PIPE_Q1 2          // the three waves on the soil wave's SIMD (5, 9, 13): 0 idle, 1 reducers (few dynamic
#endif                     // parameters only), 2 row drainers (whenever a trajectory is kept, not with MANY)

// Run `body(tt, has_next)` for the nt days of a tile; full tiles are unrolled so that every LDS
// address is base + immediate and the loop-carried registers need no rotation moves.
#define PIPE_DAYS(nt, body)                                                                        \
    do {                                                                                           \
        if ((nt) == KT && !PIPE_NO_UNROLL) {                                                       \
            _Pragma("unroll") for (int tt_ = 0; tt_ < KT; tt_++) body(tt_, tt_ + 1 < KT);          \
        } else {                                                                                   \
            for (int tt_ = 0; tt_ < (nt); tt_++) body(tt_, tt_ + 1 < (nt));                        \
        }                                                                                          \
    } while (0)

struct PipeArgs {
    hbvx_desc d;
    hbvx_fwd_out o;
    int lgMp;
    int Kt;
    int ckptK;   // HBVX_TRAJ_CKPT: keep only the storages entering every ckptK-th day ([ceil(T/K),5,N]); 0: rows
    // hbvx_fwd_out.zero_ptr: blocks nwg .. gridDim.x-1 of the launch take no part in the recurrence -- they sit on the CUs
    // the 168-workgroup latency chain leaves idle and write the caller's zeros, a 256 KB piece per wave at a time, until
    // the recurrence's workgroups have finished (fill_state[1]); fill_state[0] counts the pieces claimed.  fill_n16 = 0: none
    void *fill_ptr;
    unsigned *fill_state;
    unsigned long long fill_n16;
    int nwg;          // workgroups of the recurrence
    int fill_waves;   // waves of a fill workgroup that write (the others leave at once): the fill's share of HBM
};

typedef float pipe_f4 __attribute__((ext_vector_type(4)));
constexpr unsigned PIPE_FILL_PIECE16 = HBVX_ZERO_PIECE / 16;   // 16-byte stores per piece

// The last stage of a recurrence workgroup has stored its final storages: one count per workgroup.
__device__ __forceinline__ void pipe_signal_done(const PipeArgs &A)
{
    if (A.fill_n16 && (threadIdx.x & 63) == 0)
        __hip_atomic_fetch_add(A.fill_state + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The fill role of k_fwd_pipe's surplus workgroups.  Every wave for itself: look whether the recurrence is still running,
// claim the next piece, write it with nontemporal 16-byte stores (a wave instruction = 1 KB contiguous).  A claimed piece
// is always completed, so fill_state[0] pieces are zero when the launch ends.  No LDS, no barrier: the workgroup leaves
// through here.
__device__ __forceinline__ void pipe_fill_role(const PipeArgs &A)
{
    const unsigned lane = threadIdx.x & 63;
    if ((int)(threadIdx.x >> 6) >= A.fill_waves) return;
    const unsigned long long npiece = (A.fill_n16 + PIPE_FILL_PIECE16 - 1) / PIPE_FILL_PIECE16;
    pipe_f4 *base = static_cast<pipe_f4 *>(A.fill_ptr);
    const pipe_f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    for (;;) {
        unsigned done = 0, c = 0;
        if (lane == 0) done = __hip_atomic_load(A.fill_state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane(done) >= (unsigned)A.nwg) return;
        if (lane == 0) c = __hip_atomic_fetch_add(A.fill_state, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c >= npiece) return;
        const unsigned long long i0 = (unsigned long long)c * PIPE_FILL_PIECE16;
        const unsigned long long left = A.fill_n16 - i0;
        const unsigned n = left < PIPE_FILL_PIECE16 ? (unsigned)left : PIPE_FILL_PIECE16;
        pipe_f4 *p = base + i0;
#pragma unroll 4
        for (unsigned i = lane; i < n; i += 64) __builtin_nontemporal_store(z, p + i);
    }
}

// LDS layout in floats for Kt days per tile
struct PipeLds {
    int xin, ab, bc, oa, ob, oc, pin, total;
    __host__ __device__ explicit PipeLds(int Kt, int pd = 0, bool cap = false)
    {
        const int obr = cap ? 8 : 7;   // + capillary flux
        xin = 0;                    // [4][Kt][64][4]
        ab = xin + 4 * Kt * 256;    // [2][Kt][2][64]
        bc = ab + 2 * Kt * 128;     // [2][Kt][2][64]
        oa = bc + 2 * Kt * 128;     // [2][Kt][4][64]  SWE, tosoil | SNOWPACK, MELTWATER
        ob = oa + 2 * Kt * 256;     // [2][Kt][7|8][64]  AET, recharge, excs, evapfactor (, capillary) | SM, sw0, ef0
        oc = ob + 2 * Kt * obr * 64; // [2][Kt][7][64]  Qsim, Q0, Q1, Q2, PERC | SUZ, SLZ
        pin = oc + 2 * Kt * 448;    // [5][Kt][pd][64]  de-scaled dynamic parameters (pd rows per day)
        total = pin + 5 * Kt * pd * 64;
    }
};

// Drain work of the helper waves.  A stage's output tile holds NSER series per day: NFS flux series
// first (global flux index = nibble i of FMAP), then NSER - NFS storage series.  Everything that
// indexes the layout is a compile-time constant.
//   pipe_reduce_pass: ensemble means of 64 (day, series, basin) items -> flux series
//   muq: learned ensemble weights (hbv.py:508-511): the stepper staged Qsim x weight, so series QSIM is the plain SUM
//   over the members, every other series the mean (what hbv_tiled.h does: same products, same add tree)
template <int NSER, int NFS, unsigned FMAP>
__device__ __forceinline__ void pipe_reduce_pass(const hbvx_desc &d, const hbvx_fwd_out &o, const float *buf,
                                                 int t0, int items, int pass, int lane, int lgMp, int b0, bool muq = false)
{
    const int T = d.T, B = d.B;
    const float invM = 1.0f / (float)d.M;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(o.flux, 0, -1, 0x00020000);
    ens_reduce_pass<NSER, NFS>(buf, items, pass * 64, lane, d.M, lgMp, [&](int tt, int ks, int bl, float acc) {
        const int kk = (int)((FMAP >> (4 * ks)) & 0xFu);
        const unsigned off = b0 + bl < B ? (unsigned)(((kk * T + (t0 + tt)) * B + b0 + bl) * 4) : 0xFFFFFFFFu;
        const float v = (muq && kk == HBVX_F_QSIM) ? acc : acc * invM;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
    });
}

#ifdef PIPE_PROBE
__device__ unsigned long long g_pipe_probe[32];
__device__ unsigned long long g_pipe_blocks[4096];   // busy + wait of the soil wave, per workgroup
#define PIPE_BARRIER()                                                                             \
    do {                                                                                           \
        unsigned long long t_a = __builtin_readcyclecounter();                                     \
        probe_busy += t_a - probe_t;                                                               \
        lds_barrier();                                                                             \
        probe_t = __builtin_readcyclecounter();                                                    \
        probe_wait += probe_t - t_a;                                                               \
    } while (0)
#else
#define PIPE_BARRIER() lds_barrier()
#endif

// Dynamic parameters in the stepper waves: DY_DECL names the per-parameter flag / LDS row / prefetch
// register, DY_LOAD reads the value of a day from the staged tile (one day ahead of its use, like
// the forcings), DY_USE moves it into the parameter array.
// Branch-free: a static parameter reads row 0 of the tile (always valid) and keeps its value through
// a select, so the stepper loop has no scalar branches.
// SC (slot combo, like the adjoint's SlotCombo): when the dynamic set is known at compile time -- 1:
// {BETA, BETAET}, 2: {BETA, K0, BETAET}, the delta-MG defaults -- the flags fold, the static parameters'
// reciprocals and products leave the day loop and the stages without a dynamic parameter run their
// static code.  0: flags from the descriptor.
template <int SC>
__device__ __forceinline__ constexpr int pipe_sc_flag(int X)
{
    return SC == 0 ? -1 : ((X == P_BETA || X == P_BETAET || (SC == 2 && X == P_K0)) ? 1 : 0);
}
#define DY_DECL(X) const bool dy_##X = DYN && (pipe_sc_flag<SC>(X) >= 0 ? pipe_sc_flag<SC>(X) != 0 : (((dmask >> X) & 1u) != 0)); const int ix_##X = dy_##X ? dyn_index(dmask, X) * 64 : 0; float pn_##X = 0.0f
#define DY_LOAD(X, ptr) do { pn_##X = (ptr)[ix_##X]; } while (0)
#define DY_USE(X) do { p[X] = dy_##X ? pn_##X : p[X]; } while (0)

// TRAJ: the forward also saves the storage trajectory (and, in HBVX_SAVE_POW builds, the pow results: traj and aux given).
//   1: through LDS and row-drainer waves (the steppers touch only LDS + VALU);  2: the steppers store it themselves and
//   every helper that is not a filler reduces (DIRECT below).  Measured (profiles/r04_ab_direct.txt): with 16 members
//   (4 basins per wave, 5.5 reducer passes per 8-day tile) form 1 is 15 % faster -- a buffer store costs the lone
//   stepper wave more than the LDS write it replaces; with 4 members (16 basins per wave: 24 passes per tile) the
//   reducers bind and twice as many of them make form 2 36 % faster.  The host picks by basins per wave.
// DYN: parameters vary per day: filler waves de-scale them (sigmoid, range, dy_drop blend) into LDS
// tiles five deep (snow reads tile it, groundwater tile it-2, the fillers write it+2).
// MANY: more than PIPE_FEWDYN of them: 4-day tiles, the filler waves share the rows round-robin.
template <int MODEL, bool BETAET, int TRAJ, bool DYN, bool MANY = false, int SC = 0>
__global__ void __launch_bounds__(1024) k_fwd_pipe(const PipeArgs A)
{
    // ADJ: the implicit scheme (hbv_adj.py) with the staged solve of hbv_adj_step.h -- its blocks snow -> soil
    // moisture -> upper / lower zone feed forward exactly like the explicit HBV 1.0 stages, so the same three
    // waves run them one tile apart; flux = Q only, no saved powers
    constexpr bool ADJ = MODEL == MODEL_HBVADJ;
    constexpr bool PIPE_NO_UNROLL = ADJ;   // a day of the implicit scheme is an iteration, not 70 straight-line instructions
    constexpr bool CAP = MODEL != MODEL_HBV10 && !ADJ;   // two-stage pipeline (see the header comment)
    constexpr int KT = MANY ? PIPE_KT_MANY : (DYN ? PIPE_KT : PIPE_KT_STATIC);
    constexpr int OBR = CAP ? 8 : 7, NFB = CAP ? 5 : 4;
#ifdef PIPE_PROBE
    unsigned long long probe_busy = 0, probe_wait = 0, probe_t = __builtin_readcyclecounter();
    struct ProbeFlush {
        unsigned long long &b, &w;
        __device__ ~ProbeFlush()
        {
            if ((threadIdx.x >> 6) == 1 && (threadIdx.x & 63) == 0 && blockIdx.x < 4096) g_pipe_blocks[blockIdx.x] = b + w;
            if (blockIdx.x == 5 && (threadIdx.x & 63) == 0) {
                g_pipe_probe[(threadIdx.x >> 6) * 2] = b;
                g_pipe_probe[(threadIdx.x >> 6) * 2 + 1] = w;
            }
        }
    } probe_flush{probe_busy, probe_wait};
#endif
    constexpr int NP = NParamT<MODEL, BETAET>::value;
    extern __shared__ __align__(16) float lds[];
    if (A.fill_n16 && (int)blockIdx.x >= A.nwg) {
        pipe_fill_role(A);
        return;
    }
    const hbvx_desc &d = A.d;
    const hbvx_fwd_out &o = A.o;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const LaneT L = lane_t(d, A.lgMp);
    const int lane = L.lane;
    const int T = d.T, Kt = A.Kt;
    const int nT = (T + Kt - 1) / Kt;
    const int nIt = nT + 3;
    const int64_t N = (int64_t)d.B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const unsigned row_bytes = (unsigned)(N * 4);
    const unsigned voff = L.active ? (unsigned)(L.n * 4) : 0xFFFFFFFFu;
    unsigned dmask = 0;
    if (DYN) {
#pragma unroll
        for (int i = 0; i < NP; i++) dmask |= d.p[i].dyn ? (1u << i) : 0u;
    }
    // learned ensemble weights `muwts` [T,B,nmul] travel as one more staged row behind the dynamic parameters' (raw: no
    // sigmoid, no range): the stage that forms Qsim multiplies, the reducers SUM that series (hbv.py:508-511)
    const bool has_mu = DYN && !ADJ && d.muwts != nullptr;
    const int mrow = __builtin_popcount(dmask);
    const int ix_mu = has_mu ? mrow * 64 : 0;
    const int PD = DYN ? (MANY ? mrow + (has_mu ? 1 : 0) : PIPE_FEWDYN) : 0;   // staged rows per day
    const PipeLds P(Kt, PD, CAP);
    const float nz = d.nearzero;
    const float ac = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.ac[L.b] : 0.0f;
    const float elev = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.elev[L.b] : 0.0f;
    auto pin_of = [&](int tile) { return lds + P.pin + (tile % 5) * Kt * PD * 64 + lane; };

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
    auto tile_nt = [&](int tile) { return min(Kt, T - tile * Kt); };

    // DIRECT (TRAJ == 2): the stepper waves store the storages entering each day straight to the trajectory rows -- one
    // buffer store where they otherwise write the value to LDS for a drainer wave -- so the six drainer waves are free
    // to reduce.  One descriptor per storage series and tile (base = the row of the tile's
    // first day, range = the tile's rows: only one tile has to fit 32 bits), day offset in the scalar operand, the
    // lane's offset constant, out-of-range lanes dropped by the hardware.
    constexpr bool DIRECT = TRAJ == 2 && !SAVE_POW;
    const int64_t SRt = (int64_t)(T + 1) * N;
    auto trj_rsrc = [&](int k, int tile) {
        return A.ckptK ? __builtin_amdgcn_make_buffer_rsrc(o.traj, 0, -1, 0x00020000)
                       : __builtin_amdgcn_make_buffer_rsrc(o.traj + k * SRt + (int64_t)tile * Kt * N, 0, (int)(row_bytes * KT),
                                                           0x00020000);
    };
    auto trj_put = [&](__amdgpu_buffer_rsrc_t r, int tile, int tt, int k, float v) __attribute__((always_inline)) {
        if (A.ckptK) {   // K-day checkpoints: rows [(day / K) * 5 + storage], only the days that are multiples of K
            const int day = tile * Kt + tt;
            if (day % A.ckptK == 0)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff,
                                                      (unsigned)(((int64_t)(day / A.ckptK) * 5 + k) * row_bytes), 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, (unsigned)tt * row_bytes, 0);
        }
    };

    if (wave == 0) {
        // ------------------------------ snow ------------------------------
        __builtin_amdgcn_s_setprio(3);
        constexpr float S0 = ADJ ? 0.0f : 0.001f;   // hbv.py:128-136 / hbv_adj.py:254
        float SP = d.state_in ? d.state_in[0 * N + L.n] : S0;
        float MW = d.state_in ? d.state_in[1 * N + L.n] : S0;
        PIPE_BARRIER();
        for (int it = 0; it < nIt; it++) {
            const int tile = it;
            if (tile < nT) {
                const int nt = tile_nt(tile);
                const float4 *in4 = reinterpret_cast<const float4 *>(lds + P.xin + (tile & 3) * Kt * 256);
                float *ab = lds + P.ab + (tile & 1) * Kt * 128 + lane;
                float *oa = lds + P.oa + (tile & 1) * Kt * 256 + lane;
                float4 fn = in4[lane];
                const float *pin = pin_of(tile);
                const auto rSP = trj_rsrc(0, tile), rMW = trj_rsrc(1, tile);
                DY_DECL(P_TT); DY_DECL(P_CFMAX); DY_DECL(P_CFR); DY_DECL(P_CWH);
                if (DYN) { DY_LOAD(P_TT, pin); DY_LOAD(P_CFMAX, pin); DY_LOAD(P_CFR, pin); DY_LOAD(P_CWH, pin); }
                auto day = [&](int tt, bool more) __attribute__((always_inline)) {
                    const float4 f = fn;
                    if (DYN) { DY_USE(P_TT); DY_USE(P_CFMAX); DY_USE(P_CFR); DY_USE(P_CWH); }
                    if (more) {
                        fn = in4[(tt + 1) * 64 + lane];
                        if (DYN) {
                            const float *pt = pin + (tt + 1) * PD * 64;
                            DY_LOAD(P_TT, pt); DY_LOAD(P_CFMAX, pt); DY_LOAD(P_CFR, pt); DY_LOAD(P_CWH, pt);
                        }
                    }
                    float *q = oa + tt * 256;
                    if constexpr (ADJ) {
                        float y0, y1, rfv, isn;
                        AdjStaged<BETAET>::snow(p, f.x, f.y, 1.0f, SP, MW, y0, y1, rfv, isn);
                        ab[tt * 128] = rfv;
                        ab[tt * 128 + 64] = isn;
                        if (TRAJ) {
                            if (DIRECT) { trj_put(rSP, tile, tt, 0, SP); trj_put(rMW, tile, tt, 1, MW); }
                            else { q[128] = SP; q[192] = MW; }
                        }
                        SP = y0; MW = y1;
                    } else {
                        Step<MODEL, BETAET> s;
                        s.P = f.x; s.Tf = f.y;
                        s.SP = SP; s.MW = MW;
                        s.fwd_snow(p, elev);
                        ab[tt * 128] = s.RAIN;
                        ab[tt * 128 + 64] = s.tosoil;
                        q[0] = s.SP3; q[64] = s.tosoil;
                        if (TRAJ) {
                            if (DIRECT) { trj_put(rSP, tile, tt, 0, SP); trj_put(rMW, tile, tt, 1, MW); }
                            else { q[128] = SP; q[192] = MW; }
                        }
                        SP = s.SP3; MW = s.MW3;
                    }
                };
                PIPE_DAYS(nt, day);
            }
            PIPE_BARRIER();
        }
        if (L.active) {
            o.state_out[0 * N + L.n] = SP;
            o.state_out[1 * N + L.n] = MW;
            if (TRAJ && !A.ckptK) {
                o.traj[((int64_t)0 * (T + 1) + T) * N + L.n] = SP;
                o.traj[((int64_t)1 * (T + 1) + T) * N + L.n] = MW;
            }
        }
    } else if (wave == 1) {
      if constexpr (CAP) {
        // -------------- soil + capillary rise + groundwater (one recurrence) --------------
        __builtin_amdgcn_s_setprio(3);
        float SM = d.state_in ? d.state_in[2 * N + L.n] : 0.001f;
        float SUZ = d.state_in ? d.state_in[3 * N + L.n] : 0.001f;
        float SLZ = d.state_in ? d.state_in[4 * N + L.n] : 0.001f;
        PIPE_BARRIER();
        for (int it = 0; it < nIt; it++) {
            const int tile = it - 1;
            if (tile >= 0 && tile < nT) {
                const int nt = tile_nt(tile);
                const float4 *in4 = reinterpret_cast<const float4 *>(lds + P.xin + (tile & 3) * Kt * 256);
                const float *ab = lds + P.ab + (tile & 1) * Kt * 128 + lane;
                float *ob = lds + P.ob + (tile & 1) * Kt * OBR * 64 + lane;
                float *oc = lds + P.oc + (tile & 1) * Kt * 448 + lane;
                float npet = in4[lane].z, nrain = ab[0], nts = ab[64];
                const float *pin = pin_of(tile);
                const auto rSM = trj_rsrc(2, tile), rSUZ = trj_rsrc(3, tile), rSLZ = trj_rsrc(4, tile);
                DY_DECL(P_BETA); DY_DECL(P_FC); DY_DECL(P_LP); DY_DECL(P_BETAET); DY_DECL(P_C);
                DY_DECL(P_K0); DY_DECL(P_K1); DY_DECL(P_K2); DY_DECL(P_PERC); DY_DECL(P_UZL);
                DY_DECL(P_RT); DY_DECL(P_AC); DY_DECL(P_F0); DY_DECL(P_FMIN); DY_DECL(P_ALPHA);
#define DY_ALL(OP, ptr)                                                                            \
    do {                                                                                           \
        OP(P_BETA, ptr); OP(P_FC, ptr); OP(P_LP, ptr); OP(P_BETAET, ptr); OP(P_C, ptr); OP(P_K0, ptr); \
        OP(P_K1, ptr); OP(P_K2, ptr); OP(P_PERC, ptr); OP(P_UZL, ptr);                             \
        if (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) { OP(P_RT, ptr); OP(P_AC, ptr); }       \
        if (MODEL == MODEL_HOURLY) { OP(P_F0, ptr); OP(P_FMIN, ptr); OP(P_ALPHA, ptr); }           \
    } while (0)
#define DY_USE2(X, unused) DY_USE(X)
                // Arbitrary dynamic sets (SC == 0) of HBV 2.0 / hourly on this fused stage: up to 17 staged
                // rows.  One-day-ahead prefetch registers for all of them pushed the wave over its 128
                // VGPRs (3-44 spilled values per lane); there the values are read from the LDS tile at the
                // start of their day instead (independent reads, one wait).  HBV 1.1p (14 rows) fits and
                // keeps the prefetch: without it config 3's forward went from 2.6 to 3.8 ms.
                constexpr bool PREF = SC != 0 || MODEL == MODEL_HBV11P;
#define DY_NOW(X, ptr) do { p[X] = dy_##X ? (ptr)[ix_##X] : p[X]; } while (0)
                if (DYN && PREF) DY_ALL(DY_LOAD, pin);
                auto day = [&](int tt, bool more) __attribute__((always_inline)) {
                    Step<MODEL, BETAET> s;
                    s.PET = npet; s.RAIN = nrain; s.tosoil = nts;
                    if (DYN && PREF) DY_ALL(DY_USE2, 0);
                    if (DYN && !PREF) {
                        const float *pc = pin + tt * PD * 64;
                        DY_ALL(DY_NOW, pc);
                    }
                    if (more) {
                        npet = in4[(tt + 1) * 64 + lane].z;
                        nrain = ab[(tt + 1) * 128];
                        nts = ab[(tt + 1) * 128 + 64];
                        if (DYN && PREF) {
                            const float *pt = pin + (tt + 1) * PD * 64;
                            DY_ALL(DY_LOAD, pt);
                        }
                    }
                    s.SM = SM; s.SUZ = SUZ; s.SLZ = SLZ;
                    if constexpr (MODEL == MODEL_HOURLY) {
                        s.template fwd_rest<false>(p, nz, ac, 0.0f, 0.0f);
                    } else {
                        s.template fwd_soil<false, true>(p, nz, 0.0f, 0.0f);
                        s.fwd_cap(p, nz);
                        s.fwd_gw(p, ac);
                    }
                    float *q = ob + tt * OBR * 64;
                    q[0] = s.ET; q[64] = s.rech; q[128] = s.exc; q[192] = s.ef; q[256] = s.cap;
                    if (TRAJ && !DIRECT) { q[320] = SM; if (SAVE_POW) { q[384] = s.sw0; q[448] = s.ef0; } }
                    float *r = oc + tt * 448;
                    // (the weight is read where it is used: it only scales an OUTPUT, off the day-to-day chain, and a
                    // prefetch register for it pushed three instances of this stage over their 128)
                    r[0] = has_mu ? s.Q * pin[tt * PD * 64 + ix_mu] : s.Q; r[64] = s.Q0; r[128] = s.Q1; r[192] = s.Q2; r[256] = s.PERC;
                    if (TRAJ) {
                        if (DIRECT) { trj_put(rSM, tile, tt, 2, SM); trj_put(rSUZ, tile, tt, 3, SUZ); trj_put(rSLZ, tile, tt, 4, SLZ); }
                        else { r[320] = SUZ; r[384] = SLZ; }
                    }
                    SM = s.SM4; SUZ = s.SUZ4; SLZ = s.SLZ2;
                };
                PIPE_DAYS(nt, day);
#undef DY_ALL
#undef DY_USE2
#undef DY_NOW
            }
            PIPE_BARRIER();
        }
        pipe_signal_done(A);
        if (L.active) {
            o.state_out[2 * N + L.n] = SM;
            o.state_out[3 * N + L.n] = SUZ;
            o.state_out[4 * N + L.n] = SLZ;
            if (TRAJ && !A.ckptK) {
                o.traj[((int64_t)2 * (T + 1) + T) * N + L.n] = SM;
                o.traj[((int64_t)3 * (T + 1) + T) * N + L.n] = SUZ;
                o.traj[((int64_t)4 * (T + 1) + T) * N + L.n] = SLZ;
            }
        }
      } else {
        // ------------------------------ soil ------------------------------
        __builtin_amdgcn_s_setprio(3);
        float SM = d.state_in ? d.state_in[2 * N + L.n] : (ADJ ? 0.0f : 0.001f);
        PIPE_BARRIER();
        for (int it = 0; it < nIt; it++) {
            const int tile = it - 1;
            if (tile >= 0 && tile < nT) {
                const int nt = tile_nt(tile);
                const float4 *in4 = reinterpret_cast<const float4 *>(lds + P.xin + (tile & 3) * Kt * 256);
                const float *ab = lds + P.ab + (tile & 1) * Kt * 128 + lane;
                float *bc = lds + P.bc + (tile & 1) * Kt * 128 + lane;
                float *ob = lds + P.ob + (tile & 1) * Kt * OBR * 64 + lane;
                float npet = in4[lane].z, nrain = ab[0], nts = ab[64];
                const float *pin = pin_of(tile);
                const auto rSM = trj_rsrc(2, tile);
                DY_DECL(P_BETA); DY_DECL(P_FC); DY_DECL(P_LP); DY_DECL(P_BETAET);
                if (DYN) { DY_LOAD(P_BETA, pin); DY_LOAD(P_FC, pin); DY_LOAD(P_LP, pin); DY_LOAD(P_BETAET, pin); }
                auto day = [&](int tt, bool more) __attribute__((always_inline)) {
                    const float dPET = npet, dRAIN = nrain, dTS = nts;
                    if (DYN) { DY_USE(P_BETA); DY_USE(P_FC); DY_USE(P_LP); DY_USE(P_BETAET); }
                    if (more) {
                        npet = in4[(tt + 1) * 64 + lane].z;
                        nrain = ab[(tt + 1) * 128];
                        nts = ab[(tt + 1) * 128 + 64];
                        if (DYN) {
                            const float *pt = pin + (tt + 1) * PD * 64;
                            DY_LOAD(P_BETA, pt); DY_LOAD(P_FC, pt); DY_LOAD(P_LP, pt); DY_LOAD(P_BETAET, pt);
                        }
                    }
                    float *q = ob + tt * OBR * 64;
                    if constexpr (ADJ) {
                        float y2, Peff, exs;
                        AdjStaged<BETAET>::soil(p, dRAIN, dTS, dPET, 1.0f, SM, d.adj_gtol, d.adj_max_iter, y2, Peff, exs);
                        bc[tt * 128] = Peff;
                        bc[tt * 128 + 64] = exs;
                        if (TRAJ) { if (DIRECT) trj_put(rSM, tile, tt, 2, SM); else q[256] = SM; }
                        SM = y2;
                    } else {
                        Step<MODEL, BETAET> s;
                        s.PET = dPET; s.RAIN = dRAIN; s.tosoil = dTS;
                        s.SM = SM;
                        s.template fwd_soil<false, true>(p, nz, 0.0f, 0.0f);
                        bc[tt * 128] = s.rech;
                        bc[tt * 128 + 64] = s.exc;
                        q[0] = s.ET; q[64] = s.rech; q[128] = s.exc; q[192] = s.ef;
                        if (TRAJ) { if (DIRECT) trj_put(rSM, tile, tt, 2, SM); else { q[256] = SM; if (SAVE_POW) { q[320] = s.sw0; q[384] = s.ef0; } } }
                        SM = s.SM3;
                    }
                };
                PIPE_DAYS(nt, day);
            }
            PIPE_BARRIER();
        }
        if (L.active) {
            o.state_out[2 * N + L.n] = SM;
            if (TRAJ && !A.ckptK) o.traj[((int64_t)2 * (T + 1) + T) * N + L.n] = SM;
        }
      }
    } else if (wave == 2 && !(CAP && MANY)) {
      if constexpr (CAP) {
        PIPE_BARRIER();
        for (int it = 0; it < nIt; it++) PIPE_BARRIER();
      } else {
        // --------------------------- groundwater ---------------------------
        __builtin_amdgcn_s_setprio(3);
        float SUZ = d.state_in ? d.state_in[3 * N + L.n] : (ADJ ? 0.0f : 0.001f);
        float SLZ = d.state_in ? d.state_in[4 * N + L.n] : (ADJ ? 0.0f : 0.001f);
        PIPE_BARRIER();
        for (int it = 0; it < nIt; it++) {
            const int tile = it - 2;
            if (tile >= 0 && tile < nT) {
                const int nt = tile_nt(tile);
                const float *bc = lds + P.bc + (tile & 1) * Kt * 128 + lane;
                float *oc = lds + P.oc + (tile & 1) * Kt * 448 + lane;
                float nrech = bc[0], nexc = bc[64];
                const float *pin = pin_of(tile);
                const auto rSUZ = trj_rsrc(3, tile), rSLZ = trj_rsrc(4, tile);
                DY_DECL(P_K0); DY_DECL(P_K1); DY_DECL(P_K2); DY_DECL(P_PERC); DY_DECL(P_UZL);
                if (DYN) { DY_LOAD(P_K0, pin); DY_LOAD(P_K1, pin); DY_LOAD(P_K2, pin); DY_LOAD(P_PERC, pin); DY_LOAD(P_UZL, pin); }
                auto day = [&](int tt, bool more) __attribute__((always_inline)) {
                    const float dRECH = nrech, dEXC = nexc;
                    if (DYN) { DY_USE(P_K0); DY_USE(P_K1); DY_USE(P_K2); DY_USE(P_PERC); DY_USE(P_UZL); }
                    if (more) {
                        nrech = bc[(tt + 1) * 128];
                        nexc = bc[(tt + 1) * 128 + 64];
                        if (DYN) {
                            const float *pt = pin + (tt + 1) * PD * 64;
                            DY_LOAD(P_K0, pt); DY_LOAD(P_K1, pt); DY_LOAD(P_K2, pt); DY_LOAD(P_PERC, pt); DY_LOAD(P_UZL, pt);
                        }
                    }
                    float *q = oc + tt * 448;
                    if constexpr (ADJ) {
                        float y3, y4, Q;
                        AdjStaged<BETAET>::gw(p, dRECH, dEXC, 1.0f, SUZ, SLZ, y3, y4, Q);
                        q[0] = Q;
                        if (TRAJ) {
                            if (DIRECT) { trj_put(rSUZ, tile, tt, 3, SUZ); trj_put(rSLZ, tile, tt, 4, SLZ); }
                            else { q[320] = SUZ; q[384] = SLZ; }
                        }
                        SUZ = y3; SLZ = y4;
                    } else {
                        Step<MODEL, BETAET> s;
                        s.rech = dRECH; s.exc = dEXC;
                        s.SUZ = SUZ; s.SLZ0 = SLZ;
                        s.fwd_gw(p, 0.0f);
                        q[0] = has_mu ? s.Q * pin[tt * PD * 64 + ix_mu] : s.Q; q[64] = s.Q0; q[128] = s.Q1; q[192] = s.Q2; q[256] = s.PERC;
                        if (TRAJ) {
                            if (DIRECT) { trj_put(rSUZ, tile, tt, 3, SUZ); trj_put(rSLZ, tile, tt, 4, SLZ); }
                            else { q[320] = SUZ; q[384] = SLZ; }
                        }
                        SUZ = s.SUZ4; SLZ = s.SLZ2;
                    }
                };
                PIPE_DAYS(nt, day);
            }
            PIPE_BARRIER();
        }
        pipe_signal_done(A);
        if (L.active) {
            o.state_out[3 * N + L.n] = SUZ;
            o.state_out[4 * N + L.n] = SLZ;
            if (TRAJ && !A.ckptK) {
                o.traj[((int64_t)3 * (T + 1) + T) * N + L.n] = SUZ;
                o.traj[((int64_t)4 * (T + 1) + T) * N + L.n] = SLZ;
            }
        }
      }
    } else {
        // ------------------------------ helpers ------------------------------
        const int lgMp = A.lgMp, bpw = 64 >> lgMp;
        const int b0 = blockIdx.x * bpw;
        constexpr unsigned fmapA = HBVX_F_SWE | (HBVX_F_TOSOIL << 4);
        constexpr unsigned fmapB = HBVX_F_AET | (HBVX_F_RECHARGE << 4) | (HBVX_F_EXCS << 8) |
                                   (HBVX_F_EVAPFACTOR << 12) | (CAP ? (unsigned)HBVX_F_CAPILLARY << 16 : 0u);
        constexpr unsigned fmapC = HBVX_F_QSIM | (HBVX_F_Q0 << 4) | (HBVX_F_Q1 << 8) | (HBVX_F_Q2 << 12) |
                                   (HBVX_F_PERC << 16);
        const int nw = blockDim.x >> 6;
        const int quad = wave & 3;   // waves with the same value share a SIMD
        // fillers: wave 3 stages the forcings; with dynamic parameters waves 4, 6 and 3 (in that order)
        // share the dynamic rows -- the de-scaling (sigmoid) of a row is ~140 VALU cycles
        // per day, too much for one wave beside the forcings.
        // (in the two-stage pipeline wave 2 has no stage: with many dynamic rows it is a fourth filler)
        // (round 4, many rows: the three waves on the soil wave's SIMD -- idle in this mode -- fill as well.  Probe at
        // config 3 with four fillers: 694-784 busy of 793 cycles per day, the fused soil + capillary + groundwater
        // wave 640: the fillers bound the day)
        constexpr int NF = MANY ? (CAP ? 7 : 6) : 3;   // filler waves when DYN
        const bool is_fill = wave == 3 || (DYN && (wave == 4 || wave == 6 || (MANY && (wave == 2 || wave == 5 || wave == 9 || wave == 13))));
        const int rbase = DYN ? 8 : 4;   // first reducer wave
        if (is_fill) {
            // Registers hold the tile that goes to LDS next iteration.  Straight-line code (days past
            // the end of the record read 0 / are written but never consumed): with control flow
            // between the loads the compiler falls back to vmcnt(0) after every group.
            // Loads use buffer addressing: one descriptor per tensor and tile (base = first day of
            // the tile, range = the days that exist), the day offset is a scalar operand and the
            // per-lane offset a constant, so issuing a tile costs no vector address arithmetic.
            constexpr int FD = KT;
            // many rows: the three fillers on the soil wave's SIMD only get the issue slots that wave leaves (probe: 690-750
            // busy with two rows each against 350-400 for the others), so they take ONE row each -- rows 0 .. NS-1 -- and
            // the NF - NS others share the rest round-robin
            constexpr int NS = MANY ? 3 : 0, NFAST = NF - NS;
            constexpr int NRW = MANY ? (PIPE_MAXDYN - NS + NFAST - 1) / NFAST : 1;   // dynamic rows per filler wave
            const int fidx = wave == 3 ? 0 : (wave == 4 ? 1 : (wave == 6 ? 2 : (wave == 2 ? 3 : (CAP ? 4 : 3) + ((wave - 5) >> 2))));
            const bool forc = fidx == 0;
            const int nd = __builtin_popcount(dmask);
            float fx[FD], fy[FD], fz[FD], dv[NRW][FD];
            auto rsrc_of = [&](const float *base, int64_t row_floats, int rows) {
                return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0,
                                                         (int)(row_floats * 4 * rows), 0x00020000);
            };
            auto bload = [&](__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so) {
                return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
            };
            const unsigned xvo = (unsigned)(L.b * d.x_b_stride * 4);
            const unsigned xcp = d.ch_prcp * 4, xct = d.ch_tmean * 4, xce = d.ch_pet * 4;
            // this wave stages the dynamic rows frow, frow + NF, ... (row k = k-th set bit of dmask);
            // wave 3 also stages the forcings, so with many rows it comes last in the rotation and gets the
            // short share (with at most three rows the order 3, 4, 6 keeps them off the stage waves' SIMDs)
            // (with at most three rows too, the forcing wave comes last in the rotation: with the delta-MG default of
            // two dynamic parameters it stages no row at all -- it was the busiest wave of the workgroup, 453 of 489
            // cycles per day, with forcings AND a row)
            const bool slow = MANY && fidx >= NFAST;
            const int frow = slow ? fidx - NFAST : NS + ((MANY || PIPE_REBALANCE) ? (fidx + NFAST - 1) % NFAST : fidx);
            const int fstep = slow ? PIPE_MAXDYN : NFAST;      // (a slow filler's second row never exists)
            const float *dsrc[NRW];
            unsigned dvo[NRW];
            int64_t dts[NRW];
            float dlo[NRW], dhi[NRW], dsta[NRW];
            bool duse[NRW], dyrow[NRW], murow[NRW];
#pragma unroll
            for (int r = 0; r < NRW; r++) {
                const int k = frow + fstep * r;
                murow[r] = has_mu && k == nd;               // the ensemble weights: the row behind the parameters'
                dyrow[r] = (DYN && k < nd) || murow[r];
                dsrc[r] = d.x; dvo[r] = 0; dts[r] = 0; dlo[r] = dhi[r] = dsta[r] = 0.0f; duse[r] = false;
                if (murow[r]) {
                    dsrc[r] = d.muwts;
                    dvo[r] = (unsigned)((L.b * d.mu_b_stride + L.j) * 4);
                    dts[r] = d.mu_t_stride;
                } else if (dyrow[r]) {
                    int slot = 0;
                    for (int i = 0, c = 0; i < NP; i++)
                        if ((dmask >> i) & 1u) { if (c == k) slot = i; c++; }
                    const hbvx_param_src &ps = d.p[slot];
                    dsrc[r] = ps.dyn;
                    dvo[r] = (unsigned)((L.b * ps.dyn_b_stride + L.j) * 4);
                    dts[r] = ps.dyn_t_stride;
                    dlo[r] = ps.lo; dhi[r] = ps.hi;
                    float v = ps.sta[(int64_t)L.b * ps.sta_b_stride + L.j];
                    v = raw ? sigmoid_(v) : v;
                    dsta[r] = descale_(v, ps.lo, ps.hi);
                    duse[r] = ps.drop ? (ps.drop[ADJ ? L.n : (int64_t)L.b] == 0) : true;   // hbv_adj.py:182-189: per lane
                }
            }
            auto issue = [&](int tile) {
                const int t0 = tile * FD, rows = min(FD, T - t0);
                if (forc) {
                    const auto rx = rsrc_of(d.x + (int64_t)t0 * d.x_t_stride, d.x_t_stride, rows);
#pragma unroll
                    for (int i = 0; i < FD; i++) {
                        const unsigned so = (unsigned)(i * (int)d.x_t_stride * 4);
                        fx[i] = bload(rx, xvo, so + xcp); fy[i] = bload(rx, xvo, so + xct); fz[i] = bload(rx, xvo, so + xce);
                    }
                }
#pragma unroll
                for (int r = 0; r < NRW; r++)
                    if (dyrow[r]) {
                        const auto rd = rsrc_of(dsrc[r] + (int64_t)t0 * dts[r], dts[r], rows);
#pragma unroll
                        for (int i = 0; i < FD; i++) dv[r][i] = bload(rd, dvo[r], (unsigned)(i * (int)dts[r] * 4));
                    }
            };
            auto commit = [&](int tile) {
                if (forc) {
                    float4 *in4 = reinterpret_cast<float4 *>(lds + P.xin + (tile & 3) * FD * 256);
#pragma unroll
                    for (int i = 0; i < FD; i++) in4[i * 64 + lane] = make_float4(fx[i], fy[i], fz[i], 0.0f);
                }
#pragma unroll
                for (int r = 0; r < NRW; r++)
                    if (dyrow[r]) {
                        float *pin = lds + P.pin + (tile % 5) * FD * PD * 64 + (frow + fstep * r) * 64 + lane;
#pragma unroll
                        for (int i = 0; i < FD; i++) {
                            const float u = raw ? sigmoid_dyn_(dv[r][i]) : dv[r][i];
                            pin[i * PD * 64] = murow[r] ? dv[r][i] : (duse[r] ? descale_(u, dlo[r], dhi[r]) : dsta[r]);
                        }
                    }
            };
            issue(0);
            commit(0);
            if (nT > 1) { issue(1); commit(1); }
            if (nT > 2) issue(2);
            PIPE_BARRIER();
            for (int it = 0; it < nIt; it++) {
                // tile it+2 (loaded during the previous iteration) -> LDS slot (it+2)&3, free since
                // the soil wave finished tile it-2 before the last barrier; then start tile it+3
                if (it + 2 < nT) commit(it + 2);
                if (it + 3 < nT) issue(it + 3);
                PIPE_BARRIER();
            }
        } else if (DIRECT) {
            // every helper that is not a filler reduces.  Ranks go round the SIMDs -- the stepper-free one, the snow
            // wave's, the groundwater wave's -- row of waves by row, the soil wave's SIMD last (a tile has at most 6-8
            // passes, heaviest first: those waves only ever get work when the workgroup is short of others)
            int rank = -1, nred = 0;
            for (int pass = 0; pass < 2; pass++)
                for (int row = 0; row < 4; row++)
                    for (int qi = 0; qi < 3; qi++) {
                        const int q = pass == 0 ? (qi == 0 ? 3 : (qi == 1 ? 0 : 2)) : 1;
                        if (pass == 1 && qi > 0) continue;
                        const int wv = 4 * row + q + (q < 3 ? 4 : 0);     // waves 4.. (quad 0-2) / 3, 7, 11, 15 (quad 3)
                        if (wv < 3 || wv >= nw) continue;
                        const bool fill_w = wv == 3 || (DYN && (wv == 4 || wv == 6 || (MANY && (wv == 5 || wv == 9 || wv == 13))));
                        if (fill_w) continue;
                        if (wv == wave) rank = nred;
                        nred++;
                    }
            PIPE_BARRIER();
            for (int it = 0; it < nIt; it++) {
                const int tA = it - 1, tB = it - 2, tC = CAP ? it - 2 : it - 3;
                const int ntA = (tA >= 0 && tA < nT) ? tile_nt(tA) : 0;
                const int ntB = (tB >= 0 && tB < nT) ? tile_nt(tB) : 0;
                const int ntC = (tC >= 0 && tC < nT) ? tile_nt(tC) : 0;
                const float *bufA = lds + P.oa + (tA & 1) * Kt * 256;
                const float *bufB = lds + P.ob + (tB & 1) * Kt * OBR * 64;
                const float *bufC = lds + P.oc + (tC & 1) * Kt * 448;
                constexpr int NFC = ADJ ? 1 : 5;   // implicit scheme: Q only (hbv_adj.py:309-317)
                const int iA = ADJ ? 0 : ntA * 2 * bpw, iB = ADJ ? 0 : ntB * NFB * bpw,
                          iC = (ADJ && !o.flux) ? 0 : ntC * NFC * bpw;
                const int pA = (iA + 63) >> 6, pB = (iB + 63) >> 6, pC = (iC + 63) >> 6;
                const int nR = pA + pB + pC;
                if (rank >= 0) {
                    for (int u = rank; u < nR; u += nred) {
                        if (u < pC)
                            pipe_reduce_pass<7, NFC, fmapC>(d, o, bufC, tC * Kt, iC, u, lane, lgMp, b0, has_mu);
                        else if (u < pC + pB)
                            pipe_reduce_pass<OBR, NFB, fmapB>(d, o, bufB, tB * Kt, iB, u - pC, lane, lgMp, b0);
                        else
                            pipe_reduce_pass<4, 2, fmapA>(d, o, bufA, tA * Kt, iA, u - pC - pB, lane, lgMp, b0);
                    }
                }
                PIPE_BARRIER();
            }
        } else if ((quad == 1 && !(PIPE_Q1 == 1 && DYN && !MANY) && !(PIPE_Q1 == 2 && TRAJ && !MANY)) ||
                   !(quad == 3 ? TRAJ : true)) {
            // the soil wave's SIMD stays free; without a trajectory the row drainers have no work
            PIPE_BARRIER();
            for (int it = 0; it < nIt; it++) PIPE_BARRIER();
        } else if (quad == 3 || (PIPE_Q1 == 2 && quad == 1 && TRAJ && !MANY)) {
            // row drainers (waves 7, 11, 15): day tt of every stage tile -> 7 rows.  One buffer
            // descriptor per storage series and tile (base = row of the tile's first day, range =
            // the tile's Kt rows: the range check includes the scalar offset), the day inside the
            // tile goes into the scalar offset operand.
            // (PIPE_Q1 == 2: six drainers -- waves 7, 11, 15 and 5, 9, 13 -- take one or two of a tile's eight days each)
            constexpr bool SIX = PIPE_Q1 == 2 && !MANY;
            const int n3 = (nw - 7 + 3) >> 2;
            const int w = (SIX && quad == 1) ? n3 + ((wave - 5) >> 2) : (wave - 7) >> 2;
            const int NRD = n3 + (SIX ? (nw - 5 + 3) >> 2 : 0);
            const int64_t SR = (int64_t)(T + 1) * N;   // storage row blocks in traj
            auto rsrc = [&](float *base) {
                return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(row_bytes * KT), 0x00020000);
            };
            auto put = [&](__amdgpu_buffer_rsrc_t r, unsigned soff, float v) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
            };
            PIPE_BARRIER();
            for (int it = 0; it < nIt; it++) {
                const int tA = it - 1, tB = it - 2, tC = CAP ? it - 2 : it - 3;
                const int ntA = (tA >= 0 && tA < nT) ? tile_nt(tA) : 0;
                const int ntB = (tB >= 0 && tB < nT) ? tile_nt(tB) : 0;
                const int ntC = (tC >= 0 && tC < nT) ? tile_nt(tC) : 0;
                const float *bufA = lds + P.oa + (tA & 1) * Kt * 256 + lane;
                const float *bufB = lds + P.ob + (tB & 1) * Kt * OBR * 64 + lane;
                const float *bufC = lds + P.oc + (tC & 1) * Kt * 448 + lane;
                // (invalid tiles: nt = 0, the descriptors are built but never used)
                const int64_t dA = (int64_t)max(tA, 0) * Kt * N, dB = (int64_t)max(tB, 0) * Kt * N,
                              dC = (int64_t)max(tC, 0) * Kt * N;
                const auto rSP = rsrc(o.traj + dA), rMW = rsrc(o.traj + SR + dA);
                const auto rSM = rsrc(o.traj + 2 * SR + dB), rSW = rsrc((SAVE_POW ? o.aux : o.traj) + dB),
                           rEF = rsrc((SAVE_POW ? o.aux : o.traj) + (int64_t)T * N + dB);
                const auto rSUZ = rsrc(o.traj + 3 * SR + dC), rSLZ = rsrc(o.traj + 4 * SR + dC);
                if (A.ckptK) {
                    // checkpoints: of each stage tile only the days that are multiples of K, into
                    // rows [(day / K) * 5 + storage]; the powers are not kept
                    const int cK = A.ckptK;
                    const auto rall = __builtin_amdgcn_make_buffer_rsrc(o.traj, 0, -1, 0x00020000);
                    auto putc = [&](int day, int k, float v) {
                        const unsigned so = (unsigned)(((int64_t)(day / cK) * 5 + k) * row_bytes);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rall, voff, so, 0);
                    };
                    for (int tt = w; tt < Kt; tt += NRD) {
                        if (tt < ntA && (tA * Kt + tt) % cK == 0) {
                            putc(tA * Kt + tt, 0, bufA[tt * 256 + 128]);
                            putc(tA * Kt + tt, 1, bufA[tt * 256 + 192]);
                        }
                        if (tt < ntB && (tB * Kt + tt) % cK == 0) putc(tB * Kt + tt, 2, (bufB + (tt * OBR + NFB) * 64)[0]);
                        if (tt < ntC && (tC * Kt + tt) % cK == 0) {
                            putc(tC * Kt + tt, 3, bufC[tt * 448 + 320]);
                            putc(tC * Kt + tt, 4, bufC[tt * 448 + 384]);
                        }
                    }
                    PIPE_BARRIER();
                    continue;
                }
                for (int tt = w; tt < Kt; tt += NRD) {
                    const unsigned soff = (unsigned)tt * row_bytes;
                    float a0, a1, b0v, b1 = 0.0f, b2 = 0.0f, c0, c1;
                    if (tt < ntA) { a0 = bufA[tt * 256 + 128]; a1 = bufA[tt * 256 + 192]; }
                    if (tt < ntB) {
                        const float *rb = bufB + (tt * OBR + NFB) * 64;
                        b0v = rb[0];
                        if (SAVE_POW) { b1 = rb[64]; b2 = rb[128]; }
                    }
                    if (tt < ntC) { c0 = bufC[tt * 448 + 320]; c1 = bufC[tt * 448 + 384]; }
                    if (tt < ntA) { put(rSP, soff, a0); put(rMW, soff, a1); }
                    if (tt < ntB) {
                        put(rSM, soff, b0v);
                        if (SAVE_POW && !ADJ) { put(rSW, soff, b1); put(rEF, soff, b2); }   // the implicit scheme keeps no powers
                    }
                    if (tt < ntC) { put(rSUZ, soff, c0); put(rSLZ, soff, c1); }
                }
                PIPE_BARRIER();
            }
        } else {
            // reducers: waves 4, 8, 12 (next to the snow wave) and 6, 10, 14 (next to groundwater);
            // with dynamic parameters 4 and 6 are fillers and the reducers start at wave 8
            // with few dynamic parameters waves 4 and 6 are fillers and only four reducers are left for the six passes
            // of a tile (two of them ran 385-439 of 489 cycles per day): the three waves on the soil wave's SIMD
            // (5, 9, 13), idle otherwise, take the last -- lightest -- passes; the soil wave has the slack (309 of 489)
            const bool r1 = PIPE_Q1 == 1 && DYN && !MANY;
            const int n02 = ((nw - rbase + 3) >> 2) + ((nw - rbase - 2 + 3) >> 2);
            const int w = quad == 1 ? n02 + ((wave - 5) >> 2) : ((wave - rbase) >> 2) * 2 + (quad == 2 ? 1 : 0);
            const int NDR = n02 + (r1 ? (nw - 5 + 3) >> 2 : 0);
            PIPE_BARRIER();
            for (int it = 0; it < nIt; it++) {
                const int tA = it - 1, tB = it - 2, tC = CAP ? it - 2 : it - 3;
                const int ntA = (tA >= 0 && tA < nT) ? tile_nt(tA) : 0;
                const int ntB = (tB >= 0 && tB < nT) ? tile_nt(tB) : 0;
                const int ntC = (tC >= 0 && tC < nT) ? tile_nt(tC) : 0;
                const float *bufA = lds + P.oa + (tA & 1) * Kt * 256;
                const float *bufB = lds + P.ob + (tB & 1) * Kt * OBR * 64;
                const float *bufC = lds + P.oc + (tC & 1) * Kt * 448;
                constexpr int NFC = ADJ ? 1 : 5;   // implicit scheme: Q only (hbv_adj.py:309-317)
                const int iA = ADJ ? 0 : ntA * 2 * bpw, iB = ADJ ? 0 : ntB * NFB * bpw,
                          iC = (ADJ && !o.flux) ? 0 : ntC * NFC * bpw;
                const int pA = (iA + 63) >> 6, pB = (iB + 63) >> 6, pC = (iC + 63) >> 6;
                const int nR = pA + pB + pC;
                // heaviest passes first (C: 5 series), so the tail of the round-robin is light
                for (int u = w; u < nR; u += NDR) {
                    if (u < pC)
                        pipe_reduce_pass<7, NFC, fmapC>(d, o, bufC, tC * Kt, iC, u, lane, lgMp, b0, has_mu);
                    else if (u < pC + pB)
                        pipe_reduce_pass<OBR, NFB, fmapB>(d, o, bufB, tB * Kt, iB, u - pC, lane, lgMp, b0);
                    else
                        pipe_reduce_pass<4, 2, fmapA>(d, o, bufA, tA * Kt, iA, u - pC - pB, lane, lgMp, b0);
                }
                PIPE_BARRIER();
            }
        }
    }
}

} // namespace hbvx
