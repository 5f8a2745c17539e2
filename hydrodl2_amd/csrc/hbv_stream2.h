// hbv_stream2.h -- streaming forward / adjoint for large grids, second generation.
//
// Measured on MI355X (profiles/r02_*): the first streaming kernels (hbv_stream.h) were bound by the
// NUMBER of vector-memory instructions, stores above all -- a wave-store costs the CU ~20 cycles of
// issue whatever its width (12 flux stores of 16 useful bytes cost as much as 7 trajectory stores of
// 256 bytes: 0.85 ms each of a 2.2 ms forward at 3 waves per SIMD) -- and by dependent-instruction
// latency at low occupancy (one wave alone issues a dependent fp32 chain at ~10 cycles per
// instruction, independent work at ~2.9; tools/micro/issue_rate.hip).  Hence:
//
//   * packed trajectory (HBVX_TRAJ_PACKED): per lane-day one 16-byte record (SP, MW, SM, SUZ) and one
//     4-byte SLZ row (HBVX_SAVE_POW builds: plus one 8-byte record of the two saved powers) -- 2 (3) stores
//     in the forward and 2 (3) loads in the adjoint instead of 5 + 5 (7 + 7), inside the same caller buffer;
//   * one flux store per day: after the ensemble butterflies every lane of a basin holds every mean;
//     member lane j keeps series j and the wave writes 12 series x 4 basins with ONE store;
//   * the three forcings of a basin in one 12-byte load when their channels are adjacent;
//   * the dynamic-parameter set is a template constant (the sets users run: none, {BETA, BETAET},
//     {BETA, K0, BETAET}); any other set of up to three stays on hbv_stream.h;
//   * the adjoint keeps one day of inputs in flight, accumulates the static-parameter gradients in
//     place, and is compiled for three or more waves per SIMD.
//
// Numerics: the same Step<> code and the same ensemble add tree as every other kernel family.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/hbvx.h"
#include "hbv_step.h"
#include "hbv_tiled.h"
#include "hbv_stream.h"

namespace hbvx {

// compile-time dynamic-parameter sets (slot lists in parameter order)
template <int SC> struct StreamDyn { static constexpr int nd = 0; };
template <> struct StreamDyn<1> { static constexpr int nd = 2; };
template <> struct StreamDyn<2> { static constexpr int nd = 3; };
// SC >= 3: ANY set of up to three dynamic parameters, as a wave-uniform run-time slot list (StreamArgs.nd / .dslot,
// slots in parameter order).  The parameter vector stays in registers: it is only ever indexed by constants, a
// run-time slot goes through a compare chain on a scalar (s2_put / s2_get: ~3 scalar + 1 vector instruction per
// candidate slot).  What the compile-time sets save over this form is exactly those chains (~50 vector instructions
// per day and parameter); what this form replaces is the first-generation kernel (hbv_stream.h): 5.1 ms where a
// compiled set runs 1.6 ms at 4 096 wavefronts.
template <> struct StreamDyn<3> { static constexpr int nd = 3; };
// SC == 4: the same for four to six parameters (more rows in flight and staged; above six the tiled forward and the
// time-parallel adjoint keep the problem: 15.2 ms against 6.4 for a two-parameter sibling at 4 096 wavefronts)
#define STREAM2_LIST_MAX 6
template <> struct StreamDyn<4> { static constexpr int nd = STREAM2_LIST_MAX; };
template <int SC>
__host__ __device__ constexpr int stream_slot(int k)
{
    return SC == 1 ? (k == 0 ? P_BETA : P_BETAET) : (k == 0 ? P_BETA : (k == 1 ? P_K0 : P_BETAET));
}

// Run-time slot -> a parameter array that must stay in registers (constant indices only).  `slot` is wave-uniform (a
// kernel argument), so this is a scalar branch tree ending in ONE vector move -- not a chain of NP compare + select
// pairs: on waves that issue one instruction every ~6-8 cycles whatever its kind, 16 selects per access cost the
// adjoint 45 % (profiles/r05_slotlist_ab.jsonl: 3.44 ms against the compiled set's 2.08 at 4 096 wavefronts).  The
// empty asm statement keeps the optimiser from turning the branches back into selects (a side effect cannot be
// speculated).
#define S2_SLOT_CASES(OP)                                                                                   \
    OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15) \
    OP(16) OP(17) OP(18)
static_assert(NPARAM_MAX <= 19, "S2_SLOT_CASES lists 19 slots");
template <int NP>
__device__ __forceinline__ void s2_put(float *p, int slot, float v)
{
    switch (slot) {
#define S2_PUT_CASE(i) case i: if (i < NP) { asm volatile("" ::: "memory"); p[i < NP ? i : 0] = v; } break;
        S2_SLOT_CASES(S2_PUT_CASE)
#undef S2_PUT_CASE
    default: break;
    }
}
template <int NP>
__device__ __forceinline__ float s2_get(const float *p, int slot)
{
    float r = 0.0f;
    switch (slot) {
#define S2_GET_CASE(i) case i: if (i < NP) { asm volatile("" ::: "memory"); r = p[i < NP ? i : 0]; } break;
        S2_SLOT_CASES(S2_GET_CASE)
#undef S2_GET_CASE
    default: break;
    }
    return r;
}

#ifndef STREAM2_D
#define STREAM2_D 2      // days of forward inputs in flight (register ring; 2 measured best: 1.77 vs 1.93 ms at config 5)
#endif
#ifndef FWPE
// four waves per SIMD (<= 128 VGPRs; the {BETA, K0, BETAET} instance needs 103 without spilling): a
// 12 500-basin share is 3 125 waves, more than the 3 072 slots of three per SIMD
#define FWPE __attribute__((amdgpu_waves_per_eu(4)))
#endif
#ifndef STREAM2_LDS_ACC
#define STREAM2_LDS_ACC 1   // adjoint: static-parameter gradient sums live in LDS, one owner per word (0: registers)
#endif
#ifndef STREAM2_ST_AUX
#define STREAM2_ST_AUX 0   // cache policy of the forward's trajectory / flux stores (0 default, 2 nt, 17 sc0 sc1)
#endif
#ifndef STREAM2_EXP
#define STREAM2_EXP 0    // dev experiments: 1 no flux store, 4 no trajectory stores, 8 / 16 / 32 no record / SLZ / powers store
#endif

typedef float s2_f2 __attribute__((ext_vector_type(2)));
typedef float s2_f3 __attribute__((ext_vector_type(3)));
typedef float s2_f4 __attribute__((ext_vector_type(4)));

struct S2Buf {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    typedef unsigned u3 __attribute__((ext_vector_type(3)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *base)
    {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, -1, 0x00020000);
    }
    static __device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so)
    {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
    }
    static __device__ __forceinline__ s2_f2 ld2(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so)
    {
        return __builtin_bit_cast(s2_f2, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0));
    }
    static __device__ __forceinline__ s2_f3 ld3(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so)
    {
        return __builtin_bit_cast(s2_f3, __builtin_amdgcn_raw_buffer_load_b96(r, vo, so, 0));
    }
    static __device__ __forceinline__ s2_f4 ld4(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so)
    {
        return __builtin_bit_cast(s2_f4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
    }
    static __device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so, float v)
    {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, vo, so, 0);
    }
    // streaming variants (trajectory / flux rows: written once, read by another kernel much later)
    static __device__ __forceinline__ void sts(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so, float v)
    {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, vo, so, STREAM2_ST_AUX);
    }
    static __device__ __forceinline__ void sts2(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so, s2_f2 v)
    {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, vo, so, STREAM2_ST_AUX);
    }
    static __device__ __forceinline__ void st2(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so, s2_f2 v)
    {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, vo, so, 0);
    }
    static __device__ __forceinline__ void st3(__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so, s2_f3 v)
    {
        __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u3, v), r, vo, so, 0);
    }
    // buffer load straight into LDS (LDS-DMA), no VGPRs: lane i's bytes land at lds + i * 4 (4-byte form)
    // or lds + i * 16 (12- and 16-byte forms: the 12-byte form leaves every fourth word untouched --
    // measured, tools/micro/lds_dma.hip)
    template <int BYTES>
    static __device__ __forceinline__ void ld_lds(__amdgpu_buffer_rsrc_t r, void *lds, unsigned vo, unsigned so)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        auto *dst = (__attribute__((address_space(3))) void *)lds;
        if constexpr (BYTES == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 4, vo, so, 0, 0);
        else if constexpr (BYTES == 12) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 12, vo, so, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, vo, so, 0, 0);
#endif
    }
    // 16-byte store.  NOT the builtin: on gfx950 a buffer_store_dwordx4 whose soffset is an SGPR reads
    // its data registers late, and a VALU write to them in the very next issue slot corrupts the last
    // four lanes of every row of 16 (observed: `v_pk_mov_b32` right behind the store replaced the SUZ
    // word of the packed record; tests/test_gpu_parity.py::test_stream_forward_equals_tiled_forward).
    // LLVM's hazard recognizer covers >8-byte stores only when soffset is NOT a register, so the wait
    // state is issued here, inside the same asm block as the store (nothing can be scheduled between).
    // The compiler does not count this store in vmcnt: its waits only become more conservative
    // (in-order retirement: an uncounted younger operation never lets an older load be read early).
    static __device__ __forceinline__ u4 words(const void *base)
    {
        const unsigned long long b = (unsigned long long)base;
        const u4 w = {(unsigned)b, (unsigned)(b >> 32) & 0xFFFFu, 0xFFFFFFFFu, 0x00020000u};
        return w;
    }
    static __device__ __forceinline__ void st4(const u4 rsrc_words, unsigned vo, unsigned so, s2_f4 v)
    {
#if STREAM2_ST_AUX == 2
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1"
#elif STREAM2_ST_AUX == 17
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc0 sc1\n\ts_nop 1"
#else
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1"
#endif
                     :
                     : "v"(v), "v"(vo), "s"(rsrc_words), "s"(so)
                     : "memory");
    }
};

// XCD-aware wave -> basin-group map.  Workgroups are dealt round-robin over the 8 XCDs (block i runs
// on XCD i % 8: observed, not guaranteed -- it only matters for speed), each with its own L2.  A wave
// touches 16-byte pieces of the flux / forcing / gradient rows (4 basins x 4 bytes); with the plain
// map the 8 pieces of one 128-byte line are written by 8 different L2s and reach HBM as 8 masked
// partial writes.  Here XCD c owns the contiguous run of groups [c * per, (c + 1) * per), so the
// pieces of a line meet in one L2 and leave it as whole lines.  The grid is 8 * per blocks; groups
// past the end return at once.
struct S2Lane : LaneT {
    bool valid;    // the wave has at least one basin
    bool bvalid;   // this lane's basin exists (its member may be padding)
    int b0;        // first basin of the wave
};
__device__ __forceinline__ S2Lane s2_lane(const hbvx_desc &d, int lgMp, int per_xcd)
{
    S2Lane L;
    const int i = blockIdx.x;
    const int group = (i & 7) * per_xcd + (i >> 3);
    L.lane = threadIdx.x & 63;
    const int Mp = 1 << lgMp;
    L.jm = L.lane & (Mp - 1);
    L.bl = L.lane >> lgMp;
    const int b = group * (64 >> lgMp) + L.bl;
    L.b0 = group * (64 >> lgMp);
    L.valid = L.b0 < d.B;
    L.bvalid = b < d.B;
    L.active = (b < d.B) && (L.jm < d.M);
    L.b = b < d.B ? b : d.B - 1;
    L.j = L.jm < d.M ? L.jm : d.M - 1;
    L.n = (int64_t)L.b * d.M + L.j;
    return L;
}

// The same for workgroups of MW waves (k_fwd_stream2, MW > 1): the workgroup owns MW consecutive basin
// groups, the XCD map deals workgroups.
__device__ __forceinline__ S2Lane s2_lane_mw(const hbvx_desc &d, int lgMp, int per_xcd, int mw)
{
    S2Lane L;
    const int i = blockIdx.x;
    const int group = ((i & 7) * per_xcd + (i >> 3)) * mw + (int)(threadIdx.x >> 6);
    L.lane = threadIdx.x & 63;
    const int Mp = 1 << lgMp;
    L.jm = L.lane & (Mp - 1);
    L.bl = L.lane >> lgMp;
    const int b = group * (64 >> lgMp) + L.bl;
    L.b0 = group * (64 >> lgMp);
    L.valid = L.b0 < d.B;
    L.bvalid = b < d.B;
    L.active = (b < d.B) && (L.jm < d.M);
    L.b = b < d.B ? b : d.B - 1;
    L.j = L.jm < d.M ? L.jm : d.M - 1;
    L.n = (int64_t)L.b * d.M + L.j;
    return L;
}

// fp32 add into an LDS word this lane owns: plain read-modify-write (STREAM2_LDS_ACC 1) or the LDS
// atomic ds_add_f32 (2).  Measured at config 5: the atomic form took the adjoint from 2.9 to 13 ms.
__device__ __forceinline__ void s2_lds_add(float *w, float v)
{
#if defined(__HIP_DEVICE_COMPILE__) && STREAM2_LDS_ACC == 2
    __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)w, v, 0, 0, false);
#else
    *w += v;
#endif
}

// one of three forcing values by a wave-uniform channel number
__device__ __forceinline__ float s2_pick(const s2_f3 v, int ch)
{
    return ch == 0 ? v.x : (ch == 1 ? v.y : v.z);
}

// Ensemble sums of NF series over the 16 member lanes of a basin row, leaving series jm on member lane
// jm (the flux store wants exactly that).  A butterfly that adds every series on every lane costs 4 DPP
// adds per series (48) plus an 11-deep select to pick the lane's series; halving the number of live
// values at every stage costs 33, with the SAME add tree (lane pairs, quads, quad pairs, halves), so the
// result is bit-identical to ens_sum_dpp's:
//   stage 1  lanes l, l^1:  even lanes keep series 2i, odd lanes 2i+1 (2 selects + 1 DPP add)  12 -> 6
//   stage 2  lanes l, l^2:  the same on bit 1                                                   6 -> 3
//   stage 3  quads q, q^1:  one v_add_f32_dpp per input value -- row_ror brings the partner quad's
//            value, bank_mask confines the write to the quads that keep that series             3(+0) -> 2
//   stage 4  quads q, q^2:  likewise with row_ror:8                                             2 -> 1
// The DPP reads of stages 3 and 4 come straight after VALU writes of their sources; the two wait states
// that needs (ISA guide, manually inserted wait states) are the s_nop 1 at the head of each block -- the
// compiler does not look inside inline assembly.
template <int NF>
__device__ __forceinline__ float s2_ens_sum16(const float *f, bool b0, bool b1)
{
    float g[6], h[3];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const float x = f[2 * i], y = 2 * i + 1 < NF ? f[2 * i + 1] : 0.0f;
        g[i] = (b0 ? y : x) + dpp_<0xB1>(b0 ? x : y);          // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int j = 0; j < 3; j++)
        h[j] = (b1 ? g[2 * j + 1] : g[2 * j]) + dpp_<0x4E>(b1 ? g[2 * j] : g[2 * j + 1]);   // quad_perm [2,3,0,1]
    float p, r, v;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %4, %4 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xa"
        : "=&v"(p), "=&v"(r)
        : "v"(h[0]), "v"(h[1]), "v"(h[2]));
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc"
        : "=&v"(v)
        : "v"(p), "v"(r));
    return v;
}

// ---------------------------------------------------------------------------------------------
// forward.  TRJ: 0 nothing kept, 1 trajectory rows [5,T+1,N] + aux [2,T,N], 2 packed, 3 K-day
// checkpoints [ceil(T/K),5,N] (HBVX_TRAJ_CKPT; K a power of two).
// XVEC: forcing channels are {0,1,2} and a basin's three values are adjacent (one 12-byte load).
// ---------------------------------------------------------------------------------------------
// MW: waves per workgroup, 1 or 8.  With one wave per workgroup a wave's share of a flux row is 16 bytes
// (4 basins) and the 11-12 such pieces per day were the most expensive stores of the kernel: 0.44 GB of
// them cost 0.3 of 1.45 ms at the config-5 share, the 4.1 GB of trajectory 0.45 (builds without either).
// Eight waves (Mp = 16: 32 basins) collect STREAM2_FD days of flux in LDS and the workgroup stores whole
// 128-byte row segments, one barrier per STREAM2_FD days, double-buffered.
#ifndef STREAM2_FD
#define STREAM2_FD 8
#endif
// (the six-slot list of the capillary / hourly models needs more than the 128 registers of four waves per SIMD)
template <int MODEL, bool BETAET, int TRJ, int SC, bool XVEC, int MW = 1>
__global__ void __launch_bounds__(MW * 64)
__attribute__((amdgpu_waves_per_eu((SC == 4 && MW == 1 && (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY)) ? 3 : 4)))
k_fwd_stream2(const StreamArgs A)
{
    constexpr int NP = NParamT<MODEL, BETAET>::value;
    constexpr int NF = MODEL == MODEL_HBV10 ? 11 : 12;
    constexpr int ND = StreamDyn<SC>::nd;
    constexpr int D = STREAM2_D;
    const hbvx_desc &d = A.d;
    const hbvx_fwd_out &o = A.o;
    const int lgMp = A.lgMp;
    const S2Lane L = MW > 1 ? s2_lane_mw(d, lgMp, A.per_xcd, MW) : s2_lane(d, lgMp, A.per_xcd);
    if (MW == 1 && !L.valid) return;            // (with a barrier in the loop every wave of the workgroup stays)
    const bool live = L.valid;
    constexpr int FD = STREAM2_FD;
    constexpr int FROW = 4 * MW + 4;             // padded row: the 12 series of a basin land in different banks
    __shared__ float s_flux[MW > 1 ? 2 * FD * 12 * FROW : 1];      // [2][FD][12 series][4 MW basins (+ pad)]
    const int wv = MW > 1 ? (int)(threadIdx.x >> 6) : 0;
    const int T = d.T, B = d.B;
    const int64_t N = (int64_t)B * d.M;
    const bool raw = d.raw_sigmoid != 0;
    const float nz = d.nearzero, invM = 1.0f / (float)d.M;
    const float ac = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.ac[L.b] : 0.0f;
    const float elev = (MODEL == MODEL_HBV20 || MODEL == MODEL_HOURLY) ? d.elev[L.b] : 0.0f;
    const unsigned OOB = 0xFFFFFFFFu;

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

    // inputs
    const auto rx = S2Buf::rsrc(d.x);
    const unsigned xvo = (unsigned)(L.b * d.x_b_stride * 4);
    const unsigned xts = (unsigned)(d.x_t_stride * 4);
    const int cp = d.ch_prcp, ct = d.ch_tmean, ce = d.ch_pet;
    // dynamic rows: the descriptor is rebased on the day's row (64-bit scalar arithmetic), so only one
    // row of the [T, B, ny] parameter tensor has to fit a descriptor's 4 GiB, not the tensor
    const float *dbase[ND > 0 ? ND : 1];
    int64_t dts[ND > 0 ? ND : 1];
    unsigned dvo[ND > 0 ? ND : 1];
    // per lane: range and offset of the day's value, or (0, static value) where dy_drop masked the basin --
    // the reference's `dyn * (1 - mask) + static * mask` (hbv.py:246), no lane mask to keep in SGPRs
    float dlo[ND > 0 ? ND : 1], dsc[ND > 0 ? ND : 1];
    const int nd = SC >= 3 ? A.nd : ND;                 // SC >= 3: a run-time list of nd <= 3 slots (wave-uniform)
    int dsl[ND > 0 ? ND : 1];
#pragma unroll
    for (int k = 0; k < ND; k++) {
        dsl[k] = SC >= 3 ? A.dslot[k < nd ? k : 0] : stream_slot<SC>(k);
        const hbvx_param_src &ps = d.p[dsl[k]];
        dbase[k] = ps.dyn ? ps.dyn : d.x;
        dvo[k] = (unsigned)((L.b * ps.dyn_b_stride + L.j) * 4);
        dts[k] = ps.dyn_t_stride;
        const bool use = !(ps.drop && ps.drop[L.b]);
        dlo[k] = use ? ps.lo : (SC >= 3 ? s2_get<NP>(p, dsl[k]) : p[stream_slot<SC>(k)]);
        dsc[k] = use ? ps.hi - ps.lo : 0.0f;
    }
    // outputs
    const auto rflux = S2Buf::rsrc(o.flux);
    const auto rtraj = S2Buf::rsrc(o.traj), raux = S2Buf::rsrc(o.aux ? o.aux : o.traj);
    const float *const slz0 = o.traj + 4 * (int64_t)(T + 1) * N;   // packed layout: the SLZ rows
    unsigned tvo[5], avo[2];
#pragma unroll
    for (int k = 0; k < 5; k++) tvo[k] = (TRJ == 1 && L.active) ? (unsigned)((k * (int64_t)(T + 1) * N + L.n) * 4) : OOB;
#pragma unroll
    for (int k = 0; k < 2; k++) avo[k] = (TRJ == 1 && L.active && o.aux) ? (unsigned)((k * (int64_t)T * N + L.n) * 4) : OOB;
    const int ckK = TRJ == 3 ? HBVX_TRAJ_CKPT_DAYS(o.traj_layout) : 1;
    const int cklg = ckK == 16 ? 4 : (ckK == 8 ? 3 : 2);
    const unsigned cvo = (TRJ == 3 && L.active) ? (unsigned)(L.n * 4) : OOB;
    const unsigned pvo4 = (TRJ == 2 && L.active) ? (unsigned)(L.n * 16) : OOB;
    const unsigned pvo1 = (TRJ == 2 && L.active) ? (unsigned)(L.n * 4) : OOB;
    const unsigned pvo2 = (TRJ == 2 && L.active && o.aux) ? (unsigned)(L.n * 8) : OOB;
    const unsigned row4 = (unsigned)(N * 4);
    const unsigned fT = (unsigned)((int64_t)T * B * 4), fB = (unsigned)(B * 4);
    // flux, Mp = 16 (the usual nmul): s2_ens_sum16 leaves one series per member lane, one ds_bpermute hands the
    // value of (series k, basin q) to lane 4k + q, and lanes 0..4*NF-1 store -- every quad writes 16 contiguous
    // bytes (12 requests per day where one lane per series and basin would make 48 four-byte ones: the
    // address FIFO of the memory pipeline was full a fifth of the time).  Other Mp: the basin leader
    // stores the NF series one by one.
    const bool roles = MW > 1 ? true : lgMp == 4;   // (the eight-wave form is launched for Mp = 16 only)
    const int tl = L.lane;                          // as a target: series tl / 4 of basin tl % 4
    const int src_lane = (tl & 3) * 16 + (tl >> 2);  // who holds it: lane jm = series in basin row tl % 4
    const bool jb0 = (L.jm & 1) != 0, jb1 = (L.jm & 2) != 0;
    const bool all_act = __builtin_amdgcn_ballot_w64(L.active) == ~0ull;   // no padded member lanes, no basins past B
    const int tb = L.b0 + (tl & 3);
    const unsigned fvo_role = (roles && (tl >> 2) < NF && tb < B) ? (unsigned)(((int64_t)(tl >> 2) * T * B + tb) * 4) : OOB;
    const unsigned fvo_lead = (!roles && L.active && L.jm == 0) ? (unsigned)(L.b * 4) : OOB;

    float st[5];
#pragma unroll
    for (int k = 0; k < 5; k++) st[k] = d.state_in ? d.state_in[k * N + L.n] : 0.001f;

    float fx[D][3], dv[D][ND > 0 ? ND : 1];
    auto issue = [&](int t, int j) {
        const unsigned tc = (unsigned)min(t, T - 1);
        const unsigned so = tc * xts;
        if (XVEC) {
            const s2_f3 v = S2Buf::ld3(rx, xvo, so);
            fx[j][0] = v.x; fx[j][1] = v.y; fx[j][2] = v.z;
        } else {
            fx[j][0] = S2Buf::ld(rx, xvo, so + cp * 4);
            fx[j][1] = S2Buf::ld(rx, xvo, so + ct * 4);
            fx[j][2] = S2Buf::ld(rx, xvo, so + ce * 4);
        }
#pragma unroll
        for (int k = 0; k < ND; k++)
            if (SC < 3 || k < nd) dv[j][k] = S2Buf::ld(S2Buf::rsrc(dbase[k] + tc * dts[k]), dvo[k], 0);
    };
    // XVEC: a basin's three forcing values are adjacent and arrive as one 12-byte load; `ident`: already in
    // (prcp, tmean, pet) order, else three wave-uniform picks put them there (config key `variables`)
    const bool ident = cp == 0 && ct == 1 && ce == 2;
    auto day = [&](int t, int j) {
        Step<MODEL, BETAET> s;
        if (XVEC) {
            const s2_f3 v = {fx[j][0], fx[j][1], fx[j][2]};
            if (ident) { s.P = v.x; s.Tf = v.y; s.PET = v.z; }
            else { s.P = s2_pick(v, cp); s.Tf = s2_pick(v, ct); s.PET = s2_pick(v, ce); }
        } else {
            s.P = fx[j][0]; s.Tf = fx[j][1]; s.PET = fx[j][2];
        }
#pragma unroll
        for (int k = 0; k < ND; k++) {
            if (SC >= 3 && k >= nd) continue;
            const float u = raw ? sigmoid_dyn_(dv[j][k]) : dv[j][k];
            if (SC >= 3) s2_put<NP>(p, dsl[k], u * dsc[k] + dlo[k]);
            else p[stream_slot<SC>(k)] = u * dsc[k] + dlo[k];
        }
        s.SP = st[0]; s.MW = st[1]; s.SM = st[2]; s.SUZ = st[3]; s.SLZ = st[4];
        s.template fwd<false, true>(p, nz, ac, elev, 0.0f, 0.0f);
        if (TRJ == 1 && !(STREAM2_EXP & 4)) {
            const unsigned so = (unsigned)t * row4;
#pragma unroll
            for (int k = 0; k < 5; k++) S2Buf::st(rtraj, tvo[k], so, st[k]);
            if (SAVE_POW) {
                S2Buf::st(raux, avo[0], so, s.sw0);
                S2Buf::st(raux, avo[1], so, s.ef0);
            }
        }
        if (TRJ == 3) {
            if ((t & (ckK - 1)) == 0) {   // wave-uniform
                const auto rck = S2Buf::rsrc(o.traj + (int64_t)(t >> cklg) * 5 * N);   // this checkpoint's five rows
#pragma unroll
                for (int k = 0; k < 5; k++) S2Buf::st(rck, cvo, (unsigned)k * row4, st[k]);
            }
        }
        if (TRJ == 2 && !(STREAM2_EXP & 4)) {
            const s2_f4 rec = {st[0], st[1], st[2], st[3]};
            const s2_f2 pw = {s.sw0, s.ef0};
            // the day's rows through descriptors rebased on them (64-bit scalar arithmetic): one row
            // has to fit a descriptor's 4 GiB, not the trajectory
            const int64_t tN = (int64_t)t * N;
            if (!(STREAM2_EXP & 8)) S2Buf::st4(S2Buf::words(o.traj + tN * 4), pvo4, 0u, rec);
            if (!(STREAM2_EXP & 16)) S2Buf::sts(S2Buf::rsrc(slz0 + tN), pvo1, 0u, st[4]);
            if (SAVE_POW && !(STREAM2_EXP & 32)) S2Buf::sts2(S2Buf::rsrc(o.aux + tN * 2), pvo2, 0u, pw);
        }
        st[0] = s.SP3; st[1] = s.MW3; st[2] = s.SM4; st[3] = s.SUZ4; st[4] = s.SLZ2;
        const float act = L.active ? 1.0f : 0.0f;
        float f[HBVX_MAX_FLUX];
        f[HBVX_F_QSIM] = s.Q; f[HBVX_F_Q0] = s.Q0; f[HBVX_F_Q1] = s.Q1; f[HBVX_F_Q2] = s.Q2;
        f[HBVX_F_AET] = s.ET; f[HBVX_F_SWE] = s.SP3; f[HBVX_F_RECHARGE] = s.rech; f[HBVX_F_EXCS] = s.exc;
        f[HBVX_F_EVAPFACTOR] = s.ef; f[HBVX_F_TOSOIL] = s.tosoil; f[HBVX_F_PERC] = s.PERC;
        f[HBVX_F_CAPILLARY] = s.cap;
        if (!all_act) {
#pragma unroll
            for (int k = 0; k < NF; k++) f[k] *= act;
        }
        if (!roles) ens_sum_dpp<NF>(f, lgMp);
        const unsigned fso = (unsigned)t * fB;
        if (STREAM2_EXP & 1) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < NF; k++) acc += f[k];
            if (acc == 123.456f) S2Buf::st(rflux, fvo_role, fso, acc);
        } else if (MW > 1) {
            // member lane jm holds series jm of its basin row: one LDS word per (day, series, basin)
            const float v = s2_ens_sum16<NF>(f, jb0, jb1) * invM;
            const int slot = t % (2 * FD);             // [buffer][day of the tile] in one index
            if (L.jm < NF) s_flux[(slot * 12 + L.jm) * FROW + wv * 4 + (L.lane >> 4)] = v;
        } else if (roles) {
            float v = s2_ens_sum16<NF>(f, jb0, jb1);
            v = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane * 4, __builtin_bit_cast(int, v * invM)));
            S2Buf::sts(rflux, fvo_role, fso, v);
        } else {
            unsigned so = fso;
#pragma unroll
            for (int k = 0; k < NF; k++) {
                S2Buf::st(rflux, fvo_lead, so, f[k] * invM);
                so += fT;
            }
        }
    };

    if (MW == 1) {
#pragma unroll
        for (int j = 0; j < D; j++) issue(j, j);
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
    } else {
        static_assert(STREAM2_FD % STREAM2_D == 0, "a flux tile holds whole rounds of the input ring");
        if (live) {
#pragma unroll
            for (int j = 0; j < D; j++) issue(j, j);
        }
        const int wb0 = ((int)(blockIdx.x & 7) * A.per_xcd + (int)(blockIdx.x >> 3)) * (4 * MW);   // first basin of the workgroup
        for (int ta = 0; ta < T; ta += FD) {
            const int tb = min(T, ta + FD);
            if (live) {
                int t = ta;
                for (; t + D <= tb; t += D) {
#pragma unroll
                    for (int j = 0; j < D; j++) {
                        day(t + j, j);
                        issue(t + j + D, j);
                    }
                }
#pragma unroll
                for (int j = 0; j < D; j++)
                    if (t + j < tb) day(t + j, j);
            }
            __syncthreads();
            // the tile's rows: (day, series) pairs dealt over half-waves, 4 MW consecutive basins each
            const int col = threadIdx.x & (4 * MW - 1);
            const bool cok = wb0 + col < B;
            const unsigned cvo = (unsigned)((wb0 + col) * 4);
            const int nrow = (tb - ta) * NF, rstep = (MW * 64) / (4 * MW);
            for (int r = threadIdx.x / (4 * MW); r < nrow; r += rstep) {
                const int dd = r / NF, k = r - dd * NF;
                const float v = s_flux[(((ta + dd) % (2 * FD)) * 12 + k) * FROW + col];
                S2Buf::sts(rflux, cok ? cvo + (unsigned)(ta + dd) * fB + (unsigned)k * fT : OOB, 0u, v);
            }
        }
    }
    if (L.active) {
#pragma unroll
        for (int k = 0; k < 5; k++) o.state_out[k * N + L.n] = st[k];
        if (TRJ == 1) {
#pragma unroll
            for (int k = 0; k < 5; k++) o.traj[((int64_t)k * (T + 1) + T) * N + L.n] = st[k];
        }
        if (TRJ == 2) {
            float *rec = o.traj + ((int64_t)T * N + L.n) * 4;
            rec[0] = st[0]; rec[1] = st[1]; rec[2] = st[2]; rec[3] = st[3];
            o.traj[4 * (int64_t)(T + 1) * N + (int64_t)T * N + L.n] = st[4];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// adjoint: one serial pass, one wave per 64 lanes.  TRJ: 1 rows, 2 packed.
// ---------------------------------------------------------------------------------------------
#ifndef STREAM2_DB
#define STREAM2_DB 1     // days of adjoint inputs in flight
#endif
#ifndef STREAM2_BWD_WAVES
#define STREAM2_BWD_WAVES 3
#endif
#ifndef STREAM2_LDSPF
#define STREAM2_LDSPF 1  // adjoint: the next day's inputs travel HBM -> LDS (LDS-DMA) instead of waiting in 17-25 VGPRs
#endif


// waves per SIMD the adjoint is compiled for: three (<= 168 VGPRs) wherever that needs no spill; the
// hourly step and the capillary models with all twelve gradient series live keep two
template <int MODEL, bool GFULL, int SC = 0>
constexpr int s2_bwd_waves()
{
    return (MODEL == MODEL_HOURLY || ((MODEL == MODEL_HBV20 || MODEL == MODEL_HBV11P) && GFULL) ||
            (MODEL == MODEL_HBV20 && SC == 4)) ? 2 : STREAM2_BWD_WAVES;
}

// W4: compiled for four waves per SIMD (<= 128 VGPRs, a few spilled values) instead of three.  Slower
// per wave, but a grid that overflows the three-per-SIMD slots by a little (a 12 500-basin share: 3 125
// waves for 3 072 slots) otherwise pays a whole extra round for the overflow: measured 2.45 -> 2.15 ms
// there, 1.94 -> 2.20 ms at 3 072 waves.  The host picks per grid (launch_stream.hip).
template <int MODEL, bool BETAET, int TRJ, int SC, bool GFULL, bool XVEC, bool W4 = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W4 ? 4 : s2_bwd_waves<MODEL, GFULL, SC>())))
k_bwd_stream2(const StreamBwdArgs A)
{
    constexpr int NP = NParamT<MODEL, BETAET>::value;
    constexpr int NF = MODEL == MODEL_HBV10 ? 11 : 12;
    constexpr int NG = GFULL ? NF : 4;
    constexpr int ND = StreamDyn<SC>::nd;
    constexpr int D = STREAM2_DB;
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

    // p[]: physical values of the day; gacc[]: sum over days of dL/d(physical value) per slot --
    // in LDS (each lane owns its word: no contention, plain fp32 adds in day order) so that the
    // sixteen accumulators do not cost registers the step needs
    float p[NPARAM_MAX];
#if STREAM2_LDS_ACC
    __shared__ float s_acc[NP][64];
    float *const gacc_lds = &s_acc[0][threadIdx.x & 63];
#define S2_ACC_ADD(i, v) s2_lds_add(gacc_lds + (i) * 64, (v))
#define S2_ACC_GET(i) gacc_lds[(i) * 64]
#else
    float gacc[NP];
#define S2_ACC_ADD(i, v) gacc[i] += (v)
#define S2_ACC_GET(i) gacc[i]
#endif
#pragma unroll
    for (int i = 0; i < NPARAM_MAX; i++) p[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const hbvx_param_src &s = d.p[i];
        float v = s.sta[(int64_t)L.b * s.sta_b_stride + L.j];
        v = raw ? sigmoid_(v) : v;
        p[i] = descale_(v, s.lo, s.hi);
#if STREAM2_LDS_ACC
        gacc_lds[i * 64] = 0.0f;
#else
        gacc[i] = 0.0f;
#endif
    }

    const auto rx = S2Buf::rsrc(d.x), rtraj = S2Buf::rsrc(io.traj), raux = S2Buf::rsrc(SAVE_POW ? io.aux : io.traj);
    const float *const slz0 = io.traj + 4 * (int64_t)(T + 1) * N;   // packed layout: the SLZ rows
    const auto rgf = S2Buf::rsrc(io.grad_flux ? io.grad_flux : io.grad_flux4);
    const auto rg4 = S2Buf::rsrc(io.grad_flux4 ? io.grad_flux4 : io.grad_flux);
    const bool has_gf = io.grad_flux != nullptr, has_g4 = io.grad_flux4 != nullptr;
    const unsigned xvo = (unsigned)(L.b * d.x_b_stride * 4), xts = (unsigned)(d.x_t_stride * 4);
    const int cp = d.ch_prcp, ct = d.ch_tmean, ce = d.ch_pet;
    unsigned tvo[5], avo[2];
#pragma unroll
    for (int k = 0; k < 5; k++) tvo[k] = (unsigned)((k * (int64_t)(T + 1) * N + L.n) * 4);
#pragma unroll
    for (int k = 0; k < 2; k++) avo[k] = (unsigned)((k * (int64_t)T * N + L.n) * 4);
    const unsigned pvo4 = (unsigned)(L.n * 16), pvo1 = (unsigned)(L.n * 4), pvo2 = (unsigned)(L.n * 8);
    const unsigned row4 = (unsigned)(N * 4);
    const unsigned gvo = (unsigned)(L.b * 4);
    const unsigned fT = (unsigned)((int64_t)T * B * 4), fB = (unsigned)(B * 4);

    // dynamic rows and their gradient rows: descriptors rebased per day (see the forward)
    const float *dbase[ND > 0 ? ND : 1];
    float *gdbase[ND > 0 ? ND : 1];
    int64_t dts[ND > 0 ? ND : 1], gdts[ND > 0 ? ND : 1];
    unsigned dvo[ND > 0 ? ND : 1], gdvo[ND > 0 ? ND : 1];
    // the adjoint is short of VGPRs, not SGPRs: range / offset stay wave-uniform, the dy_drop mask a lane mask
    float dlo[ND > 0 ? ND : 1], dsc[ND > 0 ? ND : 1], dsta[ND > 0 ? ND : 1];
    bool duse[ND > 0 ? ND : 1], dmasked[ND > 0 ? ND : 1];     // dmasked: the slot has a dy_drop mask at all (wave-uniform)
    const int nd = SC >= 3 ? A.nd : ND;                 // SC >= 3: a run-time list of nd <= 3 slots (wave-uniform)
    int dsl[ND > 0 ? ND : 1];
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

    // gf: the gradient of the flux series (or 0); g4: the routing adjoint's share of the four runoff series, added where
    // the day uses it (GF below) and only if `add4` -- never where the loads are issued: an add there made the compiler
    // wait for all of the day's loads on the spot (hbv_chunked.h::chunk_issue, profiles/r04_ab_chunk_prefetch.txt)
    struct In {
        float fx[3], st[5], ax[2], gf[NG], g4[4], dv[ND > 0 ? ND : 1];
        bool add4;
    };
    auto issue = [&](int t, In &I) {
        const unsigned tc = (unsigned)max(t, 0);
        const unsigned so = tc * xts, sr = tc * row4, sg = tc * fB;
        if (XVEC) {
            const s2_f3 v = S2Buf::ld3(rx, xvo, so);
            I.fx[0] = v.x; I.fx[1] = v.y; I.fx[2] = v.z;
        } else {
            I.fx[0] = S2Buf::ld(rx, xvo, so + cp * 4);
            I.fx[1] = S2Buf::ld(rx, xvo, so + ct * 4);
            I.fx[2] = S2Buf::ld(rx, xvo, so + ce * 4);
        }
        if (TRJ == 2) {
            const int64_t tN = (int64_t)tc * N;   // descriptors rebased on the day's rows (see the forward)
            const s2_f4 rec = S2Buf::ld4(S2Buf::rsrc(io.traj + tN * 4), pvo4, 0u);
            I.st[0] = rec.x; I.st[1] = rec.y; I.st[2] = rec.z; I.st[3] = rec.w;
            I.st[4] = S2Buf::ld(S2Buf::rsrc(slz0 + tN), pvo1, 0u);
            if (SAVE_POW) {
                const s2_f2 pw = S2Buf::ld2(S2Buf::rsrc(io.aux + tN * 2), pvo2, 0u);
                I.ax[0] = pw.x; I.ax[1] = pw.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 5; k++) I.st[k] = S2Buf::ld(rtraj, tvo[k], sr);
            if (SAVE_POW) { I.ax[0] = S2Buf::ld(raux, avo[0], sr); I.ax[1] = S2Buf::ld(raux, avo[1], sr); }
        }
        if (!SAVE_POW) I.ax[0] = I.ax[1] = 0.0f;
#pragma unroll
        for (int k = 0; k < NG; k++) {
            I.gf[k] = 0.0f;
            if (GFULL) { if (has_gf) I.gf[k] = S2Buf::ld(rgf, gvo, sg + (unsigned)k * fT); }
            if (k < 4) I.g4[k] = has_g4 ? S2Buf::ld(rg4, gvo, sg + (unsigned)k * fT) : 0.0f;
        }
        I.add4 = has_g4;
#pragma unroll
        for (int k = 0; k < ND; k++)
            if (SC < 3 || k < nd) I.dv[k] = S2Buf::ld(S2Buf::rsrc(dbase[k] + tc * dts[k]), dvo[k], 0);
    };
    const bool ident = cp == 0 && ct == 1 && ce == 2;   // (see the forward)
    auto day = [&](int t, const In &I) {
        Step<MODEL, BETAET> s;
        if (XVEC) {
            const s2_f3 v = {I.fx[0], I.fx[1], I.fx[2]};
            if (ident) { s.P = v.x; s.Tf = v.y; s.PET = v.z; }
            else { s.P = s2_pick(v, cp); s.Tf = s2_pick(v, ct); s.PET = s2_pick(v, ce); }
        } else {
            s.P = I.fx[0]; s.Tf = I.fx[1]; s.PET = I.fx[2];
        }
        s.SP = I.st[0]; s.MW = I.st[1]; s.SM = I.st[2]; s.SUZ = I.st[3]; s.SLZ = I.st[4];
        float ud[ND > 0 ? ND : 1];
#pragma unroll
        for (int k = 0; k < ND; k++) {
            ud[k] = 0.0f;
            if (SC >= 3 && k >= nd) continue;
            ud[k] = raw ? sigmoid_dyn_(I.dv[k]) : I.dv[k];
            const float pv = duse[k] ? ud[k] * dsc[k] + dlo[k] : dsta[k];
            if (SC >= 3) s2_put<NP>(p, dsl[k], pv);
            else p[stream_slot<SC>(k)] = pv;
        }
        s.template fwd<SAVE_POW>(p, nz, ac, elev, I.ax[0], I.ax[1]);
        FluxGrad g;
        auto GF = [&](int k) -> float {
            if (k >= NG) return 0.0f;
            const float v = (k < 4 && I.add4) ? I.gf[k] + I.g4[k < 4 ? k : 0] : I.gf[k];
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
        // static slots: physical-space sums (the range factor and sigmoid' are applied once, at the
        // end); a dynamic slot contributes to the static row only where dy_drop masked the basin
        if constexpr (SC >= 3) {
            // run-time slots: the day's gradient of each listed slot leaves through its row (and is taken out of the
            // static sum where the lane's basin uses the dynamic value), then every slot adds to its static sum
#pragma unroll
            for (int k = 0; k < ND; k++) {
                if (k >= nd) continue;
                const float gpk = s2_get<NP>(gp, dsl[k]);
                const float gu = gpk * dsc[k];
                const float gr = raw ? gu * (ud[k] * (1.0f - ud[k])) : gu;
                S2Buf::st(S2Buf::rsrc(gdbase[k] + t * gdts[k]), gdvo[k], 0, duse[k] ? gr : 0.0f);
                // with a dy_drop mask the masked basins' share stays in the static sum; without one (wave-uniform) the
                // slot's static sum is simply discarded at the end
                if (dmasked[k]) s2_put<NP>(gp, dsl[k], duse[k] ? 0.0f : gpk);
            }
#pragma unroll
            for (int i = 0; i < NP; i++) S2_ACC_ADD(i, gp[i]);
        } else {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            bool dyn_slot = false;
            int kd = 0;
#pragma unroll
            for (int k = 0; k < ND; k++)
                if (stream_slot<SC>(k) == i) { dyn_slot = true; kd = k; }
            if (!dyn_slot) {
                S2_ACC_ADD(i, gp[i]);
            } else {
                const float gu = gp[i] * dsc[kd];
                const float gr = raw ? gu * (ud[kd] * (1.0f - ud[kd])) : gu;
                S2Buf::st(S2Buf::rsrc(gdbase[kd] + t * gdts[kd]), gdvo[kd], 0, duse[kd] ? gr : 0.0f);
                S2_ACC_ADD(i, duse[kd] ? 0.0f : gp[i]);
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

    if (D == 1 && STREAM2_LDSPF) {
        // One day in flight, staged through LDS: rows of 64 lanes x (12 | 16 | 4 | 8 | 4...) bytes.  Per
        // day: wait for the DMA, pull the day's inputs into registers, re-arm the DMA for the day
        // before, compute.  Only ONE set of inputs occupies registers, and it dies as the step
        // consumes it.
        __shared__ s2_f4 l_rec[64];
        __shared__ float l_x[64 * 4], l_st[5][64], l_ax[SAVE_POW ? 2 * 64 : 1], l_gf[NG][64], l_dv[ND > 0 ? ND : 1][64];
        const int ln = threadIdx.x & 63;
        auto arm = [&](int t) {
            const unsigned tc = (unsigned)max(t, 0);
            const unsigned so = tc * xts, sr = tc * row4, sg = tc * fB;
            if (XVEC) {
                S2Buf::ld_lds<12>(rx, l_x, xvo, so);
            } else {
                S2Buf::ld_lds<4>(rx, l_x, xvo, so + cp * 4);
                S2Buf::ld_lds<4>(rx, l_x + 64, xvo, so + ct * 4);
                S2Buf::ld_lds<4>(rx, l_x + 128, xvo, so + ce * 4);
            }
            if (TRJ == 2) {
                const int64_t tN = (int64_t)tc * N;
                S2Buf::ld_lds<16>(S2Buf::rsrc(io.traj + tN * 4), l_rec, pvo4, 0u);
                S2Buf::ld_lds<4>(S2Buf::rsrc(slz0 + tN), l_st[4], pvo1, 0u);
                if (SAVE_POW) {
                    // 8-byte LDS-DMA does not exist: the two saved powers travel as two 4-byte rows
                    const auto rpw = S2Buf::rsrc(io.aux + tN * 2);
                    S2Buf::ld_lds<4>(rpw, l_ax, pvo2, 0u);
                    S2Buf::ld_lds<4>(rpw, l_ax + 64, pvo2 + 4u, 0u);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 5; k++) S2Buf::ld_lds<4>(rtraj, l_st[k], tvo[k], sr);
                if (SAVE_POW) {
                    S2Buf::ld_lds<4>(raux, l_ax, avo[0], sr);
                    S2Buf::ld_lds<4>(raux, l_ax + 64, avo[1], sr);
                }
            }
#pragma unroll
            for (int k = 0; k < NG; k++) {
                // one of the two gradient sources per series travels by DMA; if both exist the second
                // is added from a plain load in `pull`
                if (GFULL && has_gf) S2Buf::ld_lds<4>(rgf, l_gf[k], gvo, sg + (unsigned)k * fT);
                else if (k < 4 && has_g4) S2Buf::ld_lds<4>(rg4, l_gf[k], gvo, sg + (unsigned)k * fT);
            }
#pragma unroll
            for (int k = 0; k < ND; k++)
                if (SC < 3 || k < nd) S2Buf::ld_lds<4>(S2Buf::rsrc(dbase[k] + tc * dts[k]), l_dv[k], dvo[k], 0);
        };
        auto pull = [&](int t, In &I) {
            if (XVEC) {
                I.fx[0] = l_x[ln * 4]; I.fx[1] = l_x[ln * 4 + 1]; I.fx[2] = l_x[ln * 4 + 2];
            } else {
                I.fx[0] = l_x[ln]; I.fx[1] = l_x[64 + ln]; I.fx[2] = l_x[128 + ln];
            }
            if (TRJ == 2) {
                const s2_f4 rec = l_rec[ln];
                I.st[0] = rec.x; I.st[1] = rec.y; I.st[2] = rec.z; I.st[3] = rec.w;
                I.st[4] = l_st[4][ln];
            } else {
#pragma unroll
                for (int k = 0; k < 5; k++) I.st[k] = l_st[k][ln];
            }
            if (SAVE_POW) { I.ax[0] = l_ax[ln]; I.ax[1] = l_ax[64 + ln]; } else { I.ax[0] = I.ax[1] = 0.0f; }
            const unsigned sg = (unsigned)t * fB;
#pragma unroll
            for (int k = 0; k < NG; k++) {
                float v = 0.0f;
                if (GFULL && has_gf) {
                    v = l_gf[k][ln];
                    if (k < 4 && has_g4) v += S2Buf::ld(rg4, gvo, sg + (unsigned)k * fT);
                } else if (k < 4 && has_g4) {
                    v = l_gf[k][ln];
                }
                I.gf[k] = v;
                if (k < 4) I.g4[k] = 0.0f;
            }
            I.add4 = false;   // (this form has already summed the two sources above)
#pragma unroll
            for (int k = 0; k < ND; k++) I.dv[k] = (SC < 3 || k < nd) ? l_dv[k][ln] : 0.0f;
        };
        arm(T - 1);
        for (int t = T - 1; t >= 0; t--) {
            In cur;
            pull(t, cur);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS reads are done before the DMA rewrites the rows
            arm(t - 1);
            day(t, cur);
        }
    } else if (D == 1) {
        // one day in flight: the landed inputs move to `cur`, the next day's loads are issued, then
        // the day is computed
        In nxt;
        issue(T - 1, nxt);
        for (int t = T - 1; t >= 0; t--) {
            const In cur = nxt;
            issue(t - 1, nxt);
            day(t, cur);
        }
    } else {
        In ring[D];
#pragma unroll
        for (int j = 0; j < D; j++) issue(T - 1 - j, ring[j]);
        for (int t0 = T - 1; t0 >= 0; t0 -= D) {
#pragma unroll
            for (int j = 0; j < D; j++) {
                const int t = t0 - j;
                if (t >= 0) day(t, ring[j]);
                issue(t - D, ring[j]);
            }
        }
    }
    if (L.active) {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            if (!io.g[i].sta) continue;
            const hbvx_param_src &s = d.p[i];
            if (SC >= 3 && s.dyn && !s.drop) continue;     // a dynamic slot without a mask has no static share
            float gr = S2_ACC_GET(i) * (s.hi - s.lo);
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

#undef S2_ACC_ADD
#undef S2_ACC_GET

} // namespace hbvx
