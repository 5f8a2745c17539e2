// lstm_seq.h -- the caller side of the hot path: the parameter network that
// feeds hbvx_forward is an LSTM over the same [T, B] grid (delta-MG's CudnnLstmModel; not part of the
// reference repository).  Once the HBV recurrence costs well under a millisecond the per-step
// launches of a library LSTM dominate a training step, so the recurrence is one persistent kernel
// per direction:
//
//   gates_t = gx_t + h_{t-1} W_hh^T          (gx = x W_ih^T + b_ih + b_hh: one library GEMM, host side)
//   i, f, o = sigmoid, g = tanh;  c_t = f c_{t-1} + i g;  h_t = o tanh(c_t)      (torch.nn.LSTM)
//
// Basins are independent, so the batch is cut into row tiles of 16 basins (the N of
// v_mfma_f32_16x16x4_f32: exact f32 products, a k-ordered fmaf chain) and only the H/16 workgroups
// that share a row tile ever talk to each other.  Forward: workgroup = 16 hidden units = four groups of
// 4 units x 4 gates (the 16 MFMA rows); wave w multiplies K quarter w of h_{t-1} (H/64 16-byte loads
// per lane) into all four groups (its W_hh entries live in H/4 VGPRs for the whole sequence), the
// partial tiles are summed through LDS in wave order and wave w finishes unit group w.  Backward: workgroup = 16 hidden units, its 4 waves
// split the 4H gate rows of dh_{t-1} = dgates_t W_hh (again H/4 VGPRs of W_hh each), partial tiles
// are summed through LDS in wave order, then one thread per (unit, basin) does the cell adjoint.
// Gate vectors use the (unit, gate) layout [T,B,H,4] so that a lane's four gates are one 16-byte access.
//
// Hand-off between the waves of a row tile, once per time step: the exchange buffer holds one slab per
// time step and row tile, never reused inside a launch and filled with the sentinel word 0xFFFFFFFF by
// k_lstm_arm (lstm.hip) before the launch.  A producer writes its slice with sc1 stores (write
// through, 4-byte granules are atomic) and goes on; a consumer re-issues its sc1 loads of the slab (L1
// bypassed: MI355X_MICROARCH.md "inter-workgroup visibility") until none of the words it needs is the
// sentinel -- the payload is its own flag, there are no counters, fences or barriers on the path, and a
// word that is not the sentinel can only be the value written in this launch.  h and the gate gradients
// are never the sentinel pattern (arithmetic NaNs are canonical 0x7FC00000).  One to three workgroups
// per CU (compiled for three waves per SIMD; the host sizes the LDS request so that exactly that many
// fit) and a launch never holds more workgroups than that, so all partners are resident.  Every spin
// is bounded: on a time-out the wave gives up, sets the error word (hbvx_lstm_check) and poisons its
// outputs.
//
// Measured and not kept (round 5, profiles/r05_lstm_xcd.jsonl; T = 730, B = 100, H = 256): a row tile's workgroups on
// ONE XCD (blocks b, b + 8, ... under the observed round-robin dispatch) with hand-off stores that stay in that XCD's L2
// (plain instead of sc1): forward 1.81 -> 1.67 ms, backward 2.28 -> 2.23; the mapping alone 1.71 / 2.66.  4 % of the
// pair for a protocol whose correctness would hang on placement (a run-time XCC_ID handshake could guard it): the
// hand-off's price sits in the consumer CU's memory queue, not in which cache serves it.
#pragma once
#include <hip/hip_runtime.h>

#define LSTM_ROWS 16                 // basins per row tile
#define LSTM_UNITS 16                // hidden units per workgroup
#define LSTM_SPIN_LIMIT (1u << 21)   // polls before a tile gives up (seconds, not minutes)
#define LSTM_SC1 16                  // buffer-instruction cache-policy bit: sc1

typedef float lstm_f4 __attribute__((ext_vector_type(4)));

struct LstmArgs {
    int T, B, tile0, ntile;          // this launch covers row tiles tile0 ..; ntile = tiles of the whole batch
    const float *w_hh;               // [4H,H], torch row order (i, f, g, o blocks)
    const float *gx;                 // forward: [T,B,H,4] input projection; backward: activated gates
    float *gates;                    // forward: activated gates (may alias gx); backward: grad of the pre-activations
    const float *c_in;               // backward: c_all
    const float *dh;                 // backward: grad_h [T,B,H]
    float *c_all, *h_all;            // forward outputs [T,B,H]
    float *xch;                      // exchange slabs, one per time step and row tile, sentinel-filled by k_lstm_arm
    unsigned *cnt;                   // cnt[0] = error word (zeroed by k_lstm_arm)
    unsigned spin_limit;             // polls before a hand-off gives up (LSTM_SPIN_LIMIT; lowered by the fault-injection test)
    int drop_wg;                     // fault injection (tests): this workgroup never publishes its slab; -1: none
};

// Gate non-linearities on the hardware transcendentals (v_exp_f32 / v_rcp_f32, about 1 ulp each): absolute
// error <= 2e-7 on values in [-1, 1], far inside the test tolerance, and 4-5 instructions instead of the
// ~30 of an IEEE division and the ~40 of ocml's tanhf -- the cell sits on the per-step latency chain.
// Saturation is exact: exp2 -> inf gives rcp -> 0.
__device__ __forceinline__ float lstm_sigmoid(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
__device__ __forceinline__ float lstm_tanh(float x)
{
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.88539008177792681f));
}

#define LSTM_SENTINEL 0xFFFFFFFFu
#define LSTM_AUX (LSTM_SC1 | 0x80000000)   // sc1 + volatile: the polling loads are never hoisted or merged

// Loads the wave's KB 16-byte granules per lane (granule g0 + 4 j + kq of row n) until none of them
// holds a sentinel word.  false: timed out (error word set).
template <int KB>
__device__ __forceinline__ bool lstm_fetch(const LstmArgs &a, const float *slab_ptr, int slab_bytes, int g0,
                                           int kq, int n, lstm_f4 (&v)[KB])
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(slab_ptr), 0, slab_bytes, 0x00020000);
    for (unsigned spins = 0;; ++spins) {
#pragma unroll
        for (int j = 0; j < KB; ++j)
            v[j] = __builtin_bit_cast(lstm_f4, __builtin_amdgcn_raw_buffer_load_b128(r, ((g0 + 4 * j + kq) * LSTM_ROWS + n) * 16, 0, LSTM_AUX));
        unsigned mx = 0;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const auto u = __builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v[j]);
            mx = max(max(mx, u[0]), u[1]);
            mx = max(max(mx, u[2]), u[3]);
        }
        if (__ballot(mx == LSTM_SENTINEL) == 0) return true;
        if (spins > a.spin_limit) {
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(&a.cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");
    }
}

// UG: unit groups (4 units x 4 gates = the 16 MFMA rows) per workgroup: 4 (16 units, H/16 workgroups per
// row tile) or, when the batch is small enough for twice the workgroups to fit one launch, 2 (8 units: half
// the MFMAs per wave on the latency chain; waves 2 and 3 only multiply).
template <int H, int UG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_lstm_fwd(LstmArgs a)
{
    constexpr int NWG = H / (4 * UG), KQ = H / 64;     // KQ: 16-byte loads per lane = the wave's quarter of h
    __shared__ lstm_f4 part[2][4][UG][64];             // [buffer][K quarter][unit group][lane]
    const int tile = a.tile0 + blockIdx.x / NWG, s = blockIdx.x % NWG;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, kq = l >> 4, n = l & 15;
    const bool fin = w < UG;                        // this wave finishes unit group w
    const int u0 = (s * UG + (fin ? w : 0)) * 4;    // its four units
    const int unit = u0 + kq;
    const int row = tile * LSTM_ROWS + n, rowc = row < a.B ? row : a.B - 1;
    const bool live = row < a.B && fin;

    // Wave w multiplies K quarter w of h_{t-1} into all UG groups, so no byte of the slab is loaded
    // twice per workgroup.  A operand of group g: MFMA row m = 4*unit_local + gate; k of block (j, i) =
    // 4 (w H/16 + 4 j + kq) + i, like the loads.
    float wreg[UG][KQ * 4];
    {
        const int m = l & 15;
#pragma unroll
        for (int g = 0; g < UG; ++g) {
            const float *wr = a.w_hh + (size_t)((m & 3) * H + (s * UG + g) * 4 + (m >> 2)) * H + 4 * (w * (H / 16) + kq);
#pragma unroll
            for (int j = 0; j < KQ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) wreg[g][j * 4 + i] = wr[16 * j + i];
        }
    }
    const size_t slab = (size_t)H * LSTM_ROWS;      // floats per (step, tile): [H/4][16 rows][4 units]
    // A hand-off time-out must never be silent: the wave that timed out raises this flag BEFORE the
    // step's barrier, every wave reads it AFTER the barrier, the finishing waves poison the last output
    // step and the whole workgroup leaves (a wave that simply returned would let the others sum stale
    // partial tiles into finite, wrong values).  Its row-tile partners time out on its missing slab.
    __shared__ int timed_out;
    if (threadIdx.x == 0) timed_out = 0;
    __syncthreads();
    const bool publish = (int)blockIdx.x != a.drop_wg;
    float c = 0.0f;
    for (int t = 0; t < a.T; ++t) {
        const size_t e = ((size_t)t * a.B + rowc) * H + unit;
        lstm_f4 acc = {0, 0, 0, 0};
        if (fin) acc = *reinterpret_cast<const lstm_f4 *>(a.gx + e * 4);
        if (t > 0) {
            lstm_f4 hv[KQ];
            if (!lstm_fetch<KQ>(a, a.xch + ((size_t)(t - 1) * a.ntile + tile) * slab, (int)(slab * 4), w * (H / 16), kq, n, hv))
                timed_out = 1;                      // every lane of the wave stores the same word
            lstm_f4 p[UG];
#pragma unroll
            for (int g = 0; g < UG; ++g) p[g] = lstm_f4{0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < KQ; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int g = 0; g < UG; ++g)
                        p[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[g][j * 4 + i], hv[j][i], p[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < UG; ++g) part[t & 1][w][g][l] = p[g];
            __syncthreads();                        // the only barrier of a step; part[] is two deep
            if (timed_out) {
                if (live) a.h_all[((size_t)(a.T - 1) * a.B + row) * H + unit] = __builtin_nanf("");
                return;
            }
            if (fin)
                acc += (part[t & 1][0][w][l] + part[t & 1][1][w][l]) + (part[t & 1][2][w][l] + part[t & 1][3][w][l]);
        }
        if (!fin) continue;
        const float ig = lstm_sigmoid(acc[0]), fg = lstm_sigmoid(acc[1]), gg = lstm_tanh(acc[2]), og = lstm_sigmoid(acc[3]);
        c = fg * c + ig * gg;
        const float h = og * lstm_tanh(c);
        if (live) {
            const lstm_f4 act = {ig, fg, gg, og};
            *reinterpret_cast<lstm_f4 *>(a.gates + e * 4) = act;
            a.c_all[e] = c;
            a.h_all[e] = h;
        }
        if (t + 1 < a.T && publish) {
            // the wave's 64 lanes cover 256 contiguous bytes: two whole lines, one store instruction
            float *xp = a.xch + ((size_t)t * a.ntile + tile) * slab + ((size_t)(u0 >> 2) * LSTM_ROWS + n) * 4 + kq;
            __hip_atomic_store(xp, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_lstm_bwd(LstmArgs a)
{
    constexpr int NWG = H / LSTM_UNITS, KB = H / 16;
    __shared__ float part[2][4][4][64];
    const int tile = a.tile0 + blockIdx.x / NWG, s = blockIdx.x % NWG;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, kq = l >> 4, n = l & 15;
    const int unit = s * LSTM_UNITS + 4 * kq + w;   // this thread's unit in the cell adjoint
    const int row = tile * LSTM_ROWS + n, rowc = row < a.B ? row : a.B - 1;
    const bool live = row < a.B;

    // A operand: MFMA row m = hidden unit 16 s + m; the wave's quarter of the gate rows:
    // k of block (j, i) = gate i of unit w H/4 + 4 j + kq
    float wreg[KB * 4];
    {
        const int m = l & 15;
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                wreg[j * 4 + i] = a.w_hh[(size_t)(i * H + w * (H / 4) + 4 * j + kq) * H + s * LSTM_UNITS + m];
    }
    const size_t slab = (size_t)H * LSTM_ROWS * 4;  // floats per (step, tile): [H units][16 rows][4 gates]
    __shared__ int timed_out;                       // as in k_lstm_fwd: raised before the barrier, acted on after it
    if (threadIdx.x == 0) timed_out = 0;
    __syncthreads();
    const bool publish = (int)blockIdx.x != a.drop_wg;
    float dc_carry = 0.0f;
    for (int t = a.T - 1; t >= 0; --t) {
        const size_t e = ((size_t)t * a.B + rowc) * H + unit;
        const lstm_f4 act = *reinterpret_cast<const lstm_f4 *>(a.gx + e * 4);
        const float ct = a.c_in[e];
        const float cp = t > 0 ? a.c_in[e - (size_t)a.B * H] : 0.0f;
        float dh = a.dh[e];
        if (t + 1 < a.T) {
            lstm_f4 gv[KB];
            if (!lstm_fetch<KB>(a, a.xch + ((size_t)(t + 1) * a.ntile + tile) * slab, (int)(slab * 4), w * (H / 4), kq, n, gv))
                timed_out = 1;
            lstm_f4 p0 = {0, 0, 0, 0}, p1 = {0, 0, 0, 0}, p2 = {0, 0, 0, 0}, p3 = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < KB; ++j) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float av = wreg[j * 4 + i], bv = gv[j][i];
                    if (i == 0) p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, p0, 0, 0, 0);
                    else if (i == 1) p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, p1, 0, 0, 0);
                    else if (i == 2) p2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, p2, 0, 0, 0);
                    else p3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, p3, 0, 0, 0);
                }
            }
            p0 = (p0 + p1) + (p2 + p3);
            // result row m = 4 kq + i of the 16 units: thread (w, l) finishes i = w
#pragma unroll
            for (int i = 0; i < 4; ++i) part[t & 1][w][i][l] = p0[i];
            __syncthreads();                        // the only barrier of a step; part[] is two deep
            if (timed_out) {
                // poison this step's gate gradients (nothing overwrites them: the workgroup leaves); every
                // weight gradient sums over them, so the failure reaches the loss scaler / optimiser as NaN
                if (live) {
                    const lstm_f4 bad = {__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
                    *reinterpret_cast<lstm_f4 *>(a.gates + e * 4) = bad;
                }
                return;
            }
            dh += (part[t & 1][0][w][l] + part[t & 1][1][w][l]) + (part[t & 1][2][w][l] + part[t & 1][3][w][l]);
        }
        const float ig = act[0], fg = act[1], gg = act[2], og = act[3];
        const float tc = lstm_tanh(ct);
        const float dc = dc_carry + dh * og * (1.0f - tc * tc);
        lstm_f4 d;
        d[0] = dc * gg * ig * (1.0f - ig);
        d[1] = dc * cp * fg * (1.0f - fg);
        d[2] = dc * ig * (1.0f - gg * gg);
        d[3] = dh * tc * og * (1.0f - og);
        dc_carry = dc * fg;
        if (live) *reinterpret_cast<lstm_f4 *>(a.gates + e * 4) = d;
        if (t > 0 && publish) {
            // 16 lanes of one store instruction cover a unit's 256-byte block: whole lines
            // (d comes out of VALU instructions: an inline-asm store placed straight after MFMAs would read the
            // accumulators before their write-back -- the compiler pads only the stores it can see)
            float *xp = a.xch + ((size_t)t * a.ntile + tile) * slab + ((size_t)unit * LSTM_ROWS + n) * 4;
            const lstm_f4 dx = live ? d : lstm_f4{0, 0, 0, 0};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(xp), "v"(dx) : "memory");
        }
    }
}
