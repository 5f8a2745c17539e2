// Gage routing of unit runoff for the hourly model (reference hbv_2_hourly.py:800-897):
// per (gage, unit) pair a normalised gamma unit hydrograph of L <= 72 taps, shifted by the
// fractional lag tau, applied to the area-weighted unit series and summed per gage.
//
// Data layout: qs [T,U] and out [T,G] are time-major (what the stepper writes / the caller reads),
// so one unit's series is a strided column: staging it for every (gage, unit) pair costs a
// 128-byte line per 4-byte element.  The host therefore transposes the runoff (and, in the
// backward, the incoming gradient) ONCE per call into caller scratch (k_transpose, LDS-tiled, both
// sides coalesced); the kernels stage contiguous [unit][t] / [gage][t] segments (256-step tile +
// 71-step halo) in LDS and run the 72-tap window out of LDS.  grad_qs is produced transposed and
// transposed back.  No atomics anywhere: sums over pairs and over time run in a fixed order,
// results are bit-reproducible.
// The FIR itself is pair-parallel (k_gage_lag_*: one block per pair and 1024-step tile, register-
// tiled), the sums over a gage's / unit's pairs are separate fixed-order kernels.  History at 4000
// units / 100 gages / 12 000 pairs / 2160 steps, forward / backward: per-gage loop with strided
// staging 0.44 / 0.78 ms; transposed staging 0.38 / 0.72 ms; a register-tiled FIR inside the
// per-gage loop was SLOWER (the serial loop over a gage's ~120 pairs with two barriers per pair was
// the bound, not LDS bandwidth); pair-parallel: see DESIGN.md.
#pragma once
#include "hbv_step.h"

namespace hbvx {

#define GAGE_TILE 256
#define GAGE_L HBVX_GAGE_MAXLEN

__device__ __forceinline__ int clampi_(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// out[c][r] = in[r][c]; in [R,C] row-major.  32x32 tiles through LDS (+1 pad), 256 threads.
__global__ void __launch_bounds__(256) k_transpose(int R, int C, const float *__restrict__ in, float *__restrict__ out)
{
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < R && c < C) ? in[(int64_t)r * C + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < C && r < R) out[(int64_t)c * R + r] = tile[tx][ty + 8 * k];
    }
}

struct GagePair {
    float a, b, tau, aa, theta, kk, f;
};

__device__ __forceinline__ GagePair gage_pair(const hbvx_gage_desc &r, int p)
{
    GagePair g;
    g.a = descale_(r.dp[p * 3 + 0], r.a_lo, r.a_hi);
    g.b = descale_(r.dp[p * 3 + 1], r.b_lo, r.b_hi);
    g.tau = descale_(r.dp[p * 3 + 2], r.tau_lo, r.tau_hi);
    g.aa = fmaxf(g.a, 0.0f) + 0.1f;
    g.theta = fmaxf(g.b, 0.0f) + 0.5f;
    g.kk = r.lag_uh ? floorf(g.tau) : 0.0f;
    g.f = r.lag_uh ? g.tau - g.kk : 0.0f;
    return g;
}

__device__ __forceinline__ float gamma_tap(const GagePair &g, float denom, int k)
{
    float t = (float)k + 0.5f;
    return 1.0f / denom * powf(t, g.aa - 1.0f) * expf(-t / g.theta);
}

// uh_gamma (uh_routing.py:5-22) + _frac_shift1d (hbv_2_hourly.py:857-897): one block per pair, one
// thread per tap (pow / exp in parallel), the normalising sum accumulated by one thread in tap order
// (the order the reference's sum and the oracle use), then the fractional shift out of LDS.
__global__ void __launch_bounds__(128) k_gage_uh(const hbvx_gage_desc r, float *__restrict__ uh)
{
    __shared__ float w[GAGE_L];
    __shared__ float ssum;
    const int p = blockIdx.x, k = threadIdx.x;
    const int L = r.L;
    GagePair g = gage_pair(r, p);
    const float denom = expf(lgammaf(g.aa)) * powf(g.theta, g.aa);
    if (k < L) w[k] = gamma_tap(g, denom, k);
    __syncthreads();
    if (k == 0) {
        float sum = 0.0f;
        for (int j = 0; j < L; j++) sum += w[j];
        ssum = sum;
    }
    __syncthreads();
    if (k >= L) return;
    const float sum = ssum;
    float *dst = uh + (int64_t)p * L;
    if (!r.lag_uh) {
        dst[k] = w[k] / sum;
        return;
    }
    const int kk = (int)g.kk;
    const int i0 = k - kk, i1 = k - kk - 1;
    const float w0 = (i0 >= 0 && i0 <= L - 1) ? w[i0] / sum : 0.0f;
    const float w1 = (i1 >= 0 && i1 <= L - 1) ? w[i1] / sum : 0.0f;
    dst[k] = (1.0f - g.f) * w0 + g.f * w1;
}

// Pair-parallel FIR: one block per (1024-step time tile, pair) -- no serial loop over the pairs of a gage / unit, 4
// outputs per thread.  The column lives in LDS as ALIGNED 16-byte chunks (thread tid's four outputs sit in chunk tid +
// nblk): four taps x four outputs = 16 FMAs per ONE ds_read_b128 -- consecutive lanes read consecutive chunks, so no
// padding and no bank conflicts -- and the window moves by renaming two chunk registers (until round 5: one scalar LDS
// read, a padded index computation, a scalar weight load and three register moves per tap for 4 FMAs).  Measured at
// 4 000 units x 2 160 hours: the forward routing call 0.246 -> 0.202 ms (profiles/r05_gage_fir.jsonl).  Taps beyond L (L rounded up to 4) carry weight 0.  Every output still accumulates its taps
// in ascending order with the same fused multiply-adds: results as before.  Writes the pair's series to lag[p][t]; the
// fixed-order sums over a gage's / unit's pairs are separate, trivially parallel kernels.
#define GAGE_TILE4 (4 * GAGE_TILE)
#define GAGE_COLN (GAGE_TILE4 + GAGE_L + 8)

// The pair's tap weights, zero-padded to a multiple of four, staged in LDS once per block: a wave-uniform 16-byte LDS
// read per four taps (a broadcast) instead of four dependent scalar loads from memory inside the tap loop.
__device__ __forceinline__ void gage_stage_weights(float *wsh, const float *__restrict__ w, int L)
{
    for (int k = threadIdx.x; k < GAGE_L + 4; k += GAGE_TILE) wsh[k] = (k < L) ? w[k] : 0.0f;
}

// lag[p][t] = sum_k uh[p,k] * qs[t-k,unit(p)] * areas[unit(p)]      (forward, causal)
__global__ void __launch_bounds__(GAGE_TILE) k_gage_lag_fwd(const hbvx_gage_desc r, const float *__restrict__ qsT,
                                                             const float *__restrict__ uh, float *__restrict__ lag)
{
    __shared__ __align__(16) float col[GAGE_COLN];
    __shared__ __align__(16) float wsh[GAGE_L + 4];
    const int p = blockIdx.x, t0 = blockIdx.y * GAGE_TILE4, tid = threadIdx.x;   // pairs on x: no 65535 limit
    const int T = r.T, L = r.L, nblk = (L + 3) >> 2;
    const int u = clampi_(r.pair_unit[p], 0, r.U - 1);
    const float ar = r.areas[u];
    const float *w = uh + (int64_t)p * L;
    // col[x] = q at time t0 - 4 nblk + x (zero history, zero beyond the record)
    for (int x = tid; x < GAGE_TILE4 + 4 * nblk; x += GAGE_TILE) {
        const int ts = t0 - 4 * nblk + x;
        col[x] = (ts >= 0 && ts < T) ? qsT[(int64_t)u * T + ts] * ar : 0.0f;
    }
    gage_stage_weights(wsh, w, L);
    __syncthreads();
    const float4 *c4 = reinterpret_cast<const float4 *>(col) + tid + nblk;       // the thread's own four outputs
    float4 cur = c4[0];
    float y0 = 0.0f, y1 = 0.0f, y2 = 0.0f, y3 = 0.0f;
    for (int b = 0; b < nblk; b++) {
        const float4 nxt = c4[-b - 1];                                           // the four steps before `cur`
        const float4 g = reinterpret_cast<const float4 *>(wsh)[b];              // taps 4 b .. 4 b + 3
        y0 += g.x * cur.x; y0 += g.y * nxt.w; y0 += g.z * nxt.z; y0 += g.w * nxt.y;
        y1 += g.x * cur.y; y1 += g.y * cur.x; y1 += g.z * nxt.w; y1 += g.w * nxt.z;
        y2 += g.x * cur.z; y2 += g.y * cur.y; y2 += g.z * cur.x; y2 += g.w * nxt.w;
        y3 += g.x * cur.w; y3 += g.y * cur.z; y3 += g.z * cur.y; y3 += g.w * cur.x;
        cur = nxt;
    }
    float *dst = lag + (int64_t)p * T + t0 + tid * 4;
    const int t = t0 + tid * 4;
    if (t < T) dst[0] = y0;
    if (t + 1 < T) dst[1] = y1;
    if (t + 2 < T) dst[2] = y2;
    if (t + 3 < T) dst[3] = y3;
}

// out[t,g] = (sum over the pairs of gage g, in CSR order, of lag[p][t]) / denom[g]
__global__ void __launch_bounds__(256) k_gage_sum_fwd(const hbvx_gage_desc r, const float *__restrict__ lag,
                                                       float *__restrict__ out)
{
    const int g = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= r.T) return;
    float acc = 0.0f;
    for (int p = r.gage_ptr[g]; p < r.gage_ptr[g + 1]; p++) acc += lag[(int64_t)p * r.T + t];
    out[(int64_t)t * r.G + g] = acc / r.denom[g];
}

// lag[p][t] = sum_k uh[p,k] * grad_out[t+k, gage(p)] / denom[gage(p)]      (backward, anti-causal; as k_gage_lag_fwd)
__global__ void __launch_bounds__(GAGE_TILE) k_gage_lag_bwd(const hbvx_gage_desc r, const float *__restrict__ uh,
                                                             const float *__restrict__ goT, float *__restrict__ lag)
{
    __shared__ __align__(16) float col[GAGE_COLN];
    __shared__ __align__(16) float wsh[GAGE_L + 4];
    const int p = blockIdx.x, t0 = blockIdx.y * GAGE_TILE4, tid = threadIdx.x;
    const int T = r.T, L = r.L, nblk = (L + 3) >> 2;
    const int g = clampi_(r.pair_gage[p], 0, r.G - 1);
    const float inv = 1.0f / r.denom[g];
    const float *w = uh + (int64_t)p * L;
    // col[x] = gradient at time t0 + x (zero beyond the record)
    for (int x = tid; x < GAGE_TILE4 + 4 * nblk + 4; x += GAGE_TILE) {
        const int ts = t0 + x;
        col[x] = (ts < T) ? goT[(int64_t)g * T + ts] * inv : 0.0f;
    }
    gage_stage_weights(wsh, w, L);
    __syncthreads();
    const float4 *c4 = reinterpret_cast<const float4 *>(col) + tid;
    float4 cur = c4[0];
    float y0 = 0.0f, y1 = 0.0f, y2 = 0.0f, y3 = 0.0f;
    for (int b = 0; b < nblk; b++) {
        const float4 nxt = c4[b + 1];                                            // the four steps after `cur`
        const float4 q = reinterpret_cast<const float4 *>(wsh)[b];
        y0 += q.x * cur.x; y0 += q.y * cur.y; y0 += q.z * cur.z; y0 += q.w * cur.w;
        y1 += q.x * cur.y; y1 += q.y * cur.z; y1 += q.z * cur.w; y1 += q.w * nxt.x;
        y2 += q.x * cur.z; y2 += q.y * cur.w; y2 += q.z * nxt.x; y2 += q.w * nxt.y;
        y3 += q.x * cur.w; y3 += q.y * nxt.x; y3 += q.z * nxt.y; y3 += q.w * nxt.z;
        cur = nxt;
    }
    float *dst = lag + (int64_t)p * T + t0 + tid * 4;
    const int t = t0 + tid * 4;
    if (t < T) dst[0] = y0;
    if (t + 1 < T) dst[1] = y1;
    if (t + 2 < T) dst[2] = y2;
    if (t + 3 < T) dst[3] = y3;
}

// grad_qsT[u][t] = areas[u] * (sum over the pairs of unit u, in CSR order, of lag[p][t])
__global__ void __launch_bounds__(256) k_gage_sum_bwd(const hbvx_gage_desc r, const float *__restrict__ lag,
                                                       float *__restrict__ gqsT)
{
    const int u = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= r.T) return;
    float acc = 0.0f;
    for (int i = r.unit_ptr[u]; i < r.unit_ptr[u + 1]; i++) {
        const int p = clampi_(r.unit_pairs[i], 0, r.NPAIR - 1);
        acc += lag[(int64_t)p * r.T + t];
    }
    gqsT[(int64_t)u * r.T + t] = acc * r.areas[u];
}

// Per pair: grad_uh[k] = sum_t gon[t] * qa[t-k] (gon = grad_out/denom of the pair's gage, qa the
// area-weighted unit series), then the chain shifted UH -> (taps, f) -> (route_a, route_b,
// route_tau) in closed form: for the normalised gamma taps w_k,
//   d w_k / d aa = w_k (ln t_k - sum_j w_j ln t_j),  d w_k / d theta = w_k (t_k - sum_j w_j t_j) / theta^2
// (the lgamma / theta^aa factor cancels in the normalisation, so no digamma is needed).
// Threads: 18 tap groups (4 taps each) x 14 contiguous time phases (76 steps) of a 1024-step tile.  Both columns live in
// LDS as aligned 16-byte chunks: per FOUR steps a thread reads one chunk of the gradient and one new chunk of the runoff
// (the previous one is kept) for 16 multiply-adds -- until round 5 one scalar read of each, an index computation and
// three register moves per step for 4.  Every tap still adds its steps in ascending order within a phase; phase
// partials are added in order (deterministic).
#define GAGE_PH 14
#define GAGE_PLEN 76                     // steps per phase: a multiple of 4, 14 x 76 >= 1024
__global__ void __launch_bounds__(GAGE_TILE) k_gage_bwd_p(const hbvx_gage_desc r, const float *__restrict__ qsT,
                                                           const float *__restrict__ goT, float *__restrict__ gdp)
{
    // qcol[x] = runoff at time t0 - HS + x with HS = H rounded up to a multiple of 4 (so that step j, tap 4 tg sits
    // on a chunk boundary); gcol[j] = gradient at time t0 + j, zero beyond the tile (phase 13 runs past 1024)
    __shared__ __align__(16) float qcol[GAGE_PH * GAGE_PLEN + GAGE_L + 8];
    __shared__ __align__(16) float gcol[GAGE_PH * GAGE_PLEN];
    __shared__ float part[GAGE_PH][GAGE_L];
    __shared__ float w[GAGE_L];
    __shared__ float guh[GAGE_L];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int T = r.T, U = r.U, G = r.G, L = r.L, H = L - 1, HS = (H + 3) & ~3;
    const int u = clampi_(r.pair_unit[p], 0, U - 1), g = clampi_(r.pair_gage[p], 0, G - 1);
    const float ar = r.areas[u], inv = 1.0f / r.denom[g];
    const int tg = tid % 18, phase = tid / 18;   // taps 4 tg .. 4 tg + 3; threads >= 252 only stage
    const int k0 = 4 * tg;
    const bool worker = phase < GAGE_PH;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int t0 = 0; t0 < T; t0 += GAGE_TILE4) {
        __syncthreads();
        for (int x = tid; x < GAGE_PH * GAGE_PLEN + HS; x += GAGE_TILE) {
            const int ts = t0 - HS + x;
            qcol[x] = (ts >= 0 && ts < T && x < GAGE_TILE4 + HS) ? qsT[(int64_t)u * T + ts] * ar : 0.0f;
        }
        for (int i = tid; i < GAGE_PH * GAGE_PLEN; i += GAGE_TILE)
            gcol[i] = (i < GAGE_TILE4 && t0 + i < T) ? goT[(int64_t)g * T + t0 + i] * inv : 0.0f;
        __syncthreads();
        if (worker && k0 <= HS) {
            // step j, tap k0 + q reads qa[t - k] at column j + HS - k0 - q: chunk (j + HS - k0) / 4 and the one before it
            const int j0 = phase * GAGE_PLEN;
            const float4 *q4 = reinterpret_cast<const float4 *>(qcol) + ((j0 + HS - k0) >> 2);
            const float4 *g4 = reinterpret_cast<const float4 *>(gcol) + (j0 >> 2);
            float4 prev = (j0 + HS - k0 >= 4) ? q4[-1] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll 2
            for (int c = 0; c < GAGE_PLEN / 4; c++) {
                const float4 cur = q4[c], gv = g4[c];
                // acc[q] += g[j + jj] * col[4 c' + jj - q], jj ascending
                acc[0] += gv.x * cur.x;  acc[0] += gv.y * cur.y;  acc[0] += gv.z * cur.z;  acc[0] += gv.w * cur.w;
                acc[1] += gv.x * prev.w; acc[1] += gv.y * cur.x;  acc[1] += gv.z * cur.y;  acc[1] += gv.w * cur.z;
                acc[2] += gv.x * prev.z; acc[2] += gv.y * prev.w; acc[2] += gv.z * cur.x;  acc[2] += gv.w * cur.y;
                acc[3] += gv.x * prev.y; acc[3] += gv.y * prev.z; acc[3] += gv.z * prev.w; acc[3] += gv.w * cur.x;
                prev = cur;
            }
        }
    }
    if (worker) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (k0 + q < GAGE_L) part[phase][k0 + q] = acc[q];
    }
    GagePair gp = gage_pair(r, p);
    if (tid < L) w[tid] = gamma_tap(gp, expf(lgammaf(gp.aa)) * powf(gp.theta, gp.aa), tid);
    __syncthreads();
    if (tid < L) {
        float sacc = 0.0f;
#pragma unroll
        for (int ph = 0; ph < GAGE_PH; ph++) sacc += part[ph][tid];
        guh[tid] = sacc;
    }
    __syncthreads();
    // Closed-form chain shifted UH -> (taps, lag fraction) -> (route_a, route_b, route_tau): one lane per
    // tap for the terms, thread 0 adds them in tap order (the sums the serial loop formed, bit for bit).
    __shared__ float t1[GAGE_L], t2[GAGE_L], t3[GAGE_L];
    __shared__ float s_sc[3];
    if (tid == 0) {
        float sum = 0.0f;
        for (int j = 0; j < L; j++) sum += w[j];
        s_sc[0] = sum;
    }
    __syncthreads();
    const bool tap = tid < L;
    const float tj = (float)tid + 0.5f;
    float wj = 0.0f, lt = 0.0f;
    if (tap) {
        wj = w[tid] / s_sc[0];
        lt = logf(tj);
        t1[tid] = wj * lt;
        t2[tid] = wj * tj;
    }
    __syncthreads();
    if (tid == 0) {
        float mlt = 0.0f, mt = 0.0f;
        for (int j = 0; j < L; j++) {
            mlt += t1[j];
            mt += t2[j];
        }
        s_sc[1] = mlt;
        s_sc[2] = mt;
    }
    __syncthreads();
    if (tap) {
        const float mlt = s_sc[1], mt = s_sc[2];
        const int kk = (int)gp.kk;
        // tap j feeds shifted taps j+kk (weight 1-f) and j+kk+1 (weight f)
        const int k0s = tid + kk, k1s = tid + kk + 1;
        const float g0 = (k0s <= L - 1) ? guh[k0s] : 0.0f, g1 = (k1s <= L - 1) ? guh[k1s] : 0.0f;
        const float gw = r.lag_uh ? (1.0f - gp.f) * g0 + gp.f * g1 : guh[tid];
        t3[tid] = wj * (g1 - g0);
        t1[tid] = gw * wj * (lt - mlt);
        t2[tid] = gw * wj * (tj - mt) / (gp.theta * gp.theta);
    }
    __syncthreads();
    if (tid != 0) return;
    float gaa = 0.0f, gth = 0.0f, gf = 0.0f;
    for (int j = 0; j < L; j++) {
        gf += t3[j];
        gaa += t1[j];
        gth += t2[j];
    }
    gdp[p * 3 + 0] = ((gp.a > 0.0f) ? gaa : 0.0f) * (r.a_hi - r.a_lo);
    gdp[p * 3 + 1] = ((gp.b > 0.0f) ? gth : 0.0f) * (r.b_hi - r.b_lo);
    gdp[p * 3 + 2] = (r.lag_uh ? gf : 0.0f) * (r.tau_hi - r.tau_lo);
}

} // namespace hbvx
