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
// (Measured at 4000 units / 100 gages / 12 000 pairs / 2160 steps: transposition 0.44 -> 0.38 ms
// forward, 0.78 -> 0.72 ms backward.  A register-tiled FIR -- 4 outputs per thread, 1024-step
// tiles, 1 LDS read per 4 FMAs -- was slower both before and after: what remains is the serial
// loop over a gage's pairs with two barriers per pair, not LDS bandwidth.)
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

// uh_gamma (uh_routing.py:5-22) + _frac_shift1d (hbv_2_hourly.py:857-897): one thread per pair.
// The unshifted taps go to uh[p,:] first; the shift runs in place from the last tap down (tap k
// only reads taps <= k).
__global__ void __launch_bounds__(64) k_gage_uh(const hbvx_gage_desc r, float *__restrict__ uh)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= r.NPAIR) return;
    const int L = r.L;
    GagePair g = gage_pair(r, p);
    float denom = expf(lgammaf(g.aa)) * powf(g.theta, g.aa);
    float *w = uh + (int64_t)p * L;
    float sum = 0.0f;
    for (int k = 0; k < L; k++) {
        float v = gamma_tap(g, denom, k);
        w[k] = v;
        sum += v;
    }
    if (!r.lag_uh) {
        for (int k = 0; k < L; k++) w[k] = w[k] / sum;
        return;
    }
    const int kk = (int)g.kk;
    for (int k = L - 1; k >= 0; k--) {
        int i0 = k - kk, i1 = k - kk - 1;
        float w0 = (i0 >= 0 && i0 <= L - 1) ? w[i0] / sum : 0.0f;
        float w1 = (i1 >= 0 && i1 <= L - 1) ? w[i1] / sum : 0.0f;
        w[k] = (1.0f - g.f) * w0 + g.f * w1;
    }
}

// out[t,g] = (sum_{p in gage g} sum_k uh[p,k] * qs[t-k,unit(p)] * areas[unit(p)]) / denom[g]
// block = (time tile, gage); the unit column segment [t0-(L-1), t0+TILE) is staged in LDS.
__global__ void __launch_bounds__(GAGE_TILE) k_gage_fwd(const hbvx_gage_desc r, const float *__restrict__ qsT,
                                                         const float *__restrict__ uh, float *__restrict__ out)
{
    __shared__ float col[GAGE_TILE + GAGE_L];
    __shared__ float wl[GAGE_L];
    const int g = blockIdx.y, t0 = blockIdx.x * GAGE_TILE, tid = threadIdx.x, t = t0 + tid;
    const int T = r.T, U = r.U, L = r.L, H = L - 1;
    float acc = 0.0f;
    for (int p = r.gage_ptr[g]; p < r.gage_ptr[g + 1]; p++) {
        const int u = clampi_(r.pair_unit[p], 0, U - 1);
        const float ar = r.areas[u];
        __syncthreads();
        for (int i = tid; i < GAGE_TILE + H; i += GAGE_TILE) {
            int ts = t0 - H + i;
            col[i] = (ts >= 0 && ts < T) ? qsT[(int64_t)u * T + ts] * ar : 0.0f;
        }
        if (tid < L) wl[tid] = uh[(int64_t)p * L + tid];
        __syncthreads();
        float y = 0.0f;
        for (int k = 0; k < L; k++) y += wl[k] * col[tid + H - k]; // zero history: col is 0 for ts < 0
        acc += y;
    }
    if (t < T) out[(int64_t)t * r.G + g] = acc / r.denom[g];
}

// grad_qs[t,u] = areas[u] * sum_{p in unit u} sum_k uh[p,k] * grad_out[t+k, gage(p)] / denom[gage(p)]
__global__ void __launch_bounds__(GAGE_TILE) k_gage_bwd_q(const hbvx_gage_desc r, const float *__restrict__ uh,
                                                           const float *__restrict__ goT, float *__restrict__ gqsT)
{
    __shared__ float col[GAGE_TILE + GAGE_L];
    __shared__ float wl[GAGE_L];
    const int u = blockIdx.y, t0 = blockIdx.x * GAGE_TILE, tid = threadIdx.x, t = t0 + tid;
    const int T = r.T, G = r.G, L = r.L, H = L - 1;
    float acc = 0.0f;
    for (int i = r.unit_ptr[u]; i < r.unit_ptr[u + 1]; i++) {
        const int p = clampi_(r.unit_pairs[i], 0, r.NPAIR - 1);
        const int g = clampi_(r.pair_gage[p], 0, G - 1);
        const float inv = 1.0f / r.denom[g];
        __syncthreads();
        for (int j = tid; j < GAGE_TILE + H; j += GAGE_TILE) {
            int ts = t0 + j;
            col[j] = (ts < T) ? goT[(int64_t)g * T + ts] * inv : 0.0f;
        }
        if (tid < L) wl[tid] = uh[(int64_t)p * L + tid];
        __syncthreads();
        float y = 0.0f;
        for (int k = 0; k < L; k++) y += wl[k] * col[tid + k];
        acc += y;
    }
    if (t < T) gqsT[(int64_t)u * T + t] = acc * r.areas[u];
}

// Per pair: grad_uh[k] = sum_t gon[t] * qa[t-k] (gon = grad_out/denom of the pair's gage, qa the
// area-weighted unit series), then the chain shifted UH -> (taps, f) -> (route_a, route_b,
// route_tau) in closed form: for the normalised gamma taps w_k,
//   d w_k / d aa = w_k (ln t_k - sum_j w_j ln t_j),  d w_k / d theta = w_k (t_k - sum_j w_j t_j) / theta^2
// (the lgamma / theta^aa factor cancels in the normalisation, so no digamma is needed).
// Threads: 3 interleaved time phases x 72 taps accumulate, phase partials are added in order.
__global__ void __launch_bounds__(GAGE_TILE) k_gage_bwd_p(const hbvx_gage_desc r, const float *__restrict__ qsT,
                                                           const float *__restrict__ goT, float *__restrict__ gdp)
{
    __shared__ float qcol[GAGE_TILE + GAGE_L];
    __shared__ float gcol[GAGE_TILE];
    __shared__ float part[3][GAGE_L];
    __shared__ float w[GAGE_L];
    __shared__ float guh[GAGE_L];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int T = r.T, U = r.U, G = r.G, L = r.L, H = L - 1;
    const int u = clampi_(r.pair_unit[p], 0, U - 1), g = clampi_(r.pair_gage[p], 0, G - 1);
    const float ar = r.areas[u], inv = 1.0f / r.denom[g];
    const int k = tid % GAGE_L, phase = tid / GAGE_L; // phases 0..2 work, threads >= 216 only stage
    const bool worker = phase < 3 && k < L;
    float acc = 0.0f;
    for (int t0 = 0; t0 < T; t0 += GAGE_TILE) {
        __syncthreads();
        for (int i = tid; i < GAGE_TILE + H; i += GAGE_TILE) {
            int ts = t0 - H + i;
            qcol[i] = (ts >= 0 && ts < T) ? qsT[(int64_t)u * T + ts] * ar : 0.0f;
        }
        gcol[tid] = (t0 + tid < T) ? goT[(int64_t)g * T + t0 + tid] * inv : 0.0f;
        __syncthreads();
        if (worker)
            for (int j = phase; j < GAGE_TILE; j += 3) acc += gcol[j] * qcol[j + H - k];
    }
    if (worker) part[phase][k] = acc;
    GagePair gp = gage_pair(r, p);
    if (tid < L) w[tid] = gamma_tap(gp, expf(lgammaf(gp.aa)) * powf(gp.theta, gp.aa), tid);
    __syncthreads();
    if (tid < L) guh[tid] = (part[0][tid] + part[1][tid]) + part[2][tid];
    __syncthreads();
    if (tid != 0) return;
    float sum = 0.0f;
    for (int j = 0; j < L; j++) sum += w[j];
    float mlt = 0.0f, mt = 0.0f;
    for (int j = 0; j < L; j++) {
        float wj = w[j] / sum, tj = (float)j + 0.5f;
        mlt += wj * logf(tj);
        mt += wj * tj;
    }
    const int kk = (int)gp.kk;
    float gaa = 0.0f, gth = 0.0f, gf = 0.0f;
    for (int j = 0; j < L; j++) {
        float wj = w[j] / sum, tj = (float)j + 0.5f;
        // tap j feeds shifted taps j+kk (weight 1-f) and j+kk+1 (weight f)
        int k0 = j + kk, k1 = j + kk + 1;
        float g0 = (k0 <= L - 1) ? guh[k0] : 0.0f, g1 = (k1 <= L - 1) ? guh[k1] : 0.0f;
        float gw = r.lag_uh ? (1.0f - gp.f) * g0 + gp.f * g1 : guh[j];
        gf += wj * (g1 - g0);
        gaa += gw * wj * (logf(tj) - mlt);
        gth += gw * wj * (tj - mt) / (gp.theta * gp.theta);
    }
    gdp[p * 3 + 0] = ((gp.a > 0.0f) ? gaa : 0.0f) * (r.a_hi - r.a_lo);
    gdp[p * 3 + 1] = ((gp.b > 0.0f) ? gth : 0.0f) * (r.b_hi - r.b_lo);
    gdp[p * 3 + 2] = (r.lag_uh ? gf : 0.0f) * (r.tau_hi - r.tau_lo);
}

} // namespace hbvx
