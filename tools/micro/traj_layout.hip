// Micro-benchmark: does the trajectory layout limit the time-parallel adjoint's bandwidth?
// Same grid as k_bwd_chunk_phi at config 2 (168 x 115 one-wave blocks, 64 days each, one-day-ahead
// register prefetch); A: seven [T][N] rows of 4 B/lane (today); B: stage records
// [T][N][2] + [T][N][4] + [T][N][2] (8/16/8 B per lane).  Prints GB/s of each.
//   hipcc --offload-arch=gfx950 -O3 -o traj_layout traj_layout.hip && ./traj_layout
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(64) k_rows(const float *p, int T, long N, int C, float *out, int work)
{
    const long n = (long)blockIdx.x * 64 + threadIdx.x;
    const int t0 = blockIdx.y * C, t1 = min(T, t0 + C);
    const long S = (long)T * N;
    float acc = 0.0f, nx[7];
    for (int k = 0; k < 7; k++) nx[k] = p[k * S + (long)(t1 - 1) * N + n];
    for (int t = t1 - 1; t >= t0; t--) {
        float c[7];
        for (int k = 0; k < 7; k++) c[k] = nx[k];
        const int tn = max(t - 1, t0);
        for (int k = 0; k < 7; k++) nx[k] = p[k * S + (long)tn * N + n];
        float v = c[0];
        for (int k = 1; k < 7; k++) v = v * 0.5f + c[k];
        for (int i = 0; i < work; i++) v = v * 1.0001f + 0.5f;
        acc += v;
    }
    out[(long)blockIdx.y * gridDim.x * 64 + n] = acc;
}

__global__ void __launch_bounds__(64) k_recs(const float2 *a, const float4 *b, const float2 *c2, int T, long N, int C,
                                            float *out, int work)
{
    const long n = (long)blockIdx.x * 64 + threadIdx.x;
    const int t0 = blockIdx.y * C, t1 = min(T, t0 + C);
    float acc = 0.0f;
    float2 na = a[(long)(t1 - 1) * N + n], nc = c2[(long)(t1 - 1) * N + n];
    float4 nb = b[(long)(t1 - 1) * N + n];
    for (int t = t1 - 1; t >= t0; t--) {
        const float2 ca = na, cc = nc;
        const float4 cb = nb;
        const int tn = max(t - 1, t0);
        na = a[(long)tn * N + n]; nb = b[(long)tn * N + n]; nc = c2[(long)tn * N + n];
        float v = ca.x;
        v = v * 0.5f + ca.y; v = v * 0.5f + cb.x; v = v * 0.5f + cb.y; v = v * 0.5f + cb.z;
        v = v * 0.5f + cc.x; v = v * 0.5f + cc.y;
        for (int i = 0; i < work; i++) v = v * 1.0001f + 0.5f;
        acc += v;
    }
    out[(long)blockIdx.y * gridDim.x * 64 + n] = acc;
}

int main()
{
    const int T = 7300, C = 64;
    const long N = 10752;  // 168 waves
    const int nchunk = (T + C - 1) / C;
    float *p, *q, *out;
    CK(hipMalloc(&p, sizeof(float) * 7 * T * N));
    CK(hipMalloc(&q, sizeof(float) * 8 * T * N));
    CK(hipMalloc(&out, sizeof(float) * nchunk * N));
    CK(hipMemset(p, 0, sizeof(float) * 7 * T * N));
    CK(hipMemset(q, 0, sizeof(float) * 8 * T * N));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 grid(N / 64, nchunk);
    for (int work = 0; work <= 300; work += 100) {
        float msA = 0, msB = 0;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_rows, grid, dim3(64), 0, 0, p, T, N, C, out, work);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) msA += ms / 3;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_recs, grid, dim3(64), 0, 0, (const float2 *)q, (const float4 *)(q + 2 * (long)T * N),
                               (const float2 *)(q + 6 * (long)T * N), T, N, C, out, work);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) msB += ms / 3;
        }
        const double gbA = 7.0 * 4 * T * N / 1e9, gbB = 8.0 * 4 * T * N / 1e9;
        printf("work %3d  rows: %.3f ms %.0f GB/s   records: %.3f ms %.0f GB/s (28 useful B: %.0f GB/s)\n", work, msA,
               gbA / msA * 1e3, msB, gbB / msB * 1e3, gbA / msB * 1e3);
    }
    return 0;
}
