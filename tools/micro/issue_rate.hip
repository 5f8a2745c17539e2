// issue_rate.hip -- how fast does ONE wavefront issue fp32 VALU work on gfx950, alone on its SIMD
// and next to 1-3 others, with all 64 lanes or only lanes 0-31 enabled?  (Design input for the
// latency-bound time-steppers: DESIGN.md §4.)
//   hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int CHAINS>
__global__ void k_issue(float *out, long long *cyc, int iters, int half)
{
    float a[CHAINS];
    for (int k = 0; k < CHAINS; k++) a[k] = threadIdx.x * 1e-3f + k;
    const float m = 1.0001f, c = 0.5f;
    const bool on = !half || (threadIdx.x & 63) < 32;
    long long t0 = 0, t1 = 0;
    if (on) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int k = 0; k < CHAINS; k++) a[k] = __builtin_fmaf(a[k], m, c);
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    float s = 0;
    for (int k = 0; k < CHAINS; k++) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// the same with packed fp32 (v_pk_fma_f32: two floats per lane and instruction) -- what "two series per lane" would
// issue for the add / mul / fma part of a time step
typedef float v2f __attribute__((ext_vector_type(2)));
template <int CHAINS>
__global__ void k_issue_pk(float *out, long long *cyc, int iters)
{
    v2f a[CHAINS];
    for (int k = 0; k < CHAINS; k++) a[k] = (v2f){threadIdx.x * 1e-3f + k, threadIdx.x * 2e-3f - k};
    const v2f m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int k = 0; k < CHAINS; k++) a[k] = __builtin_elementwise_fma(a[k], m, c);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int k = 0; k < CHAINS; k++) s += a[k].x + a[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int CHAINS>
static int run_pk(int threads, const char *what)
{
    float *out;
    long long *cyc;
    const int blocks = 256, iters = 2000;
    CK(hipMalloc(&out, sizeof(float) * blocks * threads));
    CK(hipMalloc(&cyc, sizeof(long long) * blocks * 16));
    hipLaunchKernelGGL(k_issue_pk<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k_issue_pk<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    CK(hipDeviceSynchronize());
    long long h[256 * 16];
    CK(hipMemcpy(h, cyc, sizeof(long long) * blocks * (threads / 64), hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < blocks * (threads / 64); i++) mean += (double)h[i];
    mean /= blocks * (threads / 64);
    const double n = (double)iters * 8 * CHAINS;
    printf("%-34s chains %d  waves/SIMD %.2f  packed   : %.2f cycles per v_pk_fma per wave  (%.2f per SIMD-instr, %.2f per fma of a float)\n",
           what, CHAINS, threads / 256.0, mean / n, mean / n / (threads >= 256 ? threads / 256.0 : 1.0),
           mean / n / (threads >= 256 ? threads / 256.0 : 1.0) / 2.0);
    CK(hipFree(out));
    CK(hipFree(cyc));
    return 0;
}

template <int CHAINS>
static int run(int threads, int half, const char *what)
{
    float *out;
    long long *cyc;
    const int blocks = 256, iters = 2000;
    CK(hipMalloc(&out, sizeof(float) * blocks * threads));
    CK(hipMalloc(&cyc, sizeof(long long) * blocks * 16));
    hipLaunchKernelGGL(k_issue<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, half);
    hipLaunchKernelGGL(k_issue<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, half);
    CK(hipDeviceSynchronize());
    long long h[256 * 16];
    CK(hipMemcpy(h, cyc, sizeof(long long) * blocks * (threads / 64), hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < blocks * (threads / 64); i++) mean += (double)h[i];
    mean /= blocks * (threads / 64);
    const double n = (double)iters * 8 * CHAINS;
    printf("%-34s chains %d  waves/SIMD %.2f  lanes %2d : %.2f cycles per v_fma per wave  (%.2f per SIMD-instr)\n", what,
           CHAINS, threads / 256.0, half ? 32 : 64, mean / n, mean / n / (threads >= 256 ? threads / 256.0 : 1.0));
    CK(hipFree(out));
    CK(hipFree(cyc));
    return 0;
}

int main()
{
    for (int half = 0; half < 2; half++) {
        run<1>(64, half, "1 wave per CU, dependent chain");
        run<8>(64, half, "1 wave per CU, 8 independent");
        run<8>(256, half, "1 wave per SIMD, 8 independent");
        run<1>(512, half, "2 waves per SIMD, dependent");
        run<8>(512, half, "2 waves per SIMD, 8 independent");
        run<8>(768, half, "3 waves per SIMD, 8 independent");
        run<8>(1024, half, "4 waves per SIMD, 8 independent");
    }
    run_pk<1>(64, "1 wave per CU, dependent chain");
    run_pk<8>(256, "1 wave per SIMD, 8 independent");
    run_pk<4>(512, "2 waves per SIMD, 4 independent");
    run_pk<8>(512, "2 waves per SIMD, 8 independent");
    run_pk<8>(768, "3 waves per SIMD, 8 independent");
    run_pk<8>(1024, "4 waves per SIMD, 8 independent");
    return 0;
}
