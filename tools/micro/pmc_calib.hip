// pmc_calib.hip -- known-byte-count kernels in the access shapes of the HBV kernels, to calibrate
// rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for them (MI355X_MICROARCH.md, HBM: only 16 B/lane
// streams are calibrated there; "calibrate on a known byte count in your own access pattern").
//
//   hipcc --offload-arch=gfx950 -O3 -o pmc_calib pmc_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -- ./pmc_calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/w -- ./pmc_calib
//
// Every kernel touches BYTES = 1 GiB exactly once (beyond the 256 MiB Infinity Cache); the program
// prints the byte count per kernel so the counter CSV can be divided by it
// (tools/pmc_calib_report.py -> profiles/r02_pmc_calibration.csv).
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// (a) the adjoint's trajectory reads: one wave per (64 lanes, 64-day chunk), seven [T][N] rows of
//     4 B per lane, one day ahead in registers (k_bwd_chunk_phi / k_bwd_chunk_sweep / k_bwd_stream)
__global__ void __launch_bounds__(64) calib_read_rows4(const float *p, int T, long N, int C, float *out)
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
        if (t > t0)
            for (int k = 0; k < 7; k++) nx[k] = p[k * S + (long)tn * N + n];
        float v = c[0];
        for (int k = 1; k < 7; k++) v = v * 0.5f + c[k];
        acc += v;
    }
    if (acc == 123.456f) out[n] = acc;   // keeps the loads alive, never stores
}

// (b) 16 B per lane streaming read (the guide's calibrated shape), same byte count
__global__ void __launch_bounds__(256) calib_read_vec16(const float4 *p, long n4, float *out)
{
    float acc = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = p[i];
        acc += (v.x + v.y) + (v.z + v.w);
    }
    if (acc == 123.456f) out[0] = acc;
}

// (c) 4 B per lane row stores, a wave writes 256 contiguous bytes per row and day (trajectory /
//     dynamic-parameter gradient writes)
__global__ void __launch_bounds__(64) calib_write_rows4(float *p, int T, long N, int C)
{
    const long n = (long)blockIdx.x * 64 + threadIdx.x;
    const int t0 = blockIdx.y * C, t1 = min(T, t0 + C);
    const long S = (long)T * N;
    for (int t = t0; t < t1; t++)
        for (int k = 0; k < 7; k++) p[k * S + (long)t * N + n] = (float)(t + k);
}

// (d) 16 B per lane streaming store (hbvx_zero's shape)
__global__ void __launch_bounds__(256) calib_write_vec16(float4 *p, long n4)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        p[i] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
}

// (e) the forward's flux stores: one lane in 16 stores 4 B (a wave writes 16 contiguous bytes per
//     series and day; consecutive waves -- on different XCDs -- write the neighbouring 16 bytes)
__global__ void __launch_bounds__(64) calib_write_leader(float *p, int T, long B, int NF)
{
    const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 4);
    if ((threadIdx.x & 15) != 0 || b >= B) return;
    for (int t = 0; t < T; t++)
        for (int k = 0; k < NF; k++) p[((long)k * T + t) * B + b] = (float)(t + k);
}

int main()
{
    const long BYTES = 1L << 30;
    float *buf, *out;
    CK(hipMalloc(&buf, BYTES));
    CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(buf, 0, BYTES));
    CK(hipDeviceSynchronize());
    // rows: 7 * T * N * 4 = 2^30  ->  T = 2048 days, N = 18724.57..: use N = 16384, T = 2340 -> not exact;
    // choose N = 16384 lanes, T = 2048 days, 8 "rows" would be exact; with 7 rows: 7*2048*16384*4 = 939 524 096
    const int T = 2048, C = 64;
    const long N = 16384;
    const long rows_bytes = 7L * T * N * 4;
    hipLaunchKernelGGL(calib_read_rows4, dim3(N / 64, T / C), dim3(64), 0, 0, buf, T, N, C, out);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(calib_read_vec16, dim3(4096), dim3(256), 0, 0, (const float4 *)buf, BYTES / 16, out);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(calib_write_rows4, dim3(N / 64, T / C), dim3(64), 0, 0, buf, T, N, C);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(calib_write_vec16, dim3(4096), dim3(256), 0, 0, (float4 *)buf, BYTES / 16);
    CK(hipDeviceSynchronize());
    const int TF = 730, NF = 12;
    const long BF = 12500;
    const long leader_bytes = (long)NF * TF * BF * 4;
    hipLaunchKernelGGL(calib_write_leader, dim3((BF + 3) / 4), dim3(64), 0, 0, buf, TF, BF, NF);
    CK(hipDeviceSynchronize());
    printf("calib_read_rows4 %ld\ncalib_read_vec16 %ld\ncalib_write_rows4 %ld\ncalib_write_vec16 %ld\ncalib_write_leader %ld\n",
           rows_bytes, BYTES, rows_bytes, BYTES, leader_bytes);
    return 0;
}
