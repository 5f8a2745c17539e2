// lds_dma.hip -- does buffer_load ... lds (LDS-DMA) land where one-wave workgroups expect it when
// several workgroups share a CU?  Each wave repeatedly DMA-loads rows of its own pattern (4-, 12- and
// 16-byte forms), waits, reads them back from LDS and counts mismatches.
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma lds_dma.hip && ./lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) k_dma(const float *src, int iters, int rowlen, unsigned *bad, int mode)
{
    __shared__ f4 l4[64];
    __shared__ float l3[64 * 4], l1[4][64];   // the 12-byte form writes lane i at i * 16, not i * 12
    const auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, -1, 0x00020000);
    const int ln = threadIdx.x;
    const long base = (long)blockIdx.x * 64;
    unsigned nbad = 0, n4 = 0, n3 = 0, n1 = 0;
    for (int it = 0; it < iters; it++) {
        const unsigned so = (unsigned)(((long)it * rowlen) * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)l4, 16, (unsigned)((base + ln) * 16), so * 4u, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)l3, 12, (unsigned)((base + ln) * 12), so * 3u, 0, 0);
        for (int k = 0; k < 4; k++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)l1[k], 4, (unsigned)((base + ln) * 4), so + k * 4096u, 0, 0);
        if (mode == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const f4 a = l4[ln];
        const float b0 = l3[ln * 4], b1 = l3[ln * 4 + 1], b2 = l3[ln * 4 + 2];
        float c[4];
        for (int k = 0; k < 4; k++) c[k] = l1[k][ln];
        // expected: src is filled so that src[i] = (float)(i % 65521)
        auto ex = [&](long byteoff) { return (float)((byteoff / 4) % 65521); };
        const long o4 = (base + ln) * 16 + (long)so * 4, o3 = (base + ln) * 12 + (long)so * 3, o1 = (base + ln) * 4 + so;
        n4 += a.x != ex(o4) || a.y != ex(o4 + 4) || a.z != ex(o4 + 8) || a.w != ex(o4 + 12);
        n3 += b0 != ex(o3) || b1 != ex(o3 + 4) || b2 != ex(o3 + 8);
        for (int k = 0; k < 4; k++) n1 += c[k] != ex(o1 + k * 4096);
        if (it == 3 && blockIdx.x == 1 && ln < 4) printf("lane %d x4 %g %g %g %g (want %g) x3 %g %g %g (want %g %g %g) d %g (want %g)\n", ln, a.x, a.y, a.z, a.w, ex(o4), b0, b1, b2, ex(o3), ex(o3+4), ex(o3+8), c[1], ex(o1+4096));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (n4) atomicAdd(bad, n4);
    if (n3) atomicAdd(bad + 1, n3);
    if (n1) atomicAdd(bad + 2, n1);
}

int main()
{
    const int blocks[] = {256, 2048, 4096};
    const long nfl = 1L << 28;   // 1 GiB of floats
    float *src;
    unsigned *bad;
    CK(hipMalloc(&src, nfl * 4));
    CK(hipMalloc(&bad, 12));
    float *h = (float *)malloc(nfl * 4);
    for (long i = 0; i < nfl; i++) h[i] = (float)(i % 65521);
    CK(hipMemcpy(src, h, nfl * 4, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 2; mode++)
        for (int b : blocks) {
            CK(hipMemset(bad, 0, 12));
            const int iters = 200, rowlen = 8192 * 64 / 16;   // keeps every offset inside the buffer
            hipLaunchKernelGGL(k_dma, dim3(b), dim3(64), 0, 0, src, iters, rowlen, bad, mode);
            CK(hipDeviceSynchronize());
            unsigned hb[3];
            CK(hipMemcpy(hb, bad, 12, hipMemcpyDeviceToHost));
            printf("mode %d (explicit vmcnt(0): %s)  blocks %5d (%.1f waves/SIMD): x4 %u x3 %u x1 %u mismatches (of %ld each)\n", mode,
                   mode ? "yes" : "compiler's", b, b / 1024.0, hb[0], hb[1], hb[2], (long)b * 64 * iters);
        }
    return 0;
}
