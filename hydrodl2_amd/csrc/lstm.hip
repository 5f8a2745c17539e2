// lstm.hip -- C ABI of the sequence LSTM (include/hbvx_lstm.h; kernels in lstm_seq.h).
#include "hbvx_host.h"
#include "lstm_seq.h"
#include "../../include/hbvx_lstm.h"

using namespace hbvx_host;

static int check_lstm(const hbvx_lstm_desc *d)
{
    if (!d) return fail(HBVX_E_NULL, "lstm desc is NULL");
    if (d->abi_version != HBVX_LSTM_ABI_VERSION) return fail(HBVX_E_ABI, "lstm abi_version mismatch");
    if (d->T <= 0 || d->B <= 0) return fail(HBVX_E_SHAPE, "lstm T/B out of range");
    if (d->H != 64 && d->H != 128 && d->H != 256)
        return fail(HBVX_E_UNSUPPORTED, "lstm hidden size must be 64, 128 or 256");
    return 0;
}

static uint64_t lstm_counter_bytes(const hbvx_lstm_desc *) { return 256; }   // the error word, on a line of its own

// exchange slabs of one direction: forward [T][tiles][H][16] floats, backward four gates per unit
static uint64_t lstm_slab_bytes(const hbvx_lstm_desc *d, bool backward)
{
    const uint64_t ntile = ((uint64_t)d->B + LSTM_ROWS - 1) / LSTM_ROWS;
    return (uint64_t)d->T * ntile * d->H * LSTM_ROWS * (backward ? 4 : 1) * sizeof(float);
}

// Arms the workspace of one launch: the error word's line zeroed, every exchange word the sentinel 0xFFFFFFFF
// (lstm_seq.h).  A KERNEL, not hipMemsetAsync: the launch sequence is captured into HIP graphs (examples/train_dpl.py
// --graph), and a captured memset node of this runtime was seen not to replay -- slabs that keep
// the previous replay's words would hand stale h to the consumers without any error.
typedef unsigned lstm_u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_lstm_arm(lstm_u4 *__restrict__ p, uint64_t n16, uint64_t zero16)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const unsigned v = i < zero16 ? 0u : 0xFFFFFFFFu;
        __builtin_nontemporal_store(lstm_u4{v, v, v, v}, p + i);
    }
}

static hipError_t lstm_prepare(const hbvx_lstm_desc *d, void *workspace, bool backward, hipStream_t st)
{
    const uint64_t n16 = (lstm_counter_bytes(d) + lstm_slab_bytes(d, backward)) / 16;   // both multiples of 16
    const uint64_t want = (n16 + 255) / 256;
    const unsigned grid = (unsigned)(want < 8192 ? (want ? want : 1) : 8192);
    hipLaunchKernelGGL(k_lstm_arm, dim3(grid), dim3(256), 0, st, (lstm_u4 *)workspace, n16, lstm_counter_bytes(d) / 16);
    return hipGetLastError();
}

extern "C" uint64_t hbvx_lstm_workspace_bytes(const hbvx_lstm_desc *d)
{
    if (!d || d->T <= 0 || d->B <= 0 || d->H <= 0) return 0;
    return lstm_counter_bytes(d) + lstm_slab_bytes(d, true);   // the backward slabs are the larger ones
}

static int lstm_cu_count()
{
    // per device (the caller makes the tensors' device current: hydrodl2_amd/ops.py::_device_guard)
    static int n_cu[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (n_cu[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu[dev] = v;
    }
    return n_cu[dev];
}

template <typename K>
static hipError_t launch_lstm(K kern, LstmArgs a, int nwg, hipStream_t st)
{
    const int n_cu = lstm_cu_count();
    // Residency: a launch never holds more workgroups than fit on the chip at once, so every partner
    // of a row tile is running.  The kernels are compiled for three waves per SIMD (<= 168 VGPRs), so up
    // to three workgroups share a CU; the LDS request is sized so that exactly `wpc` fit: a batch that
    // fits one launch at one workgroup per CU gets a CU per workgroup, larger batches interleave two or
    // three row tiles per SIMD, which hides one tile's hand-off latency behind the others' MFMAs.
    const int need = a.ntile * nwg;
    int wpc = env_int("HBVX_LSTM_WGS_PER_CU", (need + n_cu - 1) / n_cu);
    wpc = wpc < 1 ? 1 : (wpc > 3 ? 3 : wpc);
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern));
    if (e != hipSuccess) return e;
    const int total = wpc == 1 ? 81 * 1024 : (wpc == 2 ? 54 * 1024 + 512 : 41 * 1024);   // of 160 KB per CU
    const int lds = total > (int)fa.sharedSizeBytes ? total - (int)fa.sharedSizeBytes : 0;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    const int cap = n_cu * wpc / nwg > 0 ? n_cu * wpc / nwg : 1;       // row tiles one launch can hold
    const int launches = (a.ntile + cap - 1) / cap;
    const int per_launch = (a.ntile + launches - 1) / launches;        // balanced
    for (int t0 = 0; t0 < a.ntile; t0 += per_launch) {
        a.tile0 = t0;
        const int nt = a.ntile - t0 < per_launch ? a.ntile - t0 : per_launch;
        hipLaunchKernelGGL(kern, dim3(nt * nwg), dim3(256), lds, st, a);
    }
    return hipGetLastError();
}

extern "C" int hbvx_lstm_forward(const hbvx_lstm_desc *d, const float *w_hh, const float *gx, float *gates,
                                 float *c_all, float *h_all, void *workspace, uint64_t workspace_bytes,
                                 void *stream)
{
    int rc = check_lstm(d);
    if (rc) return rc;
    if (!w_hh || !gx || !gates || !c_all || !h_all) return fail(HBVX_E_NULL, "lstm buffer is NULL");
    if (!workspace || workspace_bytes < hbvx_lstm_workspace_bytes(d))
        return fail(HBVX_E_NULL, "lstm workspace missing or too small");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = lstm_prepare(d, workspace, false, st);
    if (e != hipSuccess) return hip_fail(e, "hbvx_lstm_forward arm");
    LstmArgs a{};
    a.T = d->T; a.B = d->B; a.ntile = (d->B + LSTM_ROWS - 1) / LSTM_ROWS;
    a.w_hh = w_hh; a.gx = gx; a.gates = gates; a.c_all = c_all; a.h_all = h_all;
    a.cnt = (unsigned *)workspace;
    a.xch = (float *)((char *)workspace + lstm_counter_bytes(d));
    a.spin_limit = (unsigned)env_int("HBVX_LSTM_SPIN_LIMIT", (int)LSTM_SPIN_LIMIT);
    a.drop_wg = env_int("HBVX_LSTM_DEBUG_DROP_WG", -1);
    // 8 units per workgroup while twice the workgroups still fit one launch at one per CU, else 16
    const int n_cu = lstm_cu_count();
    const bool small = env_int("HBVX_LSTM_UNITS", a.ntile * (d->H / 8) <= n_cu ? 8 : 16) == 8;
    if (small)
        e = d->H == 64 ? launch_lstm(k_lstm_fwd<64, 2>, a, 8, st)
          : d->H == 128 ? launch_lstm(k_lstm_fwd<128, 2>, a, 16, st) : launch_lstm(k_lstm_fwd<256, 2>, a, 32, st);
    else
        e = d->H == 64 ? launch_lstm(k_lstm_fwd<64, 4>, a, 4, st)
          : d->H == 128 ? launch_lstm(k_lstm_fwd<128, 4>, a, 8, st) : launch_lstm(k_lstm_fwd<256, 4>, a, 16, st);
    if (e != hipSuccess) return hip_fail(e, "hbvx_lstm_forward launch");
    return 0;
}

extern "C" int hbvx_lstm_backward(const hbvx_lstm_desc *d, const float *w_hh, const float *gates,
                                  const float *c_all, const float *grad_h, float *grad_gates,
                                  void *workspace, uint64_t workspace_bytes, void *stream)
{
    int rc = check_lstm(d);
    if (rc) return rc;
    if (!w_hh || !gates || !c_all || !grad_h || !grad_gates) return fail(HBVX_E_NULL, "lstm buffer is NULL");
    if (gates == grad_gates) return fail(HBVX_E_UNSUPPORTED, "lstm grad_gates must not alias gates");
    if (!workspace || workspace_bytes < hbvx_lstm_workspace_bytes(d))
        return fail(HBVX_E_NULL, "lstm workspace missing or too small");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = lstm_prepare(d, workspace, true, st);
    if (e != hipSuccess) return hip_fail(e, "hbvx_lstm_backward arm");
    LstmArgs a{};
    a.T = d->T; a.B = d->B; a.ntile = (d->B + LSTM_ROWS - 1) / LSTM_ROWS;
    a.w_hh = w_hh; a.gx = gates; a.gates = grad_gates; a.c_in = c_all; a.dh = grad_h;
    a.cnt = (unsigned *)workspace;
    a.xch = (float *)((char *)workspace + lstm_counter_bytes(d));
    a.spin_limit = (unsigned)env_int("HBVX_LSTM_SPIN_LIMIT", (int)LSTM_SPIN_LIMIT);
    a.drop_wg = env_int("HBVX_LSTM_DEBUG_DROP_WG", -1);
    e = d->H == 64 ? launch_lstm(k_lstm_bwd<64>, a, 4, st)
      : d->H == 128 ? launch_lstm(k_lstm_bwd<128>, a, 8, st) : launch_lstm(k_lstm_bwd<256>, a, 16, st);
    if (e != hipSuccess) return hip_fail(e, "hbvx_lstm_backward launch");
    return 0;
}

extern "C" int hbvx_lstm_check(const hbvx_lstm_desc *d, const void *workspace, void *stream)
{
    int rc = check_lstm(d);
    if (rc) return rc;
    if (!workspace) return fail(HBVX_E_NULL, "lstm workspace is NULL");
    unsigned word = 0;
    hipError_t e = hipMemcpyAsync(&word, (const unsigned *)workspace, sizeof word, hipMemcpyDeviceToHost,
                                  (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hbvx_lstm_check");
    if (word) return fail(HBVX_E_DEVICE, "lstm hand-off timed out: the waves of a row tile were not co-resident");
    return 0;
}

