/* hbvx_lstm.h -- C ABI of the sequence LSTM that feeds hbvx_forward (the
 * caller side of the hot path -- delta-MG's parameter network; it is not part of the reference
 * repository, the semantics are torch.nn.LSTM's: one layer, zero initial state, gate order i, f, g, o).
 * Exported by the same shared library as include/hbvx.h (libhbvx.so on the GPU, the CPU restatement
 * under oracle/ for tests).  Plain pointers and sizes; device pointers for the HIP library.
 *
 * The input projection x W_ih^T + b_ih + b_hh and the weight gradients are library GEMMs on the
 * caller's side (hydrodl2_amd/lstm.py); these entry points are the recurrence, which a GEMM library
 * cannot fuse.  Gate vectors use the (unit, gate) layout: element [t][b][u][g], g = 0..3 = i, f, g, o.
 *
 * Residency: the kernels are persistent -- the workgroups of a 16-basin row tile hand data to each
 * other inside one launch -- and each launch is sized to fit the whole GPU.  Run a call on a GPU that
 * is not executing another large kernel at the same time (other streams, other processes): partners
 * that cannot become resident are detected by bounded spins and reported (hbvx_lstm_check), never
 * waited for indefinitely.
 *
 * Errors: 0 on success, a negative HBVX_E_* (include/hbvx.h) otherwise; hbvx_last_error() has the text. */
#ifndef HBVX_LSTM_H
#define HBVX_LSTM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HBVX_LSTM_ABI_VERSION 1

typedef struct hbvx_lstm_desc {
    int32_t abi_version; /* HBVX_LSTM_ABI_VERSION */
    int32_t T, B, H;     /* steps, basins (batch), hidden units; the HIP library needs H in {64, 128, 256} */
} hbvx_lstm_desc;

/* Scratch for either call below (bytes; caller-owned, contents undefined afterwards): the per-step
 * exchange slabs through which the workgroups of a 16-basin row tile hand h_t (forward) or the gate
 * gradients (backward) to each other, and the arrival counters. */
uint64_t hbvx_lstm_workspace_bytes(const hbvx_lstm_desc *d);

/* w_hh [4H,H] (torch.nn.LSTM.weight_hh_l0), gx [T,B,H,4] = x W_ih^T + b_ih + b_hh in (unit, gate)
 * layout -> gates [T,B,H,4] (activated i, f, g, o; may alias gx), c_all and h_all [T,B,H]. */
int hbvx_lstm_forward(const hbvx_lstm_desc *d, const float *w_hh, const float *gx, float *gates,
                      float *c_all, float *h_all, void *workspace, uint64_t workspace_bytes,
                      void *stream);

/* grad_h [T,B,H] (gradient of the loss w.r.t. every h_t) -> grad_gates [T,B,H,4]: the gradient
 * w.r.t. the gate pre-activations, from which the caller forms grad_x, grad_W_ih, grad_W_hh and
 * the bias gradients with GEMMs.  grad_gates must not alias gates. */
int hbvx_lstm_backward(const hbvx_lstm_desc *d, const float *w_hh, const float *gates,
                       const float *c_all, const float *grad_h, float *grad_gates,
                       void *workspace, uint64_t workspace_bytes, void *stream);

/* Synchronises `stream` and reports whether the last call that used `workspace` completed: the
 * workgroups of a row tile wait for each other with bounded spins; a time-out (the partners were
 * not resident, e.g. the GPU was shared) poisons the outputs with NaN and is reported here. */
int hbvx_lstm_check(const hbvx_lstm_desc *d, const void *workspace, void *stream);

#ifdef __cplusplus
}
#endif
#endif
