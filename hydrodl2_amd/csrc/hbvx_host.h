// hbvx_host.h -- host-side plumbing shared by the translation units of libhbvx.so.
//
// The library is built from several .hip files compiled in parallel (one per kernel family:
// pipelined / streaming / tiled / time-parallel / implicit / LSTM) so that touching one family
// rebuilds one file.  The C ABI (include/hbvx.h) lives in hbvx.hip; its dispatchers call the
// family launchers declared here.  A launcher returns true when its family took the call and has
// then set *rc to the ABI return code.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/hbvx.h"
#include "hbv_step.h"

namespace hbvx_host {

int fail(int code, const char *msg);
int hip_fail(hipError_t e, const char *what);
// hbvx_last_dispatch (include/hbvx.h): the kernel family that took the call; dir 0 forward, 1 adjoint
void note_dispatch(int dir, const char *family);
int env_int(const char *name, int dflt);
int device_cu_count();                   // compute units of the current device (cached per device)
// hbvx_fwd_out.zero_ptr: set by the kernel family that carries fill workgroups in its own launch (the pipelined forward);
// the entry points clear it before dispatch (hbvx_zero_in_launch reports it)
bool &zero_taken();
int lg_members(int M);
int count_dyn(const hbvx_desc *d);
bool use_tiled(const hbvx_desc *d);      // false under HBVX_KERNEL=simple
int check_desc(const hbvx_desc *d);
// whether this call's aux pointers are consistent with its trajectory: only HBVX_SAVE_POW builds look at aux at all
inline bool aux_matches_traj(const hbvx_fwd_out *out) { return !hbvx::SAVE_POW || ((out->traj != nullptr) == (out->aux != nullptr)); }

static const int LDS_BUDGET = 160 * 1024 - 512; // gfx950: 160 KiB per CU, one workgroup may take it all

// launch_pipe.hip
bool try_fwd_pipe(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream, int *rc);
// launch_stream.hip
bool try_fwd_stream(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream, int *rc, bool any_size = false);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) only when a launch needs MORE than the kernel was ever granted on
// this device: the attribute holds one value per kernel (the last one set) and is a ceiling for the launch, so it is
// raised and never lowered -- a template instance whose LDS request varies with the problem (train and validation
// models with different dynamic sets) would otherwise launch "big, small, big" with the attribute left at "small".
// One process-wide table under a mutex (autograd's backward thread and the main thread share it); hbvx.hip defines it.
hipError_t set_dynamic_lds(const void *kern, int lds);
// hbvx_bwd_io.store_gate: make `st` wait for the caller's event before a kernel that stores gradients is launched
inline void store_gate(const hbvx_bwd_io *io, hipStream_t st)
{
    if (io->store_gate) (void)hipStreamWaitEvent(st, (hipEvent_t)io->store_gate, 0);
}
bool try_bwd_stream(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream, int *rc);
// HBVX_TRAJ_CKPT with the segment in LDS (hbv_stream2_ckpt.h): whether it takes this problem (then no scratch is
// wanted: hbvx_ckpt_workspace_bytes returns 0), and the launch
bool stream_ckpt_applicable(const hbvx_desc *d, int K);
bool try_bwd_stream_ckpt(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream, int *rc);
// launch_tiled.hip
bool try_fwd_tiled(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream, int *rc);
bool try_bwd_tiled(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream, int *rc);
// launch_chunked.hip
bool chunked_applicable(const hbvx_desc *d);
int chunk_days();
bool try_bwd_chunked(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream, int *rc);

// launch_ckpt.hip
bool try_bwd_ckpt(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream, int *rc);

} // namespace hbvx_host
