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

#include "../../include/hbvx.h"

namespace hbvx_host {

int fail(int code, const char *msg);
int hip_fail(hipError_t e, const char *what);
int env_int(const char *name, int dflt);
int lg_members(int M);
int count_dyn(const hbvx_desc *d);
bool use_tiled(const hbvx_desc *d);      // false under HBVX_KERNEL=simple
int check_desc(const hbvx_desc *d);

static const int LDS_BUDGET = 160 * 1024 - 512; // gfx950: 160 KiB per CU, one workgroup may take it all

// launch_pipe.hip
bool try_fwd_pipe(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream, int *rc);
// launch_stream.hip
bool try_fwd_stream(const hbvx_desc *d, const hbvx_fwd_out *out, void *stream, int *rc, bool any_size = false);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device, size) and thread instead of once per
// launch: the attribute sticks, and the call is a driver round trip on the enqueue path of every step.
inline hipError_t set_dynamic_lds(const void *kern, int lds)
{
    struct Entry { const void *k; int dev, lds; };
    static thread_local Entry seen[16];
    static thread_local int n_seen = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (int i = 0; i < n_seen; i++)
        if (seen[i].k == kern && seen[i].dev == dev && seen[i].lds == lds) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) {
        seen[n_seen < 16 ? n_seen++ : (n_seen = 1, 0)] = Entry{kern, dev, lds};
    }
    return e;
}
// hbvx_bwd_io.store_gate: make `st` wait for the caller's event before a kernel that stores gradients is launched
inline void store_gate(const hbvx_bwd_io *io, hipStream_t st)
{
    if (io->store_gate) (void)hipStreamWaitEvent(st, (hipEvent_t)io->store_gate, 0);
}
bool try_bwd_stream(const hbvx_desc *d, const hbvx_bwd_io *io, void *stream, int *rc);
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
