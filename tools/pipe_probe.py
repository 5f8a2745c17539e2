#!/usr/bin/env python3
"""Dev probe: per-wave busy / barrier-wait cycles of k_fwd_pipe (library built with -DPIPE_PROBE
as hydrodl2_amd/csrc/libhbvx_probe.so), BASELINE config 2."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import hydrodl2_amd  # noqa: E402
from hydrodl2_amd import _lib  # noqa: E402
from tests import seam  # noqa: E402
from tools.bench_configs import gen  # noqa: E402

PROBE_LIB = os.path.join(ROOT, "hydrodl2_amd", "csrc", "libhbvx_probe.so")
if not os.path.exists(PROBE_LIB) or os.path.getmtime(PROBE_LIB) < os.path.getmtime(os.path.join(ROOT, "hydrodl2_amd", "csrc", "libhbvx.so")):
    import subprocess
    subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_variant.sh"), "probe", "launch_pipe", "-DPIPE_PROBE"])
seam.use_library(PROBE_LIB)
lib = _lib.get_library()
dev = torch.device("cuda:0")
T, B, M = 7300, 671, 16
DYN = os.environ.get("PROBE_DYN", "").split(",") if os.environ.get("PROBE_DYN") else []
MOD, CLS = os.environ.get("PROBE_MODEL", "hbv:Hbv").split(":")
H = hydrodl2_amd.load_model(MOD, CLS)
if DYN == ["all"]:
    DYN = list(H(None, dev).parameter_bounds)
model = H({"nmul": M, "dynamic_params": {CLS: DYN}}, dev)
print("dynamic:", DYN)
x, g = gen(T, B, dev)
p = torch.randn((T, B, model.learnable_param_count), generator=g, device=dev)
for grad in (True, False):
    pp = p.clone().requires_grad_(grad)
    with torch.set_grad_enabled(grad):
        for _ in range(3):
            out = model({"x_phy": x}, pp)
    torch.cuda.synchronize()
    # kernel time of this (probe) build beside its cycle counts: ticks per microsecond of the counter
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.set_grad_enabled(grad):
        e0.record()
        out = model({"x_phy": x}, pp)
        e1.record()
    torch.cuda.synchronize()
    ms_call = e0.elapsed_time(e1)
    buf = (C.c_ulonglong * 32)()
    assert lib.dll.hbvx_debug_pipe_probe(buf) == 0
    print("traj" if grad else "no traj")
    nb = (B + 3) // 4
    blk = (C.c_ulonglong * nb)()
    if hasattr(lib.dll, "hbvx_debug_pipe_blocks") and lib.dll.hbvx_debug_pipe_blocks(blk, nb) == 0:
        v = sorted(x / T for x in blk)
        print(f"  per-workgroup cycles/day (soil wave, {nb} workgroups): min {v[0]:.1f}  median {v[nb // 2]:.1f}  "
              f"p90 {v[int(nb * 0.9)]:.1f}  max {v[-1]:.1f}")
        print(f"  model call {ms_call:.3f} ms (all its kernels) -> {v[-1] * T / (ms_call * 1e3):.0f} counter ticks per us "
              f"if the slowest workgroup spanned the whole call")
    roles = ["snow", "soil", "gw", "fill", "fill"] + ["drain"] * 11
    for w in range(16):
        print(f"  wave {w:2d} {roles[w]:6s} busy {buf[2*w]/T:8.1f}  wait {buf[2*w+1]/T:8.1f} counter ticks/step")
