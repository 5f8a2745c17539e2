#!/usr/bin/env python3
"""Forward-kernel probe at BASELINE config 2: time hbvx_forward alone with / without the saved
trajectory and under the launch knobs (HBVX_FWD, HBVX_PIPE_THREADS, HBVX_KT).

    python tools/fwd_probe.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import hydrodl2_amd  # noqa: E402
from tools.bench_configs import gen  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    T, B, M = 7300, 671, 16
    model = hydrodl2_amd.load_model("hbv", "Hbv")({"nmul": M, "dynamic_params": {"Hbv": []}}, dev)
    x, g = gen(T, B, dev)
    p = torch.randn((T, B, model.learnable_param_count), generator=g, device=dev)
    combos = [{}, {"HBVX_PIPE_THREADS": "768"}, {"HBVX_PIPE_THREADS": "512"}, {"HBVX_FWD": "tiled"},
              {"HBVX_FWD": "tiled", "HBVX_KT": "8"}]
    for env in combos:
        for k in ("HBVX_PIPE_THREADS", "HBVX_FWD", "HBVX_KT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for grad in (True, False):
            pp = p.clone().requires_grad_(grad)
            ts = []
            for i in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.set_grad_enabled(grad):
                    torch.cuda.synchronize()
                    e0.record()
                    out = model({"x_phy": x}, pp)
                    e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
                del out
            print(json.dumps({"env": env, "traj": grad, "module_fwd_ms": round(min(ts[2:]), 3)}), flush=True)


if __name__ == "__main__":
    main()
