#!/usr/bin/env python3
"""examples/train_dpl.py at world 2 on ONE GPU (both ranks on cuda:0, gloo carrying the gradients through
the host): the overlapped two-bucket all-reduce against the single blocking one, with the HIP kernels.
A rehearsal of the schedule, not an RCCL measurement -- that needs the driver's multi-GPU node.

    python tools/dpl_overlap_gpu.py > gpurun_out/dpl_overlap_gpu.json
"""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(extra):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2",
                   LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "examples", "train_dpl.py"), "--backend", "gloo",
                                       "--share-gpu", "--basins", "200", "--steps", "20", *extra],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        if p.returncode:
            raise SystemExit(f"rank failed ({p.returncode}):\n{e[-2000:]}")
    return json.loads(outs[0][0].strip().splitlines()[-1])


if __name__ == "__main__":
    res = {"what": "examples/train_dpl.py, world 2 sharing one MI355X, gloo, LSTM-256 fused, 200 basins x 16 x (365+365) days",
           "overlapped": run(()), "blocking": run(("--no-overlap",)), "one_rank": None}
    env = dict(os.environ)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_dpl.py"), "--basins", "100", "--steps", "20"],
                         capture_output=True, text=True, env=env)
    if one.returncode == 0:
        res["one_rank"] = json.loads(one.stdout.strip().splitlines()[-1])
    print(json.dumps(res))
