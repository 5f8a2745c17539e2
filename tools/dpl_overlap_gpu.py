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
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2",
                   LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
        # stderr goes to a file per rank and only rank 0's stdout is piped: two PIPEs drained one after the other
        # dead-lock when the rank that is not being read fills its 64 KB pipe while the other waits for it in a collective
        err = open(os.path.join(ROOT, "gpurun_out", f"dpl_overlap_rank{rank}.err"), "w")
        procs.append((subprocess.Popen([sys.executable, os.path.join(ROOT, "examples", "train_dpl.py"), "--backend", "gloo",
                                        "--share-gpu", "--basins", "200", "--steps", "20", *extra], env=env,
                                       stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=err, text=True), err))
    out0, _ = procs[0][0].communicate(timeout=600)
    for rank, (p, err) in enumerate(procs):
        p.wait(timeout=600)
        err.close()
        if p.returncode:
            raise SystemExit(f"rank {rank} failed ({p.returncode}):\n" + open(err.name).read()[-2000:])
    return json.loads(out0.strip().splitlines()[-1])


if __name__ == "__main__":
    res = {"what": "examples/train_dpl.py, world 2 sharing one MI355X, gloo, LSTM-256 fused, 200 basins x 16 x (365+365) days",
           "overlapped": run(()), "blocking": run(("--no-overlap",)), "one_rank": None}
    env = dict(os.environ)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_dpl.py"), "--basins", "100", "--steps", "20"],
                         capture_output=True, text=True, env=env)
    if one.returncode == 0:
        res["one_rank"] = json.loads(one.stdout.strip().splitlines()[-1])
    print(json.dumps(res))
