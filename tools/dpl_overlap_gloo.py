#!/usr/bin/env python3
"""examples/train_dpl.py at world 2 over gloo on the host (the oracle standing in for the HIP library,
through tests/seam.py), with the overlapped two-bucket all-reduce and with the single blocking one.
Writes profiles/r02_dpl_overlap_gloo.json.  (RCCL numbers need a multi-GPU node: the driver's.)"""
import importlib.util
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, oracle, extra, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), OMP_NUM_THREADS="2")
    import io, contextlib
    import torch
    torch.set_num_threads(2)
    from tests import seam
    seam.use_library(oracle)
    spec = importlib.util.spec_from_file_location("train_dpl", os.path.join(ROOT, "examples", "train_dpl.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.argv = ["train_dpl.py", "--basins", "48", "--rho", "120", "--warm-up", "60", "--nmul", "8", "--hidden", "256",
                "--steps", "8", "--device", "cpu", "--lstm", "torch", *extra]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        mod.main()
    if rank == 0:
        q.put(json.loads(buf.getvalue().strip().splitlines()[-1]))


def main():
    import torch.multiprocessing as mp
    import __graft_entry__ as ge
    if not os.path.exists(ge.ORACLE_LIB):
        ge.build_oracle()
    out = {}
    for name, extra in (("overlapped", ()), ("blocking", ("--no-overlap",))):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, 2, port, ge.ORACLE_LIB, extra, q)) for r in range(2)]
        for p in procs:
            p.start()
        out[name] = q.get(timeout=900)
        for p in procs:
            p.join(timeout=60)
    res = {"what": "examples/train_dpl.py, world 2, gloo on host cores, LSTM-256, 48 basins x 8 members x 180 days",
           "overlapped_ms_per_step": out["overlapped"]["ms_per_step"], "blocking_ms_per_step": out["blocking"]["ms_per_step"],
           "loss_last": [out["overlapped"]["loss_last"], out["blocking"]["loss_last"]]}
    with open(os.path.join(ROOT, "profiles", "r02_dpl_overlap_gloo.json"), "w") as f:
        f.write(json.dumps(res) + "\n")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
