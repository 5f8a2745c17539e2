"""The end-to-end dPL example (examples/train_dpl.py) runs and learns:
CPU tier through the oracle backend, GPU tier through the HIP library."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("train_dpl", os.path.join(ROOT, "examples", "train_dpl.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run(device, monkeypatch, hidden="16", lstm="fused"):
    mod = _load()
    monkeypatch.setattr(sys, "argv", ["train_dpl.py", "--basins", "6", "--rho", "40", "--warm-up", "20",
                                      "--nmul", "4", "--hidden", hidden, "--steps", "12", "--device", device,
                                      "--lstm", lstm])
    losses = mod.main()
    assert all(l == l for l in losses)           # finite
    assert min(losses[-3:]) < losses[0]          # the optimiser makes progress on 1 - NSE


def test_dpl_example_learns_cpu_oracle(oracle_backend, monkeypatch):
    _run("cpu", monkeypatch)


@pytest.mark.gpu
def test_dpl_example_learns_gpu(hip_backend, monkeypatch):
    _run("cuda", monkeypatch, hidden="64")


@pytest.mark.gpu
def test_dpl_example_fused_and_torch_lstm_agree(hip_backend, monkeypatch):
    """Same seeds, same data: the loss curve with the HIP sequence kernels follows torch.nn.LSTM's."""
    mod = _load()
    curves = []
    for lstm in ("fused", "torch"):
        monkeypatch.setattr(sys, "argv", ["train_dpl.py", "--basins", "6", "--rho", "40", "--warm-up", "20",
                                          "--nmul", "4", "--hidden", "64", "--steps", "5", "--lstm", lstm])
        curves.append(mod.main())
    for a, b in zip(*curves):
        assert abs(a - b) <= 2e-3 * max(abs(b), 1e-3), curves


@pytest.mark.gpu
def test_dpl_whole_step_graph_follows_the_eager_run(hip_backend):
    """examples/train_dpl.py --graph: the whole step (network, HBV, loss, backward, Adam) replayed as one captured HIP
    graph gives the loss curve of the eager run with the same (capturable) optimiser -- the same kernels in the same
    order, so step for step the same numbers; the LSTM's exchange slabs are re-armed by a kernel on every replay (a
    replay that reused stale slabs would not follow the eager curve past the first replayed step)."""
    import torch
    mod = _load()
    dev = torch.device("cuda:0")
    kw = dict(basins=20, rho=60, warm_up=30, nmul=4, hidden=64)
    step, _ = mod.make_trainer(dev, capturable=True, **kw)
    eager = [step() for _ in range(8)]
    gstep, info = mod.make_trainer(dev, graph=True, **kw)
    graphed = list(info["eager_losses"]) + [gstep() for _ in range(5)]
    assert all(l == l for l in graphed)
    assert graphed[:3] == eager[:3], (eager, graphed)            # the three eager steps before the capture
    for a, b in zip(graphed[3:], eager[3:]):
        assert abs(a - b) <= 1e-5 * max(abs(b), 1e-3), (eager, graphed)
    assert graphed[-1] < graphed[0]
    with pytest.raises(ValueError, match="graph=True"):
        mod.make_trainer(dev, graph=True, lstm="torch", **kw)


def _dpl_worker(rank, world, port, oracle, q, extra=()):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch
    torch.set_num_threads(1)
    from tests import seam
    seam.use_library(oracle)
    sys.argv = ["train_dpl.py", "--basins", "7", "--rho", "30", "--warm-up", "10", "--nmul", "2",
                "--hidden", "8", "--steps", "4", "--device", "cpu", *extra]
    losses = _load().main()
    if rank == 0:
        q.put(losses)


@pytest.mark.parametrize("extra", [(), ("--no-overlap",)], ids=["overlapped", "blocking"])
def test_dpl_example_two_ranks_match_one(extra, oracle_path, oracle_backend, monkeypatch):
    """The whole training step (LSTM -> HBV -> NSE -> Adam) sharded over two gloo ranks (7 basins: uneven
    shards) follows the single-process loss curve.  All the ranks exchange per step is the loss
    normalisers and the network gradients: by default in two buckets, the first (output layer) sent
    asynchronously from a gradient hook while the LSTM backward is still running; --no-overlap: one
    blocking bucket after backward."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dpl_worker, args=(r, 2, port, oracle_path, q, extra)) for r in range(2)]
    for p in procs:
        p.start()
    two = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    monkeypatch.setattr(sys, "argv", ["train_dpl.py", "--basins", "7", "--rho", "30", "--warm-up", "10", "--nmul", "2",
                                      "--hidden", "8", "--steps", "4", "--device", "cpu"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    one = _load().main()
    assert len(one) == len(two)
    for a, b in zip(two, one):
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-3), (two, one)
