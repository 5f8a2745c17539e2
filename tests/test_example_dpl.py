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
