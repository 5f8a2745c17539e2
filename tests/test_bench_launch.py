"""CPU tier: `python bench.py --gpus 2` starts its own ranks and prints one JSON line.

The ranks run on the host over gloo with the oracle standing in for the HIP library (installed in
every rank by tests/cpu_seam/sitecustomize.py): what is under test is the launcher, the sharding
(weak for cfg2, strong with an uneven split for cfg5), the max-over-ranks timing and the line's
contract -- not any arithmetic."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(oracle_path, *extra):
    env = dict(os.environ, HBVX_TEST_ABI_LIBRARY=oracle_path, HBVX_TEST_ROOT=ROOT, OMP_NUM_THREADS="1",
               PYTHONPATH=os.path.join(ROOT, "tests", "cpu_seam") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "cpu", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-secondary", *extra]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_self_launch_weak_cfg2(oracle_path):
    r = _run(oracle_path, "--config", "cfg2", "--basins", "6", "--days", "40", "--nmul", "4")
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["scaling"] == "weak"
    assert r["config"]["basins_total"] == 12 and r["config"]["basins_per_gpu"] == 6
    assert len(r["rank_ms_per_step"]) == 2 and r["ms_per_step"] == pytest.approx(max(r["rank_ms_per_step"]), rel=1e-3)
    assert r["value"] == pytest.approx(12 * 4 * 40 / (r["ms_per_step"] * 1e-3), rel=1e-6)
    assert r["metric"].startswith("basin-ensemble-timesteps/sec") and r["dtype"] == "f32"


def test_self_launch_strong_cfg5_uneven(oracle_path):
    r = _run(oracle_path, "--config", "cfg5", "--basins", "9", "--days", "30", "--nmul", "2")
    assert r["n_gpus"] == 2 and r["scaling"] == "strong"
    assert r["config"]["basins_total"] == 9 and r["config"]["basins_per_gpu"] == 5   # rank 0 of ceil split 5 + 4
    assert r["value"] == pytest.approx(9 * 2 * 30 / (r["ms_per_step"] * 1e-3), rel=1e-6)


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "cpu"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)
