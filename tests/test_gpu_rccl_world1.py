"""GPU tier: the sharding helpers over the REAL backend ("nccl" = RCCL), world size 1.

A one-GPU box cannot run a second rank (RCCL refuses two ranks on one device), but a one-rank RCCL communicator
still goes through everything a first multi-GPU run could trip on that is not the wire: library load,
communicator creation with `device_id`, `ncclAllReduce` / `ncclAllGather` launches on the HIP stream of the
HBV kernels, the persistent buckets and the self-validating check bench.py runs before it times anything.
Runs in a child process (the default process group is process-wide state)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json
sys.path.insert(0, os.environ["HBVX_ROOT"])
import torch, torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ["HBVX_PORT"], rank=0, world_size=1,
                        device_id=dev)
import bench
from hydrodl2_amd import sharding
import hydrodl2_amd
info = bench.validate_collective(dev, 1, 0)          # world 1: reports the RCCL version, no reduction
ones = torch.ones(64, device=dev)
dist.all_reduce(ones)                                 # ncclAllReduce on one rank
assert float(ones.min()) == 1.0 and float(ones.max()) == 1.0
Hbv = hydrodl2_amd.load_model("hbv", "Hbv")
m = Hbv({"nmul": 4, "dynamic_params": {"Hbv": ["parBETA"]}, "warm_up": 5}, dev)
torch.manual_seed(3)
T, B = 40, 6
x = torch.rand(T, B, 3, device=dev) * torch.tensor([8.0, 20.0, 4.0], device=dev) - torch.tensor([0.0, 5.0, 0.0], device=dev)
p = torch.randn(T, B, 13 * 4 + 2, device=dev, requires_grad=True)
xs, ps = sharding.shard_inputs({"x_phy": x}, p, 1, 0)
out = m(xs, ps)
out["streamflow"].sum().backward()
shared = p.grad.sum(1)
ref = shared.clone()
lossv = out["streamflow"].detach().sum().reshape(1)
for _ in range(2):                                   # twice: the second call reuses the persistent bucket
    sharding.all_reduce_sum_([shared, lossv])
assert torch.equal(shared, ref)
b = sharding.AsyncBucket([shared]).start()
b.finish()
assert torch.equal(shared, ref)
full = sharding.gather_flux_dict({k: v.detach() for k, v in out.items()}, B)
assert all(torch.equal(full[k], out[k].detach()) for k in out)
torch.cuda.synchronize()
dist.destroy_process_group()
print(json.dumps({"ok": True, "backend": "nccl", "rccl": info["nccl_version"]}))
'''


@pytest.mark.gpu
def test_sharding_helpers_over_a_one_rank_rccl_communicator(hip_backend):
    from .test_sharding_gloo import _free_port
    env = dict(os.environ, HBVX_ROOT=ROOT, HBVX_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert '"ok": true' in r.stdout
