import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.abi_util import make_problem, run_problem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8200
prob = make_problem(model="Hbv", T=37, B=B, M=16, dyn=(), seed=33)
a = run_problem(prob, None, device="cuda:0", backward=True)
a1 = run_problem(prob, None, device="cuda:0", backward=True)
os.environ["HBVX_STREAM"] = "0"
b = run_problem(prob, None, device="cuda:0", backward=True)
ga, ga1, gb = a["g_params"][-1], a1["g_params"][-1], b["g_params"][-1]
bad = np.abs(ga - gb) > 1e-4 * np.abs(gb) + 1e-5 * np.abs(gb).max()
print("B", B, "bad", int(bad.sum()), "of", bad.size, "repeatable", np.array_equal(ga, ga1))
if bad.any():
    idx = np.argwhere(bad)
    bb = np.unique(idx[:, 0])
    print("basins affected", len(bb), "first", bb[:12], "last", bb[-5:])
    print("groups (b//4) parity of blocks:", np.unique((bb // 4) % 8, return_counts=True))
    print("cols", np.unique(idx[:, 1] // 16, return_counts=True))
