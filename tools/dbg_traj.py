import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.abi_util import make_problem, run_problem
prob = make_problem(model="Hbv", T=37, B=8200, M=16, dyn=(), seed=33)
a = run_problem(prob, None, device="cuda:0", backward=True)
os.environ["HBVX_STREAM"] = "0"
b = run_problem(prob, None, device="cuda:0", backward=True)
for k in ("flux", "state_out", "traj"):
    x, y = a[k], b[k]
    bad = ~((x == y) | (np.isnan(x) & np.isnan(y)))
    print(k, x.shape, "mismatch", int(bad.sum()), "nan a", int(np.isnan(x).sum()), "nan b", int(np.isnan(y).sum()))
    if bad.any():
        idx = np.argwhere(bad)
        print(" first", idx[:5], "last", idx[-5:], "vals", x[tuple(idx[0])], y[tuple(idx[0])])
        print(" per-k counts", [int(bad[k].sum()) for k in range(bad.shape[0])])
        print(" per-t counts", bad.sum(axis=(0, 2))[:10], bad.sum(axis=(0,2))[-5:])
