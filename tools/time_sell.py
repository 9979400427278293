#!/usr/bin/env python3
"""A/B of sell_* tune settings on the learned-like Galerkin operators (25 / ~48 entries per row), interleaved in one process.
    python tools/time_sell.py --size 4096 --matrix L1 --set sell_nt=0 --set sell_nt=1"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, scipy.sparse as sp
from learnmultigrid_amd import ops, problems as P

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--matrix", default="L1")
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--reps", type=int, default=9)
a = ap.parse_args()
m = a.size
M, _ = P.variable_coeff_poisson_2d_structured(m, seed=44)
for li, sz in enumerate(P.level_sizes(m + 1, 3)[:2 if a.matrix == "L2" else 1]):
    l2 = P.pseudo_l2_interpolator_1d(sz)
    Q = P.learned_like(sp.kron(l2, l2).tocsr(), 43 + li)
    M = sp.csr_matrix(Q.T @ M @ Q)
M.sort_indices()
dev = "cuda:0"
dM = ops.DeviceCSR.from_scipy(M, dev)
dM.pack()
n = M.shape[0]
x = torch.rand(n, dtype=torch.float64, device=dev); b = torch.rand_like(x); y = torch.zeros_like(x)
print("matrix %s n %d nnz/row %.1f twin %s" % (a.matrix, n, M.nnz / n, "sell" if dM.sell is not None else "other"))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {s: [] for s in a.set}
for rep in range(a.reps + 1):
    for s in a.set:
        for kv in s.split(","):
            k, v = kv.split("="); ops.tune_set(k, int(v))
        ops.csr_jacobi(dM, x, b, 0.8, y); torch.cuda.synchronize()
        ev0.record()
        for _ in range(10): ops.csr_jacobi(dM, x, b, 0.8, y)
        ev1.record(); torch.cuda.synchronize()
        if rep: res[s].append(ev0.elapsed_time(ev1) / 10)
for s in a.set:
    t = np.array(res[s]); print("%-30s median %.4f ms min %.4f  (%.0f GB/s on 10 B/entry + 24 B/row)" % (s, np.median(t), t.min(), (10 * M.nnz + 24 * n) / np.median(t) / 1e6))
