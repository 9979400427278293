#!/usr/bin/env python3
"""Time the LDS-tiled smoothing passes on level 1 of cfg#4 (2049^2, 9-point Galerkin operator) for 1 / 2 / 3 sweeps per pass:
what part of a pass depends on the number of sweeps?   python tools/scan_tile.py [--size 4096] [--set key=value,...]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnmultigrid_amd import ops, problems as P
from learnmultigrid_amd.hierarchy import Hierarchy

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--level", type=int, default=1)
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
m = a.size
dev = torch.device("cuda:0")
A, rhs = P.poisson_2d_structured(m)
H = Hierarchy(A, P.geometric_hierarchy_2d(m + 1, 6 if m >= 2048 else 4), dev)
lev = H.levels[a.level]
fa = lev.A
n, nc = fa.shape[0], lev.P.shape[1]
print("level %d: n = %d, kind %s" % (a.level, n, ops._fused_kind(fa)))
x = torch.rand(n, dtype=torch.float64, device=dev); b = torch.rand_like(x); y = torch.empty_like(x); r = torch.empty_like(x)
e = torch.rand(nc, dtype=torch.float64, device=dev); bc = torch.empty_like(e)


def timeit(f, reps):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for setting in (a.set or ["base"]):
    kv = [] if setting == "base" else [s.split("=") for s in setting.split(",")]
    old = [(k, ops.tune_get(k)) for k, _ in kv]
    for k, v in kv: ops.tune_set(k, int(v))
    for S in (1, 2, 3):
        calls = {
            "rest": lambda: ops.stencil_smooth(fa, x, b, 0.8, S, y, None, restrict=(lev.R, bc)),
            "prol": lambda: ops.stencil_smooth(fa, x, b, 0.8, S, y, None, prolong=(lev.P, e)),
            "resid": lambda: ops.stencil_smooth(fa, x, b, 0.8, S, y, r),
            "plain": lambda: ops.stencil_smooth(fa, x, b, 0.8, S, y, None),
        }
        print("%-28s S=%d  " % (setting, S) + " | ".join("%s %.1f us" % (k, timeit(f, a.reps)) for k, f in calls.items()), flush=True)
    t1 = timeit(lambda: ops.csr_jacobi(fa, x, b, 0.8, y), a.reps)
    print("%-28s single sweep %.1f us" % (setting, t1))
    for k, v in old: ops.tune_set(k, v)
