#!/usr/bin/env python3
"""Time the fused smoothing passes the cycle launches on the fine level (restricting pre-smoothing pass, correcting
post-smoothing pass, and the plain / residual variants) for a list of tuning settings.

    python tools/scan_fused.py --size 4096 --set fused_seg_lines=28 --set fused_seg_lines=48,fused_slow_pct=50
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnmultigrid_amd import ops, problems as P
from learnmultigrid_amd.hierarchy import Hierarchy

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--set", action="append", default=[], help="comma-separated key=value tune settings of one run ('base' = none)")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--kinds", default="rest,prol,resid,plain")
ap.add_argument("--sweeps", type=int, default=3)
a = ap.parse_args()
m = a.size
dev = torch.device("cuda:0")
A, rhs = P.poisson_2d_structured(m)
ops.FUSED_MIN_ROWS = 0
ops.FUSED_TRANSFER_MIN_ROWS = 0
ops.set_tiled_enabled(False)
H = Hierarchy(A, P.geometric_hierarchy_2d(m + 1, 6 if m >= 2048 else 3), dev)
lev = H.levels[0]
fa = lev.A
n = fa.shape[0]
nc = lev.P.shape[1]
x = torch.rand(n, dtype=torch.float64, device=dev); b = torch.rand_like(x); y = torch.empty_like(x); r = torch.empty_like(x)
e = torch.rand(nc, dtype=torch.float64, device=dev); bc = torch.empty_like(e)
assert ops.stencil_smooth_prolong_available(fa, lev.P) and ops.stencil_smooth_restrict_available(fa, lev.R)
calls = {
    "rest": (lambda: ops.stencil_smooth(fa, x, b, 0.8, a.sweeps, y, None, restrict=(lev.R, bc)), n * 25 + nc * 9),
    "prol": (lambda: ops.stencil_smooth(fa, x, b, 0.8, a.sweeps, y, None, prolong=(lev.P, e)), n * 26 + nc * 8),
    "resid": (lambda: ops.stencil_smooth(fa, x, b, 0.8, a.sweeps, y, r), n * 33),
    "plain": (lambda: ops.stencil_smooth(fa, x, b, 0.8, a.sweeps, y, None), n * 25),
}


def timeit(f, reps):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for setting in (a.set or ["base"]):
    kv = [] if setting == "base" else [s.split("=") for s in setting.split(",")]
    old = [(k, ops.tune_get(k)) for k, _ in kv]
    for k, v in kv: ops.tune_set(k, int(v))
    out = []
    for kind in a.kinds.split(","):
        f, moved = calls[kind]
        t = timeit(f, a.reps)
        out.append("%s %.4f ms %4.0f GB/s" % (kind, t, moved / t / 1e6))
    print("%-60s %s" % (setting, " | ".join(out)), flush=True)
    for k, v in old: ops.tune_set(k, v)
