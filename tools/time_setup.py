#!/usr/bin/env python3
"""Where the hierarchy setup time goes (cfg#4 by default)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnmultigrid_amd import ops, problems as P, hierarchy as Hm
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=4096); ap.add_argument("--levels", type=int, default=6)
a = ap.parse_args()
m = a.size
t = time.perf_counter(); A, rhs = P.poisson_2d_structured(m); hier = P.geometric_hierarchy_2d(m + 1, a.levels); print("host problem + transfers %.2f s" % (time.perf_counter() - t))
torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
marks = []
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*aa, **kk):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*aa, **kk); torch.cuda.synchronize(); marks.append((label, time.perf_counter() - t0)); return r
    setattr(obj, name, g)
wrap(ops.DeviceCSR, "from_scipy", "upload (from_scipy)")
wrap(Hm, "_to_csr_host", "host csr conversion")
orig_plan = ops.SpGEMMPlan.__init__
def plan_init(self, *aa):
    torch.cuda.synchronize(); t0 = time.perf_counter(); orig_plan(self, *aa); torch.cuda.synchronize(); marks.append(("spgemm symbolic", time.perf_counter() - t0))
ops.SpGEMMPlan.__init__ = plan_init
wrap(ops.SpGEMMPlan, "numeric", "spgemm numeric")
wrap(Hm.Hierarchy, "_pack_all", "pack (all operators)")
wrap(Hm.Hierarchy, "_factor_coarsest", "coarse factorisation")
from learnmultigrid_amd import coarse as C
wrap(C, "dense_inverse", "  (coarse: dense inverses)")
wrap(ops.DeviceCSR, "transpose", "device transpose")
wrap(Hm.Hierarchy, "_inverse_diagonals", "inverse diagonals")
for run in ("cold (first hierarchy of the process)", "warm (second hierarchy)"):
    marks.clear()
    t = time.perf_counter(); H = Hm.Hierarchy(A, hier, "cuda:0"); torch.cuda.synchronize(); tot = time.perf_counter() - t
    agg = {}
    for k, v in marks: agg[k] = agg.get(k, 0.0) + v
    print("---- " + run)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]): print("%-28s %7.3f s" % (k, v))
    print("%-28s %7.3f s" % ("Hierarchy total", tot))
    del H
