#!/usr/bin/env python3
"""cProfile of the host side of the hierarchy setup (cfg#4 by default): cold (first hierarchy of the process) and warm."""
import argparse, cProfile, os, pstats, sys, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnmultigrid_amd import problems as P, hierarchy as Hm
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=4096); ap.add_argument("--levels", type=int, default=6)
ap.add_argument("--top", type=int, default=35)
a = ap.parse_args()
A, rhs = P.poisson_2d_structured(a.size); hier = P.geometric_hierarchy_2d(a.size + 1, a.levels)
torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
for run in ("cold", "warm"):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable(); H = Hm.Hierarchy(A, hier, "cuda:0"); torch.cuda.synchronize(); pr.disable()
    print("==== %s: %.3f s" % (run, time.perf_counter() - t0))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(a.top); print(s.getvalue()[:9000])
    del H
