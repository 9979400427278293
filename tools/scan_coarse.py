#!/usr/bin/env python3
"""The cfg#4 cycle with the grid-block coarse solver cut into G x G blocks, G forced: is the byte-minimal choice the fastest?
    python tools/scan_coarse.py --g 8,9,10,11,12,14"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnmultigrid_amd import coarse, ops, problems as P
from learnmultigrid_amd.hierarchy import Hierarchy

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--levels", type=int, default=6)
ap.add_argument("--g", default="0,8,9,10,11,12,14")
a = ap.parse_args()
A, rhs = P.poisson_2d_structured(a.size)
hier = P.geometric_hierarchy_2d(a.size + 1, a.levels)
dev = torch.device("cuda:0")
for g in [int(v) for v in a.g.split(",")]:
    coarse.GRID_G_FORCE = g or None
    H = Hierarchy(A, hier, dev)
    with torch.cuda.stream(H.stream):
        H.levels[0].b.copy_(torch.from_numpy(rhs.ravel().copy()).to(dev))
        gr = H.captured_cycle("Jacobi", 3, 0.8, "lexicographic")
        for _ in range(5): gr.launch()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(30): gr.launch()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 30)
    ts.sort()
    print("G = %2s: %s, %.1f MB per application, refine %s, cycle median %.4f ms" % (g or "auto", H.coarse.kind, H.coarse.dense_bytes / 1e6 if hasattr(H.coarse, "dense_bytes") else -1, H.coarse_refine, ts[2] * 1e3), flush=True)
    del H, gr
    torch.cuda.empty_cache()
