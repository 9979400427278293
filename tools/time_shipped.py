#!/usr/bin/env python3
"""One V(3,3) cycle with the reference's shipped semantics (forward Gauss-Seidel whatever the smoother's name,
Multigrid.py:88) on the device, hipGraph replay: cfg#2 and cfg#4 sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnmultigrid_amd import problems as P
from learnmultigrid_amd.hierarchy import Hierarchy

for m, levels in ((512, 3), (4096, 6)):
    A, rhs = P.poisson_2d_structured(m)
    hier = P.geometric_hierarchy_2d(m + 1, levels)
    H = Hierarchy(A, hier, torch.device("cuda:0"))
    H.levels[0].b.copy_(torch.from_numpy(rhs.ravel().copy()).to("cuda:0"))
    H.stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(H.stream):
        g = H.captured_cycle("GaussSeidel", 3, 1.0, "lexicographic")
        for _ in range(2):
            g.launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.launch()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        r = H.residual_norm(want_vector=False)
    print("%d^2, %d levels, V(3,3) exact forward Gauss-Seidel: %.2f ms per cycle (residual after 7 cycles %.3e)" % (m + 1, levels, dt * 1e3, r))
