#!/usr/bin/env python3
"""Time one exact (lexicographic) Gauss-Seidel sweep on the reference's typical problems.
`python tests/time_gs.py big` times the large 2-D grids with both executors (per-set launches
and the one-workgroup persistent kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learnmultigrid_amd import ops, problems as P
from oracle import kernels as K

big = len(sys.argv) > 1 and sys.argv[1] == "big"
if big:
    cases = [("2-D 1025^2", P.poisson_2d_structured(1024)[0]), ("2-D 2049^2", P.poisson_2d_structured(2048)[0]),
             ("2-D 4097^2", P.poisson_2d_structured(4096)[0])]
else:
    cases = [("1-D ne=14960", P.poisson_1d_fd(14960)[0]), ("1-D ne=1024", P.poisson_1d_fd(1024)[0]),
             ("2-D 101^2", P.poisson_2d_structured(100)[0]), ("2-D 513^2", P.poisson_2d_structured(512)[0])]
for name, A in cases:
    A = K.as_csr(A)
    n = A.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, "cuda:0")
    sch = ops.build_gs_schedule(A, "lexicographic", "cuda:0")
    rng = np.random.default_rng(0)
    x0 = rng.standard_normal(n)
    b = rng.standard_normal(n)
    db = torch.from_numpy(b).cuda()
    w = x0.copy()
    t = time.perf_counter(); K.gs_forward(A, w, b, 1); dc = time.perf_counter() - t
    for single_max in (2048,):
        ops.tune_set("gs_single_max", single_max)
        x = torch.from_numpy(x0.copy()).cuda()
        ops.csr_gs_schedule(dA, x, db, sch, 1); torch.cuda.synchronize()
        ok = np.array_equal(x.cpu().numpy(), w)
        t = time.perf_counter(); ops.csr_gs_schedule(dA, x, db, sch, 2); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 2
        dg = float("nan")
        if single_max == 2048:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                g = ops.CapturedGraph()
                with g:
                    ops.csr_gs_schedule(dA, x, db, sch, 1)
                g.launch(); st.synchronize()
                t = time.perf_counter(); g.launch(); g.launch(); st.synchronize(); dg = (time.perf_counter() - t) / 2
        print("%-14s n=%8d sets=%6d max_set=%5d single_max=%-10d bit-exact=%s  GPU %.3f ms/sweep (hipGraph replay %.3f)  CPU %.3f ms/sweep"
              % (name, n, sch.nsets, sch.max_set, single_max, ok, dt * 1e3, dg * 1e3, dc * 1e3), flush=True)
ops.tune_set("gs_single_max", 2048)
