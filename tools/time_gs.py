#!/usr/bin/env python3
"""Time one exact (lexicographic) Gauss-Seidel sweep on the reference's typical problems."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learnmultigrid_amd import ops, problems as P
from oracle import kernels as K
for name, A in (("1-D ne=14960", P.poisson_1d_fd(14960)[0]), ("1-D ne=1024", P.poisson_1d_fd(1024)[0]),
                ("2-D 101^2", P.poisson_2d_structured(100)[0]), ("2-D 513^2", P.poisson_2d_structured(512)[0])):
    A = K.as_csr(A); n = A.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, "cuda:0")
    sch = ops.build_gs_schedule(A, "lexicographic", "cuda:0")
    rng = np.random.default_rng(0); x0 = rng.standard_normal(n); b = rng.standard_normal(n)
    x = torch.from_numpy(x0.copy()).cuda(); db = torch.from_numpy(b).cuda()
    ops.csr_gs_schedule(dA, x, db, sch, 1); torch.cuda.synchronize()
    w = x0.copy(); K.gs_forward(A, w, b, 1)
    ok = np.array_equal(x.cpu().numpy(), w)
    t = time.perf_counter(); ops.csr_gs_schedule(dA, x, db, sch, 3); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    t = time.perf_counter(); K.gs_forward(A, w, b, 3); dc = (time.perf_counter() - t) / 3
    print("%-14s n=%7d sets=%6d max_set=%5d  bit-exact=%s  GPU %.3f ms/sweep  CPU %.3f ms/sweep" % (name, n, sch.nsets, sch.max_set, ok, dt * 1e3, dc * 1e3))
