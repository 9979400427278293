#!/usr/bin/env python3
"""Exact forward Gauss-Seidel: pipelined wavefront kernel (gs_wave.hip) vs the level-scheduled executors, and
bitwise agreement of the two (and with the CPU oracle at small sizes)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, scipy.sparse as sp
from learnmultigrid_amd import ops, problems as P
ap = argparse.ArgumentParser(); ap.add_argument("--sizes", default="64,512,1024"); ap.add_argument("--level", action="store_true")
a = ap.parse_args()
for m in [int(v) for v in a.sizes.split(",")]:
    A, rhs = P.poisson_2d_structured(m)
    cases = [("5pt %d^2" % (m + 1), A)]
    if m <= 2048:
        Pm = P.tensor_interpolator_2d(m + 1)
        G = sp.csr_matrix(Pm.T @ A @ Pm); G.sort_indices(); cases.append(("9pt %d^2" % (m // 2 + 1), G))
    for lab, M in cases:
        n = M.shape[0]
        dA = ops.DeviceCSR.from_scipy(M, "cuda:0"); dA.pack()
        assert ops.stencil_gs_available(dA), lab
        rng = np.random.default_rng(1)
        x0 = torch.from_numpy(rng.standard_normal(n)).cuda(); b = torch.from_numpy(rng.standard_normal(n)).cuda()
        x = x0.clone()
        ops.stencil_gs(dA, x, b, 1); torch.cuda.synchronize()
        ops.stencil_gs_check(dA)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps): ops.stencil_gs(dA, x, b, 1)
        torch.cuda.synchronize(); tw = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps): ops.stencil_gs(dA, x, b, 3)
        torch.cuda.synchronize(); tw3 = (time.perf_counter() - t0) / reps
        ops.tune_set("gsw_max_sweeps", 1)
        t0 = time.perf_counter()
        for _ in range(reps): ops.stencil_gs(dA, x, b, 3)
        torch.cuda.synchronize(); tw3s = (time.perf_counter() - t0) / reps
        ops.tune_set("gsw_max_sweeps", 4)
        ops.stencil_gs_check(dA)
        msg = "%s: wavefront %.3f ms/sweep; 3 sweeps pipelined in one launch %.3f ms (one launch each: %.3f ms)" % (lab, tw * 1e3, tw3 * 1e3, tw3s * 1e3)
        if a.level or m <= 1024:
            pat = sp.csr_matrix((np.ones(M.nnz, dtype=np.int8), M.indices, M.indptr), shape=M.shape)
            sched = ops.build_gs_schedule(pat, "lexicographic", "cuda:0")
            x1 = x0.clone(); x2 = x0.clone()
            ops.set_wavefront_gs_enabled(False)
            ops.csr_gs_schedule(dA, x1, b, sched, 2); torch.cuda.synchronize()
            t0 = time.perf_counter()
            ops.csr_gs_schedule(dA, x1, b, sched, 1); torch.cuda.synchronize()
            tl = time.perf_counter() - t0
            ops.set_wavefront_gs_enabled(True)
            ops.stencil_gs(dA, x2, b, 3); torch.cuda.synchronize()
            msg += "   level schedule %.3f ms/sweep (%d sets)   bitwise equal after 3 sweeps: %s" % (tl * 1e3, sched.nsets, torch.equal(x1, x2))
        print(msg, flush=True)
