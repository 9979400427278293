#!/usr/bin/env python3
"""Time the grid-stencil sweeps (csrc/stencil.hip) against the row-pattern sweeps (csrc/rpat.hip) on the
5-point fine level and on its 9-point Galerkin coarsening; checks bitwise agreement of the two."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, scipy.sparse as sp
from learnmultigrid_amd import ops, problems as P
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--wgs", default="0")
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
cases = []
A, _ = P.poisson_2d_structured(a.size); cases.append(("5pt %d^2" % (a.size + 1), A))
Pm = P.tensor_interpolator_2d(a.size + 1)
G = sp.csr_matrix(Pm.T @ A @ Pm); G.sort_indices(); cases.append(("9pt %d^2" % (a.size // 2 + 1), G))


def timeit(f, reps):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for lab, M in cases:
    dA = ops.DeviceCSR.from_scipy(M, "cuda:0"); dA.pack(); n = M.shape[0]
    S = dA.stencil
    print("%s: stencil %s W=%s npat=%d umask=%s" % (lab, S is not None, S and S.W, dA.patterns.npat, S and bin(S.umask)))
    x = torch.rand(n, dtype=torch.float64, device="cuda:0"); b = torch.rand_like(x)
    y = torch.empty_like(x); y2 = torch.empty_like(x)
    part = torch.empty(ops.partials_count(n), dtype=torch.float64, device="cuda:0")
    n2 = torch.zeros(1, dtype=torch.float64, device="cuda:0")
    stored = n + 24 * n
    fs = (("jacobi", lambda o: ops.csr_jacobi(dA, x, b, 0.8, o)),
          ("residual", lambda o: ops.csr_residual_norm2(dA, x, b, o, part, n2)),
          ("spmv", lambda o: ops.csr_spmv(dA, x, o)))
    ops.set_stencil_enabled(False)
    res = []
    for name, f in fs:
        t = timeit(lambda: f(y), a.reps)
        res.append("%s %.4f ms (%.0f GB/s stored)" % (name, t, stored / t / 1e6))
    print("  rpat           : " + "   ".join(res))
    ops.set_stencil_enabled(True)
    if True:
        for w in [int(v) for v in a.wgs.split(",")]:
            ops.tune_set("stencil_wgs_per_cu", w)
            res = []
            for name, f in fs:
                ops.set_stencil_enabled(False); f(y)
                ops.set_stencil_enabled(True); f(y2)
                same = torch.equal(y, y2)
                t = timeit(lambda: f(y2), a.reps)
                res.append("%s %.4f ms (%.0f GB/s stored)%s" % (name, t, stored / t / 1e6, "" if same else " MISMATCH"))
            print("  stencil wgs=%d: %s" % (w, "   ".join(res)))
    ops.tune_set("stencil_wgs_per_cu", 0)
