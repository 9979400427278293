#!/usr/bin/env python3
"""Time the row-pattern sweep variants (lmg_tune_set("rpat_variant", v)) on the 5-point fine
level and on a 9-point Galerkin operator."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, scipy.sparse as sp
from learnmultigrid_amd import ops, problems as P
ap = argparse.ArgumentParser(); ap.add_argument("--variants", default="0,1,2,3,4"); ap.add_argument("--size", type=int, default=4096)
a = ap.parse_args()
cases = []
A, _ = P.poisson_2d_structured(a.size); cases.append(("5pt %d^2" % (a.size + 1), A))
A2, _ = P.poisson_2d_structured(a.size // 2 * 2); Pm = P.tensor_interpolator_2d(a.size + 1)
G = sp.csr_matrix(Pm.T @ A @ Pm); G.sort_indices(); cases.append(("9pt %d^2" % (a.size // 2 + 1), G))
for lab, M in cases:
    dA = ops.DeviceCSR.from_scipy(M, "cuda:0"); R = dA.pack(); n = M.shape[0]
    print("%s: %d patterns, %d entries, max len %d" % (lab, R.npat, R.nent, R.max_len))
    x = torch.rand(n, dtype=torch.float64, device="cuda:0"); b = torch.rand_like(x); y = torch.empty_like(x)
    part = torch.empty(ops.partials_count(n), dtype=torch.float64, device="cuda:0"); n2 = torch.zeros(1, dtype=torch.float64, device="cuda:0")
    for v in [int(t) for t in a.variants.split(",")]:
        ops.tune_set("rpat_variant", v)
        res = []
        for name, f in (("jacobi", lambda: ops.csr_jacobi(dA, x, b, 0.8, y)), ("residual", lambda: ops.csr_residual_norm2(dA, x, b, y, part, n2)), ("spmv", lambda: ops.csr_spmv(dA, x, y))):
            for _ in range(3): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): f()
            e1.record(); torch.cuda.synchronize()
            res.append("%s %.4f ms" % (name, e0.elapsed_time(e1) / 30))
        print("  variant %d: %s" % (v, "   ".join(res)))
    ops.tune_set("rpat_variant", 0)
