#!/usr/bin/env python3
"""Exact forward Gauss-Seidel, one sweep: bands staged through LDS (gs_band_lds_kernel) vs the register wavefront kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learnmultigrid_amd import ops, problems as P
import scipy.sparse as sp
NINE = len(sys.argv) > 2 and sys.argv[2] == "9pt"       # python tools/time_gs_lds.py 512,1024,2048 9pt: the 9-point Galerkin operator
for m in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512,1024,2048,4096").split(",")]:
    if NINE:
        Af = P.poisson_2d_structured(2 * m)[0]; Pf = P.tensor_interpolator_2d(2 * m + 1)
        A = sp.csr_matrix(Pf.T @ Af @ Pf); A.sort_indices()
    else:
        A, _ = P.poisson_2d_structured(m)
    n = A.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, "cuda:0"); dA.pack()
    rng = np.random.default_rng(1)
    x0 = torch.from_numpy(rng.standard_normal(n)).cuda(); b = torch.from_numpy(rng.standard_normal(n)).cuda()
    res = {}
    for lds in (0, 1):
        ops.tune_set("gsw_lds", lds); ops.tune_set("gsw_max_sweeps", 1)
        x = x0.clone()
        ops.stencil_gs(dA, x, b, 1); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): ops.stencil_gs(dA, x, b, 1)
        torch.cuda.synchronize(); res[lds] = ((time.perf_counter() - t0) / 3, x)
        ops.stencil_gs_check(dA)
    ops.tune_set("gsw_max_sweeps", 4); ops.tune_set("gsw_lds", 0)
    x = x0.clone(); ops.stencil_gs(dA, x, b, 3); torch.cuda.synchronize()
    t0 = time.perf_counter(); ops.stencil_gs(dA, x, b, 3); torch.cuda.synchronize(); t3 = time.perf_counter() - t0
    ops.tune_set("gsw_lds", 1); ops.tune_set("gsw_lds_multi", 1)
    xl = x0.clone(); ops.stencil_gs(dA, xl, b, 3); torch.cuda.synchronize()
    t0 = time.perf_counter(); ops.stencil_gs(dA, xl, b, 3); torch.cuda.synchronize(); t3l = time.perf_counter() - t0
    ops.stencil_gs_check(dA)
    same3 = torch.equal(xl, x)
    ops.tune_set("gsw_lds", -1)
    print("   3 sweeps pipelined, LDS bands: %.3f ms (equal bits: %s)" % (t3l * 1e3, same3))
    print(("9pt" if NINE else "5pt") + " %d^2: one sweep register wavefront %.3f ms, LDS bands %.3f ms (equal bits: %s); 3 sweeps pipelined (register kernel) %.3f ms"
          % (m + 1, res[0][0] * 1e3, res[1][0] * 1e3, torch.equal(res[0][1], res[1][1]), t3 * 1e3), flush=True)
