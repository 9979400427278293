#!/usr/bin/env python3
"""Per-step cost and band-to-band lag of the LDS-staged Gauss-Seidel kernel: 5-point operators on nl x 4097 grids."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, scipy.sparse as sp
from learnmultigrid_amd import ops
W = 4097
for nl in (64, 128, 256, 512, 1024):
    I1, I2 = sp.identity(nl), sp.identity(W)
    T1 = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(nl, nl)); T2 = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(W, W))
    A = (sp.kron(T1, I2) + sp.kron(I1, T2)).tocsr(); A.sort_indices()
    n = A.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, "cuda:0"); dA.pack()
    assert ops.stencil_gs_available(dA) and dA.stencil.W == W
    x = torch.rand(n, dtype=torch.float64, device="cuda:0"); b = torch.rand_like(x)
    out = []
    for lds in (1, 0):
        ops.tune_set("gsw_lds", lds); ops.tune_set("gsw_max_sweeps", 1)
        ops.stencil_gs(dA, x, b, 1); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): ops.stencil_gs(dA, x, b, 1)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 3 * 1e3)
    ops.tune_set("gsw_lds", -1); ops.tune_set("gsw_max_sweeps", 4)
    print("%4d lines (%2d bands): LDS %.3f ms, register %.3f ms" % (nl, (nl + 63) // 64, out[0], out[1]), flush=True)
