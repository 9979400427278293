#!/usr/bin/env python3
"""Per-kernel cost of the small-level kernels inside a hipGraph (a chain of 40 dependent launches of ONE kernel,
replayed): what a launch of the cycle's tail really costs, next to tools/probe/launch_floor.hip (1.9 us for a
trivial kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, scipy.sparse as sp
from learnmultigrid_amd import ops, problems as P

dev = torch.device("cuda:0")
st = torch.cuda.Stream(dev)


def chain_us(f, chain=40, reps=20):
    with torch.cuda.stream(st):
        f()
        g = ops.CapturedGraph()
        with g:
            for _ in range(chain):
                f()
        for _ in range(3):
            g.launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            g.launch()
        e1.record(st)
        e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * chain)


for side in (257, 513, 1025):
    A5, _ = P.poisson_2d_structured(2 * (side - 1))
    Pm = P.tensor_interpolator_2d(2 * (side - 1) + 1)
    G = sp.csr_matrix(Pm.T @ A5 @ Pm); G.sort_indices()            # 9-point Galerkin operator on side^2
    Pc = P.tensor_interpolator_2d(side)                             # side^2 -> ((side+1)/2)^2
    with torch.cuda.stream(st):
        dA = ops.DeviceCSR.from_scipy(G, dev); dA.pack()
        dP = ops.DeviceCSR.from_scipy(sp.csr_matrix(Pc), dev); dP.pack()
        dR = dP.transpose(); dR.pack()
        n, nc = G.shape[0], Pc.shape[1]
        x = torch.rand(n, dtype=torch.float64, device=dev); b = torch.rand_like(x); y = torch.empty_like(x); r = torch.empty_like(x)
        xc = torch.rand(nc, dtype=torch.float64, device=dev); yc = torch.empty_like(xc)
        dinv = torch.rand_like(x)
    res = [
        ("vmul", chain_us(lambda: ops.vmul(0.8, dinv, b, y))),
        ("axpby", chain_us(lambda: ops.axpby(1.0, x, 1.0, y))),
        ("jacobi sweep", chain_us(lambda: ops.csr_jacobi(dA, x, b, 0.8, y))),
        ("residual", chain_us(lambda: ops.csr_residual_norm2(dA, x, b, r, None, None))),
        ("restriction", chain_us(lambda: ops.csr_spmv(dR, x, yc))),
        ("prolongation+", chain_us(lambda: ops.csr_spmv(dP, xc, y, 1.0, 1.0))),
    ]
    print("%d^2: " % side + "   ".join("%s %.2f us" % kv for kv in res))
    # the LDS-tiled fused passes of the same level (what the cycle launches): pre = 3 sweeps from zero + residual,
    # post = 3 sweeps
    for rows in (16, 32):
        ops.tune_set("tile_rows", rows)
        pre = chain_us(lambda: ops.stencil_smooth(dA, None, b, 0.8, 3, y, r))
        post = chain_us(lambda: ops.stencil_smooth(dA, x, b, 0.8, 3, y, None))
        print("      tiled, %d-line tiles: pre-smoothing pass %.2f us   post-smoothing pass %.2f us" % (rows, pre, post))
    ops.tune_set("tile_rows", 0)
