#!/usr/bin/env python3
"""Randomised bitwise comparison of the fused smoothing passes (register and tiled kernels, with and without the transfers
folded in, 1 - 3 sweeps) with the separate launches they replace, on grids of random sizes.
    python tools/fuzz_fused.py --cases 40 --seed 1"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learnmultigrid_amd import ops, problems as P
from learnmultigrid_amd.hierarchy import Hierarchy

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=30)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max", type=int, default=1400)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda:0")
bad = 0
checked = {}
for case in range(a.cases):
    m = 2 * int(rng.integers(33, a.max // 2))            # even number of cells: odd number of nodes per side
    A, _ = P.poisson_2d_structured(m)
    H = Hierarchy(A, P.geometric_hierarchy_2d(m + 1, 3), dev)
    for li in (0, 1):                                    # 5-point fine level, 9-point Galerkin level
        lev = H.levels[li]
        fa, n, nc = lev.A, lev.A.shape[0], lev.P.shape[1]
        x = torch.from_numpy(rng.standard_normal(n)).to(dev); b = torch.from_numpy(rng.standard_normal(n)).to(dev)
        e = torch.from_numpy(rng.standard_normal(nc)).to(dev)
        for kind in ("reg", "tile"):
            ops.FUSED_MIN_ROWS = 0 if kind == "reg" else 1 << 40
            ops.FUSED_TRANSFER_MIN_ROWS = 0
            if ops._fused_kind(fa) != kind:
                continue
            for S in (1, 2, 3):
                # reference: separate launches
                xs = x.clone(); t = torch.empty_like(x)
                for _ in range(S):
                    ops.csr_jacobi(fa, xs, b, 0.8, t); xs, t = t, xs
                r = torch.empty_like(x); ops.csr_residual_norm2(fa, xs, b, r, None, None)
                bc = torch.zeros(nc, dtype=torch.float64, device=dev); ops.csr_spmv(lev.R, r, bc)
                xp = x.clone(); ops.csr_spmv(lev.P, e, xp, 1.0, 1.0)
                t2 = torch.empty_like(x)
                for _ in range(S):
                    ops.csr_jacobi(fa, xp, b, 0.8, t2); xp, t2 = t2, xp
                for fast in (1, 0):
                    ops.tune_set("fused_fast", fast)
                    y = torch.empty_like(x); rr = torch.empty_like(x)
                    ops.stencil_smooth(fa, x, b, 0.8, S, y, rr)
                    ok = torch.equal(y, xs) and torch.equal(rr, r)
                    y2 = torch.empty_like(x); ops.stencil_smooth(fa, x, b, 0.8, S, y2, None)
                    ok = ok and torch.equal(y2, xs)
                    if ops.stencil_smooth_restrict_available(fa, lev.R):
                        y3 = torch.empty_like(x); bc2 = torch.full((nc,), 7.0, dtype=torch.float64, device=dev)
                        ops.stencil_smooth(fa, x, b, 0.8, S, y3, None, restrict=(lev.R, bc2))
                        ok = ok and torch.equal(y3, xs) and torch.equal(bc2, bc)
                    if ops.stencil_smooth_prolong_available(fa, lev.P):
                        y4 = torch.empty_like(x); ops.stencil_smooth(fa, x, b, 0.8, S, y4, None, prolong=(lev.P, e))
                        ok = ok and torch.equal(y4, xp)
                    checked[(kind, li)] = checked.get((kind, li), 0) + 1
                    if not ok:
                        bad += 1
                        print("MISMATCH m=%d level=%d kind=%s S=%d fast=%d" % (m, li, kind, S, fast), flush=True)
                ops.tune_set("fused_fast", 1)
    print("case %d: m = %d ok so far, mismatches %d" % (case, m, bad), flush=True)
    del H
print("done: %d mismatches; comparisons per (kernel, level): %s" % (bad, checked))
sys.exit(1 if bad else 0)
