#!/usr/bin/env python3
"""A/B the sweep-kernel tile variants in ONE process, interleaved (DVFS and box-to-box
spread are a few per cent, larger than most variant differences).

    python tools/tune_sweep.py --size 4096 --variants 1,2,3,4,5 --reps 15 [--matrix A0|A1|P0|R0]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--variants", default="1,2,3,4,5,6")
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--inner", type=int, default=10)
    ap.add_argument("--matrix", default="A0")
    ap.add_argument("--mode", default="jacobi", choices=["jacobi", "residual", "spmv"])
    ap.add_argument("--ju", default="", help="packed kernel unroll factors to compare, e.g. 1,2,3,5 (variant = -ju)")
    ap.add_argument("--packed", type=int, default=0, help="1: also time the packed twin (reported as variant -1)")
    args = ap.parse_args()
    import torch
    import scipy.sparse as sp
    from learnmultigrid_amd import ops, problems as P
    dev = "cuda:0"
    m = args.size
    A, rhs = P.poisson_2d_structured(m)
    if os.environ.get("LMG_SELL_MIN_AVG"):
        ops.SELL_MIN_AVG = float(os.environ["LMG_SELL_MIN_AVG"])
    if args.matrix == "A0":
        M = A
    elif args.matrix == "V0":                         # variable coefficients: 5 entries per row, all values distinct
        M = P.variable_coeff_poisson_2d_structured(m, seed=44)[0]
    elif args.matrix == "J0":                         # jittered mesh: 7 entries per row, all values distinct
        M = P.jittered_poisson_2d(m, seed=42)[0] if hasattr(P, "jittered_poisson_2d") else A
    elif args.matrix in ("L1", "L2"):
        # learned-like (pseudo-L2 support, perturbed weights) Galerkin operators: 25 / ~48 entries per row
        Av, _ = P.variable_coeff_poisson_2d_structured(m, seed=44)
        M = Av
        for li, sz in enumerate(P.level_sizes(m + 1, 3)[:2 if args.matrix == "L2" else 1]):
            l2 = P.pseudo_l2_interpolator_1d(sz)
            Q = P.learned_like(sp.kron(l2, l2).tocsr(), 43 + li)
            M = sp.csr_matrix(Q.T @ M @ Q)
        M.sort_indices()
    else:
        Pm = P.tensor_interpolator_2d(m + 1)
        M = {"A1": lambda: sp.csr_matrix(Pm.T @ A @ Pm), "P0": lambda: Pm,
             "R0": lambda: Pm.T.tocsr()}[args.matrix]()
        M.sort_indices()
    dM = ops.DeviceCSR.from_scipy(M, dev)
    n, nc = M.shape
    x = torch.rand(nc, dtype=torch.float64, device=dev)
    b = torch.rand(n, dtype=torch.float64, device=dev)
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    mode = args.mode if n == nc else "spmv"
    nbytes = 12 * M.nnz + 4 * (n + 1) + 8 * nc + 8 * n + (8 * n if mode != "spmv" else 0)

    def launch():
        if mode == "jacobi":
            ops.csr_jacobi(dM, x, b, 0.8, y)
        elif mode == "residual":
            ops.csr_residual_norm2(dM, x, b, y, None, None)
        else:
            ops.csr_spmv(dM, x, y, 1.0, 0.0)

    variants = [int(v) for v in args.variants.split(",")]
    jus = [int(v) for v in args.ju.split(",")] if args.ju else []
    if args.packed:
        pk = dM.pack()
        print("twin: %s, %d bytes (csr %d)" % (type(pk).__name__, pk.bytes(), dM.bytes()))
        variants = ([-j for j in jus] if jus else [-1]) + variants
    ops.set_packed_enabled(False)
    times = {v: [] for v in variants}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(args.reps + 1):
        for v in variants:
            ops.set_packed_enabled(v < 0)
            if v >= 0:
                ops.tune_set("sweep_variant", v)
            elif jus:
                ops.tune_set("pcsr_ju", -v)
            launch()
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(args.inner):
                launch()
            ev1.record()
            torch.cuda.synchronize()
            if rep > 0:
                times[v].append(ev0.elapsed_time(ev1) / args.inner)
    print("matrix %s %dx%d nnz %d (%.2f/row) mode %s bytes %d" % (args.matrix, n, nc, M.nnz, M.nnz / n, mode, nbytes))
    for v in variants:
        t = np.array(times[v])
        med = float(np.median(t))
        print("variant %2d  median %.4f ms  min %.4f  max %.4f   %.0f GB/s (%.1f%% of 8 TB/s)"
              % (v, med, t.min(), t.max(), nbytes / med / 1e6, nbytes / med / 1e6 / 80.0))


if __name__ == "__main__":
    main()
