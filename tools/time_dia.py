#!/usr/bin/env python3
"""Time the fused smoothing pass on a variable-coefficient / jittered fine operator (lmg_dia_smooth) against the
separate packed-CSR sweeps."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnmultigrid_amd import ops, problems as P
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--problem", default="varcoeff")
a = ap.parse_args()
A, _ = (P.variable_coeff_poisson_2d_structured(a.size, seed=44) if a.problem == "varcoeff" else P.jittered_poisson_2d(a.size, seed=42))
dA = ops.DeviceCSR.from_scipy(A, "cuda:0"); dA.pack(); n = A.shape[0]
assert dA.dia is not None
x = torch.rand(n, dtype=torch.float64, device="cuda:0"); b = torch.rand_like(x)
y = torch.empty_like(x); y2 = torch.empty_like(x); r = torch.empty_like(x)


def timeit(f, reps=10):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def separate():
    ops.csr_jacobi(dA, x, b, 0.8, y); ops.csr_jacobi(dA, y, b, 0.8, y2); ops.csr_jacobi(dA, y2, b, 0.8, y)
    ops.csr_residual_norm2(dA, y, b, r, None, None)


ns = dA.dia.nslots
print("%s %d^2: %d slots; separate 3 sweeps + residual %.4f ms" % (a.problem, a.size + 1, ns, timeit(separate)))
for rows in (32, 64):
    ops.tune_set("dia_rows", rows)
    t = timeit(lambda: ops.stencil_smooth(dA, x, b, 0.8, 3, y, r))
    t2 = timeit(lambda: ops.stencil_smooth(dA, x, b, 0.8, 3, y, None))
    moved = n * (8 * ns + 32)
    print("  tile rows %d: 3 sweeps + residual %.4f ms (%.0f GB/s compulsory), 3 sweeps %.4f ms" % (rows, t, moved / t / 1e6, t2))
