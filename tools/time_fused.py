#!/usr/bin/env python3
"""Time the fused smoothing pass (lmg_stencil_smooth) against separate stencil sweeps."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, scipy.sparse as sp
from learnmultigrid_amd import ops, problems as P
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--seg", default="0")
ap.add_argument("--pf", default="2")
a = ap.parse_args()
ops.FUSED_MIN_ROWS = 0          # time the fused pass at every size
cases = []
A, _ = P.poisson_2d_structured(a.size); cases.append(("5pt %d^2" % (a.size + 1), A))
Pm = P.tensor_interpolator_2d(a.size + 1)
G = sp.csr_matrix(Pm.T @ A @ Pm); G.sort_indices(); cases.append(("9pt %d^2" % (a.size // 2 + 1), G))


def timeit(f, reps=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for lab, M in cases:
    dA = ops.DeviceCSR.from_scipy(M, "cuda:0"); dA.pack(); n = M.shape[0]
    x = torch.rand(n, dtype=torch.float64, device="cuda:0"); b = torch.rand_like(x)
    y = torch.empty_like(x); y2 = torch.empty_like(x); r = torch.empty_like(x); r2 = torch.empty_like(x)

    def separate(S, resid, zero):
        src = x
        bufs = [y, y2]
        for s in range(S):
            ops.csr_jacobi(dA, src, b, 0.8, bufs[s % 2]); src = bufs[s % 2]
        if resid:
            ops.csr_residual_norm2(dA, src, b, r, None, None)
        return src
    t1 = timeit(lambda: ops.csr_jacobi(dA, x, b, 0.8, y))
    print("%s: one sweep %.4f ms" % (lab, t1))
    for S, resid, zero in ((3, False, False), (3, True, False), (2, True, True), (3, False, True), (1, True, False), (2, False, False)):
        t_sep = timeit(lambda: separate(S, resid, zero))
        ref = separate(S, resid, zero).clone() if not zero else None
        rref = r.clone()
        for seg in [int(v) for v in a.seg.split(",")]:
            for pf in [int(v) for v in a.pf.split(",")]:
                ops.tune_set("fused_seg_lines", seg); ops.tune_set("fused_pf", pf)
                out = torch.empty_like(x)
                f = lambda: ops.stencil_smooth(dA, None if zero else x, b, 0.8, S, out, r2 if resid else None)
                t = timeit(f)
                ok = "" if zero or (torch.equal(out, ref) and (not resid or torch.equal(r2, rref))) else " MISMATCH"
                moved = n * (1 + (0 if zero else 8) + 8 + 8 + (8 if resid else 0))
                print("   S=%d resid=%d zero=%d seg=%d pf=%d: fused %.4f ms (%.0f GB/s compulsory) vs separate %.4f ms%s"
                      % (S, resid, zero, seg, pf, t, moved / t / 1e6, t_sep, ok))
    ops.tune_set("fused_seg_lines", 0); ops.tune_set("fused_pf", 0)
