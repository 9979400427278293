#!/usr/bin/env python3
"""Time prolongation (u += P e) and restriction (r_c = R r) of the tensor-product transfer between
4097^2 and 2049^2: row patterns with a column-base map vs packed CSR vs plain CSR."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, scipy.sparse as sp
from learnmultigrid_amd import ops, problems as P
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=4096); a = ap.parse_args()
Pm = P.tensor_interpolator_2d(a.size + 1).tocsr()
Rm = sp.csr_matrix(Pm.T)


def timeit(f, reps=30):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for lab, M, beta in (("prolongation u += P e", Pm, 1.0), ("restriction r_c = R r", Rm, 0.0)):
    x = torch.rand(M.shape[1], dtype=torch.float64, device="cuda:0")
    y = torch.rand(M.shape[0], dtype=torch.float64, device="cuda:0")
    vec_bytes = 8 * M.shape[1] + (16 if beta else 8) * M.shape[0]
    res = []
    for mode in ("grid patterns", "packed CSR", "CSR"):
        ops.set_grid_maps_enabled(mode == "grid patterns")
        dM = ops.DeviceCSR.from_scipy(M, "cuda:0")
        tw = dM.pack() if mode != "CSR" else None
        for v in ((0, 1, 2, 3, 4) if mode == "grid patterns" else (0,)):
            ops.tune_set("rpat_variant", v)
            t = timeit(lambda: ops.csr_spmv(dM, x, y, 1.0, beta))
            moved = vec_bytes + (tw.bytes() if tw is not None else dM.bytes())
            res.append("%s%s %.4f ms (%.0f GB/s moved)" % (mode, " v%d" % v if mode == "grid patterns" else "", t, moved / t / 1e6))
        ops.tune_set("rpat_variant", 0)
    ops.set_grid_maps_enabled(True)
    print(lab + ":\n    " + "\n    ".join(res))
