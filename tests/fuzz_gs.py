#!/usr/bin/env python3
"""Randomised bitwise comparison of the LDS-staged Gauss-Seidel bands (single and pipelined sweeps, 5- and 9-point operators)
with the register wavefront kernel and -- on the small cases -- with the sequential CPU sweep of oracle/lmg_oracle.c.
    python tests/fuzz_gs.py --cases 40 --seed 1   (under tests/: it uses the oracle as the checker)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from learnmultigrid_amd import ops, problems as P
from oracle import kernels as K

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=30)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max", type=int, default=1500)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = "cuda:0"
bad = 0
for case in range(a.cases):
    m = int(rng.integers(70, a.max))
    nine = bool(rng.integers(0, 2))
    if nine:
        Af = P.poisson_2d_structured(2 * m)[0]; Pf = P.tensor_interpolator_2d(2 * m + 1)
        A = sp.csr_matrix(Pf.T @ Af @ Pf); A.sort_indices()
    else:
        A = P.poisson_2d_structured(m)[0]
    A = K.as_csr(A)
    n = A.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, dev); dA.pack()
    x0 = rng.standard_normal(n); b = rng.standard_normal(n)
    db = torch.from_numpy(b).to(dev)
    sweeps = int(rng.integers(1, 5))
    res = {}
    for lds in (1, 0):
        ops.tune_set("gsw_lds", lds)
        x = torch.from_numpy(x0.copy()).to(dev)
        for rep in range(3):                      # repeated: a protocol race would not show every time
            xx = x.clone(); ops.stencil_gs(dA, xx, db, sweeps)
            if rep == 0: res[lds] = xx
            elif not torch.equal(xx, res[lds]):
                bad += 1; print("NOT REPRODUCIBLE m=%d nine=%s lds=%d sweeps=%d" % (m, nine, lds, sweeps), flush=True)
        ops.stencil_gs_check(dA)
    ok = torch.equal(res[0], res[1])
    if n <= 400000:
        want = x0.copy(); K.lib().orc_csr_gs_forward(n, A.indptr, A.indices, A.data, want, b, sweeps)
        ok = ok and np.array_equal(res[1].cpu().numpy(), want)
    if not ok:
        bad += 1; print("MISMATCH m=%d nine=%s sweeps=%d" % (m, nine, sweeps), flush=True)
    print("case %d: %s %d^2, %d sweep(s): %s" % (case, "9pt" if nine else "5pt", m + 1, sweeps, "ok" if ok else "BAD"), flush=True)
ops.tune_set("gsw_lds", -1)
print("done: %d problems" % bad)
sys.exit(1 if bad else 0)
