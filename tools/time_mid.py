#!/usr/bin/env python3
"""Which smoothing pass for mid-sized grid-stencil levels (1 - 4 M rows): separate sweeps, the LDS-tiled pass
(stencil_tile.hip) or the register-blocked pass (stencil_fused.hip)?  hipGraph chains of 20 dependent launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, scipy.sparse as sp
from learnmultigrid_amd import ops, problems as P
from time_small import chain_us, dev, st          # noqa: E402  (prints its own table first)

for side, kind in ((1025, "5pt"), (1449, "5pt"), (2049, "5pt"), (1449, "9pt"), (2049, "9pt")):
    if kind == "5pt":
        A = P.poisson_2d_structured(side - 1)[0]
    else:
        Af = P.poisson_2d_structured(2 * (side - 1))[0]
        Pf = P.tensor_interpolator_2d(2 * (side - 1) + 1)
        A = sp.csr_matrix(Pf.T @ Af @ Pf); A.sort_indices()
    with torch.cuda.stream(st):
        dA = ops.DeviceCSR.from_scipy(A, dev); dA.pack()
        n = A.shape[0]
        x = torch.rand(n, dtype=torch.float64, device=dev); b = torch.rand_like(x); y = torch.empty_like(x); r = torch.empty_like(x)
    t_sw = chain_us(lambda: ops.csr_jacobi(dA, x, b, 0.8, y), chain=20)
    t_rs = chain_us(lambda: ops.csr_residual_norm2(dA, x, b, r, None, None), chain=20)
    out = ["%s %d^2: sweep %.1f us, residual %.1f us" % (kind, side, t_sw, t_rs)]
    save = ops.FUSED_MIN_ROWS
    for lab, fm in (("tile", 1 << 40), ("reg", 0)):
        ops.FUSED_MIN_ROWS = fm
        pre = chain_us(lambda: ops.stencil_smooth(dA, x, b, 0.8, 3, y, r), chain=20)
        post = chain_us(lambda: ops.stencil_smooth(dA, x, b, 0.8, 3, y, None), chain=20)
        out.append("%s: 3 sweeps + residual %.1f us (separate %.1f), 3 sweeps %.1f us (separate %.1f)"
                   % (lab, pre, 3 * t_sw + t_rs, post, 3 * t_sw))
    ops.FUSED_MIN_ROWS = save
    print("   ".join(out))
