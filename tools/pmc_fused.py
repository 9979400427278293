#!/usr/bin/env python3
"""The fine-level kernels of the cfg#4 cycle, five launches each, for counter passes:
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- python3 tools/pmc_fused.py      (and WRITE_SIZE, SQ counters)
Launches: fused 3 sweeps + residual, the same with the restriction folded in, fused 3 sweeps with the correction folded in, fused 3 sweeps, one stencil
sweep, restriction, prolongation -- on 4097^2 / 2049^2 (tensor-product transfer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, scipy.sparse as sp
from learnmultigrid_amd import ops, problems as P

m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
A, _ = P.poisson_2d_structured(m)
Pm = sp.csr_matrix(P.tensor_interpolator_2d(m + 1))
dA = ops.DeviceCSR.from_scipy(A, dev); dA.pack()
dP = ops.DeviceCSR.from_scipy(Pm, dev); dP.pack()
dR = dP.transpose(); dR.pack()
n, nc = A.shape[0], Pm.shape[1]
x = torch.rand(n, dtype=torch.float64, device=dev); b = torch.rand_like(x)
y = torch.empty_like(x); r = torch.empty_like(x)
e = torch.rand(nc, dtype=torch.float64, device=dev); bc = torch.empty_like(e)
for _ in range(5):
    ops.stencil_smooth(dA, x, b, 0.8, 3, y, r)
    ops.stencil_smooth(dA, x, b, 0.8, 3, y, None, restrict=(dR, bc))
    ops.stencil_smooth(dA, x, b, 0.8, 3, y, None, prolong=(dP, e))
    ops.stencil_smooth(dA, x, b, 0.8, 3, y, None)
    ops.csr_jacobi(dA, x, b, 0.8, y)
    ops.csr_spmv(dR, r, bc)
    ops.csr_spmv(dP, e, y, 1.0, 1.0)
torch.cuda.synchronize()
print("done", n, nc)
