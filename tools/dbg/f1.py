import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from learnmultigrid_amd import ops, problems as P
A, _ = P.poisson_2d_structured(4096)
dA = ops.DeviceCSR.from_scipy(A, "cuda:0"); dA.pack(); n = A.shape[0]
x = torch.rand(n, dtype=torch.float64, device="cuda:0"); b = torch.rand_like(x); out = torch.empty_like(x); r = torch.empty_like(x)
for _ in range(5):
    ops.stencil_smooth(dA, x, b, 0.8, 3, out, None)
    ops.stencil_smooth(dA, x, b, 0.8, 3, out, r)
    ops.csr_jacobi(dA, x, b, 0.8, out)
torch.cuda.synchronize()
