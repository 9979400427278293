import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from learnmultigrid_amd import ops, problems as P
from oracle import kernels as K
m = int(sys.argv[1]) if len(sys.argv) > 1 else 16
A, rhs = P.poisson_2d_structured(m)
A = K.as_csr(A); n = A.shape[0]; W = m + 1
dA = ops.DeviceCSR.from_scipy(A, "cuda:0"); dA.pack()
rng = np.random.default_rng(1)
x0 = rng.standard_normal(n); b = rng.standard_normal(n)
x = torch.from_numpy(x0.copy()).cuda()
ops.stencil_gs(dA, x, torch.from_numpy(b).cuda(), 1); torch.cuda.synchronize()
want = x0.copy()
K.lib().orc_csr_gs_forward(n, A.indptr, A.indices, A.data, want, b, 1)
got = x.cpu().numpy()
bad = np.flatnonzero(got != want)
print("n", n, "W", W, "mismatches", bad.size)
for i in bad[:20]:
    print("  i=%d (y=%d,x=%d) got %.17g want %.17g old %.17g" % (i, i // W, i % W, got[i], want[i], x0[i]))
