import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from learnmultigrid_amd import ops, problems as P
m=4096
A, _ = P.poisson_2d_structured(m); n = A.shape[0]
dA = ops.DeviceCSR.from_scipy(A, "cuda:0"); dA.pack()
rng = np.random.default_rng(1)
x = torch.from_numpy(rng.standard_normal(n)).cuda(); b = torch.from_numpy(rng.standard_normal(n)).cuda()
ops.tune_set("gsw_lds", 1); ops.tune_set("gsw_max_sweeps", 1)
ops.stencil_gs(dA, x, b, 1); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): ops.stencil_gs(dA, x, b, 1)
torch.cuda.synchronize(); print(os.environ.get("LMG_LIB_PATH","product")[-12:], "LDS band sweep %.3f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
