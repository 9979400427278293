#!/usr/bin/env python3
"""Cold timing of the twin construction steps of DeviceCSR.pack() (first call of a fresh process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, scipy.sparse as sp
from learnmultigrid_amd import ops, problems as P
m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
A, _ = P.poisson_2d_structured(m)
Pm = sp.csr_matrix(P.tensor_interpolator_2d(m + 1))
torch.zeros(1, device=dev); torch.cuda.synchronize()
dA = ops.DeviceCSR.from_scipy(A, dev); dP = ops.DeviceCSR.from_scipy(Pm, dev)


def tick(label, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    print("%-40s %.3f s" % (label, time.perf_counter() - t0)); return r


R = tick("RowPatterns.from_csr(A)", lambda: ops.RowPatterns.from_csr(dA))
tick("StencilTwin.from_patterns(A)", lambda: ops.StencilTwin.from_patterns(R, dA.shape))
gm = ops.RowPatterns.grid_map_candidates(dP.shape)[0]
RP = tick("RowPatterns.from_csr(P, grid map)", lambda: ops.RowPatterns.from_csr(dP, gm))
tick("ProlongTwin.from_patterns(P)", lambda: ops.ProlongTwin.from_patterns(RP, dP.shape))
dR = tick("transpose(P)", lambda: dP.transpose())
gmr = ops.RowPatterns.grid_map_candidates(dR.shape)[0]
RR = tick("RowPatterns.from_csr(R, grid map)", lambda: ops.RowPatterns.from_csr(dR, gmr))
tick("RestrictTwin.from_patterns(R)", lambda: ops.RestrictTwin.from_patterns(RR, dR.shape))
tick("RowPatterns.from_csr(A) again", lambda: ops.RowPatterns.from_csr(dA))
