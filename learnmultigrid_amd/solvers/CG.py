"""Conjugate gradients -- mirrors learn_multigrid/solvers/CG.py:12-50 (same signature, the
initial residual is track_res[0], one entry per iteration after it, stop on ||r|| <= error).
The reference's CG only runs under NumPy < 1.23 (`np.asscalar`, CG.py:30,:32,:47); the
arithmetic is the textbook recurrence and is executed here by the HIP kernels (SpMV, dot,
axpby), two 8-byte D2H reads per iteration for alpha and beta.

Build-only extension (keyword-only): `preconditioner=Hierarchy` turns it into multigrid-
preconditioned CG -- one V(nu,nu) Jacobi cycle from a zero guess per application (the
"step after the hot path" of SURVEY.md section 8 f4)."""
import math

import numpy as np
import torch

from .. import ops
from ..ops import F64
from .Solver import IterativeSolver, on_device


class CG(IterativeSolver):

    def __init__(self, matrix, rhs, **kw):
        super().__init__(matrix, rhs, **kw)
        self._log("Selected CG")
        self.label = "CG"

    @on_device
    def solve(self, max_iterations=1000, error=1e-08, initial_guess=None, *, preconditioner=None,
              precond_steps=2, precond_omega=0.8):
        A = self._device_matrix()
        A.pack()
        n = self.dim
        dev = self._device
        b = self._to_device(self.rhs)
        x = torch.zeros(n, dtype=F64, device=dev) if initial_guess is None else self._to_device(initial_guess)
        r = torch.empty_like(x)
        Ap = torch.empty_like(x)
        part = torch.empty(ops.partials_count(n), dtype=F64, device=dev)
        s = torch.zeros(1, dtype=F64, device=dev)
        ops.csr_residual_norm2(A, x, b, r, part, s)                    # CG.py:21-22
        self.residual = math.sqrt(s.item())
        track = [self.residual]
        H = preconditioner

        def apply_M(src, dst):
            if H is None:
                ops.copy(src, dst)
                return
            fine = H.levels[0]
            ops.copy(src, fine.b)
            H.cycle("Jacobi", precond_steps, precond_omega, x_is_zero=True)
            ops.copy(fine.x, dst)

        z = torch.empty_like(x)
        apply_M(r, z)
        p = z.clone()
        ops.dot(r, z, part, s)
        rz = s.item()
        for _ in range(max_iterations):
            self.iterations += 1
            ops.csr_spmv(A, p, Ap, 1.0, 0.0)                           # :31
            ops.dot(p, Ap, part, s)
            pAp = s.item()
            alpha = rz / pAp                                           # :34
            ops.axpby(alpha, p, 1.0, x)                                # :35
            ops.axpby(-alpha, Ap, 1.0, r)                              # :37
            ops.dot(r, r, part, s)
            self.residual = math.sqrt(s.item())                        # :41
            track.append(self.residual)
            if self.residual <= error:
                self._log("Reached convergence")
                break
            apply_M(r, z)
            if H is None:
                rz_new = s.item()
            else:
                ops.dot(r, z, part, s)
                rz_new = s.item()
            beta = rz_new / rz                                         # :47
            rz = rz_new
            ops.axpby(1.0, z, beta, p)                                 # :48
        self.solution = self._column(x)
        self.residual_vector = self._column(r)
        self.track_res = np.array(track, dtype=float).reshape(-1, 1)
