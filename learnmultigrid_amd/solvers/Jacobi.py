"""Stand-alone Jacobi iteration -- mirrors learn_multigrid/solvers/Jacobi.py:15-37:
every iteration computes r = b - A x and ||r||, records it, stops on ||r|| <= error,
otherwise x += D^-1 r.  `omega` (keyword-only, default 1 = the reference) damps the
update.  Each iteration is two HIP kernels (residual+norm, sweep) and one 8-byte D2H
read for the stop test."""
import math

import numpy as np
import torch

from .. import ops
from ..ops import F64
from .Solver import IterativeSolver, on_device


class Jacobi(IterativeSolver):

    def __init__(self, matrix, rhs, **kw):
        super().__init__(matrix, rhs, **kw)
        self._log("Selected Jacobi")
        self.label = "Jacobi"

    @on_device
    def solve(self, max_iterations=1000, error=1e-12, initial_guess=None, *, omega=1.0):
        A = self._device_matrix()
        n = self.dim
        b = self._to_device(self.rhs)
        x = torch.zeros(n, dtype=F64, device=self._device) if initial_guess is None \
            else self._to_device(initial_guess)
        y = torch.empty_like(x)
        r = torch.empty_like(x)
        part = torch.empty(ops.partials_count(n), dtype=F64, device=self._device)
        n2 = torch.zeros(1, dtype=F64, device=self._device)
        track = []
        for _ in range(max_iterations):
            self.iterations += 1
            ops.csr_residual_norm2(A, x, b, r, part, n2)               # Jacobi.py:28-29
            self.residual = math.sqrt(n2.item())
            track.append(self.residual)
            if self.residual <= error:                                 # :32
                self._log("Reached convergence")
                break
            ops.csr_jacobi(A, x, b, omega, y)                          # :35
            x, y = y, x
        self.solution = self._column(x)
        self.residual_vector = self._column(r)
        self.track_res = np.array(track, dtype=float).reshape(-1, 1)
