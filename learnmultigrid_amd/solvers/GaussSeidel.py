"""Stand-alone forward Gauss-Seidel -- mirrors learn_multigrid/solvers/GaussSeidel.py:16-39.
The reference forms (D+L)^-1 explicitly (O(n^2) fill) and does x += (D+L)^-1 (b - A x);
that is one lexicographic forward sweep, executed here exactly (level-scheduled HIP
kernel, same row arithmetic as pyamg's sweep) or, with gs_mode="multicolor", in colour
order (faster, different ordering)."""
import math

import numpy as np
import torch

from .. import ops
from ..ops import F64
from .Solver import IterativeSolver, on_device


class GaussSeidel(IterativeSolver):

    def __init__(self, matrix, rhs, **kw):
        super().__init__(matrix, rhs, **kw)
        self._log("Selected Gauss-Seidel")
        self.label = "Gauss-Seidel"

    @on_device
    def solve(self, max_iterations=1000, error=1e-12, initial_guess=None, *, gs_mode="lexicographic"):
        A = self._device_matrix()
        n = self.dim
        import scipy.sparse as sp
        if gs_mode == "lexicographic" and A.shape[0] >= 4096:
            A.pack()                       # large grid operators: the wavefront kernel needs the stencil twin
        wave = gs_mode == "lexicographic" and ops.stencil_gs_available(A)
        sched = None if wave else ops.build_gs_schedule(sp.csr_matrix(self.matrix), gs_mode, self._device)
        b = self._to_device(self.rhs)
        x = torch.zeros(n, dtype=F64, device=self._device) if initial_guess is None \
            else self._to_device(initial_guess)
        r = torch.empty_like(x)
        part = torch.empty(ops.partials_count(n), dtype=F64, device=self._device)
        n2 = torch.zeros(1, dtype=F64, device=self._device)
        track = []
        for _ in range(max_iterations):
            self.iterations += 1
            ops.csr_residual_norm2(A, x, b, r, part, n2)               # GaussSeidel.py:29-30
            self.residual = math.sqrt(n2.item())
            if wave:
                ops.stencil_gs_check(A)        # synchronised by the norm read: a timed-out band raises here
            track.append(self.residual)
            if self.residual <= error:
                self._log("Reached convergence Gauss")
                break
            if wave:
                ops.stencil_gs(A, x, b, 1)                             # :37, pipelined wavefront (gs_wave.hip)
            else:
                ops.csr_gs_schedule(A, x, b, sched, 1)                 # :37
        if wave:
            ops.stencil_gs_check(A)
        self.solution = self._column(x)
        self.residual_vector = self._column(r)
        self.track_res = np.array(track, dtype=float).reshape(-1, 1)
