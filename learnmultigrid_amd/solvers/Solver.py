"""Base solver API -- mirrors learn_multigrid/solvers/Solver.py:8-84 of the reference
(same constructor, getters/setters, attributes and shapes) with the state living on an
MI355X.  Host arrays handed back by getters are fresh NumPy arrays of shape (n, 1)."""
import functools

import numpy as np
import scipy.sparse as sp
import torch

from .. import _lib, ops
from ..ops import DeviceCSR, F64


def on_device(method):
    """Run a solver method with the solver's GPU as the current device: the kernel wrappers launch on the
    CURRENT device's current stream, which must be the device that holds the tensors (Solver(..., device="cuda:1")
    while cuda:0 is current)."""
    @functools.wraps(method)
    def wrapper(self, *args, **kw):
        dev = getattr(self, "_device", None)
        if dev is not None and dev.type == "cuda":
            with torch.cuda.device(dev):
                return method(self, *args, **kw)
        return method(self, *args, **kw)
    return wrapper


def default_device():
    _lib.lib()                                   # raises loudly if the HIP library is missing
    if not torch.cuda.is_available():
        raise _lib.LmgError("no MI355X visible (torch.cuda.is_available() is False); "
                            "learnmultigrid_amd has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


class Solver:
    """Solver(matrix, rhs): matrix is anything scipy.sparse.csc_matrix accepts
    (Solver.py:18); rhs is the (n, 1) float64 column every reference script passes."""

    def __init__(self, matrix, rhs, *, device=None, verbose=False):
        rhs = np.asarray(rhs)
        self.dim = rhs.size                                       # Solver.py:15
        self.residual_vector = np.empty(shape=rhs.shape)
        self.residual = 0.0
        self.matrix = sp.csc_matrix(matrix)                       # Solver.py:18
        self.rhs = rhs
        self.solution = np.empty(shape=rhs.shape)
        self.track_res = np.ndarray(shape=(0, 1), dtype=float)
        self.verbose = verbose
        self._device = torch.device(device) if device is not None else default_device()
        self._dA = None
        if self.matrix.shape[0] != self.matrix.shape[1] or self.matrix.shape[0] != self.dim:
            raise ValueError("matrix %s does not match rhs of size %d" % (self.matrix.shape, self.dim))

    # -- reference accessors (Solver.py:23-48) --------------------------------------------
    def set_matrix(self, matrix):
        self.matrix = matrix
        self._invalidate()

    def get_matrix(self):
        return self.matrix

    def get_residual_vector(self):
        return self.residual_vector

    def get_residual(self):
        return self.residual

    def set_rhs(self, rhs):
        self.rhs = rhs

    def get_rhs(self):
        return self.rhs

    def get_solution(self):
        return self.solution

    def get_dimension(self):
        return self.dim

    def get_track_res(self):
        return self.track_res

    # -- device plumbing ---------------------------------------------------------------------
    def _invalidate(self):
        self._dA = None

    def _device_matrix(self):
        if self._dA is None:
            self._dA = DeviceCSR.from_scipy(self.matrix, self._device)
        return self._dA

    def _to_device(self, v):
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
        if a.size != self.dim:
            raise ValueError("vector of size %d, expected %d" % (a.size, self.dim))
        return torch.from_numpy(a).to(self._device)

    def _column(self, t):
        return t.detach().cpu().numpy().reshape(self.dim, 1).copy()

    def _log(self, *a):
        if self.verbose:
            print(*a)


class DirectSolver(Solver):
    """Solver.py:51-59.  The reference calls SuperLU (`spsolve`); here the operator is
    inverted once on the device (dense fp64) and applied with two steps of iterative
    refinement by the HIP kernels -- meant for the small systems the scripts use it on."""

    def solve(self):
        from ..coarse import MAX_DENSE as MAX_DENSE_COARSE, csr_to_dense, dense_inverse
        if self.dim > MAX_DENSE_COARSE:
            raise ValueError("DirectSolver on the device is limited to %d unknowns" % MAX_DENSE_COARSE)
        A = self._device_matrix()
        n = self.dim
        inv = dense_inverse(csr_to_dense(A))
        b = self._to_device(self.rhs)
        x = torch.empty_like(b)
        r = torch.empty_like(b)
        d = torch.empty_like(b)
        ops.dense_gemv(inv, b, x)
        for _ in range(2):
            ops.csr_residual_norm2(A, x, b, r, None, None)
            ops.dense_gemv(inv, r, d)
            ops.axpby(1.0, d, 1.0, x)
        part = torch.empty(ops.partials_count(n), dtype=F64, device=self._device)
        n2 = torch.zeros(1, dtype=F64, device=self._device)
        ops.csr_residual_norm2(A, x, b, r, part, n2)
        self.solution = self._column(x)
        self.residual_vector = self._column(r)
        self.residual = float(np.sqrt(n2.item()))


class IterativeSolver(Solver):
    """Solver.py:62-84."""

    def __init__(self, matrix, rhs, **kw):
        super().__init__(matrix, rhs, **kw)
        self.iterations = 0
        self.label = "Iterative Solver"

    def plot(self, scale="linear"):
        import matplotlib.pyplot as plt
        plt.plot(self.track_res, label=self.label)
        plt.yscale(scale)
        plt.legend()
        plt.title("Residual decreasing")
        plt.ylabel("residual")
        plt.xlabel("iterations")

    def get_iterations(self):
        return self.iterations
