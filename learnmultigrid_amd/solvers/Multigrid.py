"""Multigrid V-cycle solvers with the reference's call signatures
(learn_multigrid/solvers/Multigrid.py:26-197), executed on an MI355X.

    GeometricMG(A, rhs).solve(levels=3, smoother="GaussSeidel", smooth_steps=3, ...)
    SemiGeometricMG(A, rhs, Q).solve(levels=2, smoother="GaussSeidel", smooth_steps=3,
                                     error=1e-10, max_iterations=40)
    mg.get_track_res(), mg.get_iterations(), mg.get_solution(), mg.get_residual()

Reference behaviour that is reproduced on purpose:
  * the first recorded residual is ||1|| = sqrt(n) (Multigrid.py:64-66);
  * convergence is tested BEFORE each cycle; `iterations` counts started iterations;
  * `levels` counts grids (levels-1 coarsenings, direct solve on the last, :78,:102);
  * as shipped, the `smoother` name is ignored and forward lexicographic Gauss-Seidel
    is always applied (:79-88,:121): that is `smoother_semantics="as_shipped"`, the
    default.  `smoother_semantics="as_named"` honours the name ("Jacobi" with `omega`,
    "GaussSeidel" with `gs_mode`), which is what the commented-out lines :85-86 intend;
  * SemiGeometricMG uses the supplied Q between levels 0 and 1 only and the 1-D
    geometric interpolator below (:188-197) unless a `hierarchy` list is given.
Deviations (all louder, none silent): unknown cycle / smoother / levels < 2 raise
ValueError instead of sys.exit(0) / TypeError / unbounded recursion; printing is
opt-in (verbose=True); the caller's initial_guess is only overwritten when
mutate_initial_guess=True (the reference smooths it in place, :43,:88).
What the reference recomputes in every cycle (transfer lookup, R A P, SuperLU
factorisation) is done once per solve() in the Hierarchy setup.
"""
import numpy as np
import scipy.sparse as sp
import torch

from .. import ops, problems
from ..hierarchy import Hierarchy
from .Solver import IterativeSolver, on_device

_SMOOTHERS = ("GaussSeidel", "Jacobi", "CG")          # Multigrid.py:149-155


class Multigrid(IterativeSolver):

    def __init__(self, matrix, rhs, **kw):
        super().__init__(matrix, rhs, **kw)
        self._log("Selected Multigrid")
        self.label = "Multigrid"
        self.hierarchy = None              # optional list of transfer operators (all levels)
        self._hier = None
        self._hier_key = None

    # -- transfer-operator dispatch (Multigrid.py:91 -> :126 / :171 / :188) --------------------
    def interpolator(self, dimension, _first_call=False):
        """1-D geometric interpolator (n x floor((n-1)/2)+1), CSR with the values of
        Multigrid.py:126-147 (the reference returns the same matrix dense)."""
        return problems.geometric_interpolator_1d(dimension)

    def _transfers(self, levels, first_call):
        n = self.dim
        out = []
        for l in range(levels - 1):
            if self.hierarchy is not None:
                if l >= len(self.hierarchy):
                    raise ValueError("hierarchy has %d operators, levels=%d needs %d"
                                     % (len(self.hierarchy), levels, levels - 1))
                P = sp.csr_matrix(self.hierarchy[l])
            else:
                P = self.interpolator(n, first_call if l == 0 else False)   # :103-104
            if P.shape[0] != n:
                raise ValueError("transfer operator %d has %d rows, expected %d" % (l, P.shape[0], n))
            out.append(P)
            n = P.shape[1]
        return out

    def _setup(self, levels, first_call, coarse_refine):
        key = (levels, bool(first_call), id(self.matrix), str(coarse_refine),
               None if self.hierarchy is None else tuple(id(h) for h in self.hierarchy))
        if self._hier is None or self._hier_key != key:
            self._hier = Hierarchy(self.matrix, self._transfers(levels, first_call), self._device,
                                   coarse_refine=coarse_refine, verbose=self.verbose)
            self._hier_key = key
        return self._hier

    def _invalidate(self):
        super()._invalidate()
        self._hier = None

    @staticmethod
    def _effective_smoother(smoother, semantics):
        if smoother not in _SMOOTHERS:
            raise ValueError("unknown smoother %r (reference: 'Jacobi', 'GaussSeidel', 'CG')" % (smoother,))
        if semantics == "as_shipped":
            return "GaussSeidel"
        if semantics != "as_named":
            raise ValueError("smoother_semantics must be 'as_shipped' or 'as_named'")
        if smoother == "CG":
            raise ValueError("CG is not a V-cycle smoother in this build (unused by the reference too)")
        return smoother

    # -- Multigrid.solve (Multigrid.py:36-75) ---------------------------------------------------
    @on_device
    def solve(self, levels=2, smoother="Jacobi", smooth_steps=1, max_iterations=100, error=1e-08,
              initial_guess=None, cycle="V", first_call=False, *, omega=1.0,
              smoother_semantics="as_shipped", gs_mode="lexicographic", coarse_refine="auto",
              use_graph=False, mutate_initial_guess=False):
        if cycle != "V":
            raise ValueError("Cycle type unknown: %r" % (cycle,))            # :55-57
        if levels < 2:
            raise ValueError("levels must be >= 2 (levels counts grids)")
        eff = self._effective_smoother(smoother, smoother_semantics)
        H = self._setup(levels, first_call, coarse_refine)
        H.stream.wait_stream(torch.cuda.current_stream(self._device))
        with torch.cuda.stream(H.stream):
            self._solve_on_stream(H, eff, smooth_steps, max_iterations, error, initial_guess, omega,
                                  gs_mode, use_graph, mutate_initial_guess)
        torch.cuda.current_stream(self._device).wait_stream(H.stream)

    def _solve_on_stream(self, H, eff, smooth_steps, max_iterations, error, initial_guess, omega,
                         gs_mode, use_graph, mutate_initial_guess):
        fine = H.levels[0]
        if initial_guess is None:
            self._log("You should put an initial guess. Used zero vector")
            ops.zero(fine.x)
        else:
            fine.x.copy_(self._to_device(initial_guess))
        fine.b.copy_(self._to_device(self.rhs))
        hook = None
        if mutate_initial_guess and initial_guess is not None:
            state = {"done": False}

            def hook(x_dev):
                if not state["done"]:
                    np.asarray(initial_guess).reshape(-1)[:] = x_dev.cpu().numpy()
                    state["done"] = True
        graph = H.captured_cycle(eff, smooth_steps, omega, gs_mode) if use_graph else None
        track = []
        for _ in range(max_iterations):                                      # :59
            self.iterations += 1
            self.residual = H.residual_norm()                                # :62-63
            if eff == "GaussSeidel":
                H.check_smoothers()          # the norm read has synchronised: a timed-out wavefront band raises here
            if self.iterations <= 1:                                         # :64-66
                self.residual = float(np.linalg.norm(np.ones(shape=(self.dim, 1))))
            track.append(self.residual)
            self._log("It: ", self.iterations, self.residual)
            if self.residual <= error:                                       # :69-71
                break
            if graph is not None and hook is None:
                graph.launch()
            else:
                H.cycle(eff, smooth_steps, omega, gs_mode, after_presmooth=hook)   # :73
                hook = None
        self.solution = self._column(fine.x)
        self.residual_vector = (np.ones(shape=(self.dim, 1)) if self.iterations <= 1
                                else self._column(H.outer_r))
        self.track_res = np.array(track, dtype=float).reshape(-1, 1)         # :75
        self.level_dims = H.sizes

    # -- Multigrid.v_cycle (Multigrid.py:77) ------------------------------------------------------
    @on_device
    def v_cycle(self, A, u0, rhs, smoother, smooth_steps, error, levels, first_call=False, *,
                omega=1.0, smoother_semantics="as_shipped", gs_mode="lexicographic",
                coarse_refine="auto"):
        """One V-cycle on (A, rhs) from u0; returns a fresh (n,1) array.  u0 receives the
        pre-smoothed iterate like in the reference (:88-89)."""
        if levels < 2:
            raise ValueError("levels must be >= 2")
        eff = self._effective_smoother(smoother, smoother_semantics)
        if A is self.matrix:
            H = self._setup(levels, first_call, coarse_refine)
        else:
            saved = self.matrix, self.dim, self._hier, self._hier_key
            self.matrix, self.dim = sp.csc_matrix(A), A.shape[0]
            self._hier = None
            try:
                H = self._setup(levels, first_call, coarse_refine)
            finally:
                self.matrix, self.dim, self._hier, self._hier_key = saved
        fine = H.levels[0]
        n = fine.n
        u0a = np.asarray(u0)
        fine.x.copy_(torch.from_numpy(np.ascontiguousarray(u0a, dtype=np.float64).reshape(-1)).to(self._device))
        fine.b.copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(rhs), dtype=np.float64).reshape(-1)).to(self._device))

        def hook(x_dev):
            if u0a.dtype == np.float64 and u0a.flags.writeable:
                u0a.reshape(-1)[:] = x_dev.cpu().numpy()
        H.stream.wait_stream(torch.cuda.current_stream(self._device))
        with torch.cuda.stream(H.stream):
            H.cycle(eff, smooth_steps, omega, gs_mode, after_presmooth=hook)
            out = fine.x.cpu().numpy().reshape(n, 1).copy()
            if eff == "GaussSeidel":
                H.check_smoothers()
        torch.cuda.current_stream(self._device).wait_stream(H.stream)
        return out

    def smoother_to_method(self, smoother):
        from .Jacobi import Jacobi
        from .GaussSeidel import GaussSeidel
        table = {"GaussSeidel": GaussSeidel, "Jacobi": Jacobi}
        if smoother not in table:
            raise ValueError("Invalid smoother %r" % (smoother,))
        return table[smoother]

    def cycle_to_method(self, cycle):
        if cycle != "V":
            raise ValueError("Invalid cycle %r" % (cycle,))
        return self.v_cycle


class GeometricMG(Multigrid):
    """Multigrid.py:164-173: 1-D geometric interpolator on every level."""

    def __init__(self, matrix, rhs, **kw):
        super().__init__(matrix, rhs, **kw)
        self._log("Selected Geometric Multigrid")
        self.label = "GeometricMG"


class SemiGeometricMG(Multigrid):
    """Multigrid.py:176-197: l2_proj (any matrix with one row per fine unknown -- an L2
    projection or a learned Q) between levels 0 and 1, geometric below.  `hierarchy`
    (keyword-only) supplies the operators of ALL levels instead: the consumer that
    NeuralMG_2D.define_hierarchy's l_hierarchy (Multigrid.py:741-765) never had."""

    def __init__(self, matrix, rhs, l2_proj, *, hierarchy=None, **kw):
        super().__init__(matrix, rhs, **kw)
        self._log("Selected Semi - Geometric Multigrid")
        self.label = "SemiGeometricMG"
        self.l2_proj = sp.csr_matrix(l2_proj)                               # :182
        if hierarchy is not None:
            self.hierarchy = [sp.csr_matrix(h) for h in hierarchy]

    def solve(self, levels=2, smoother="Jacobi", smooth_steps=1, max_iterations=100, error=1e-08,
              initial_guess=None, cycle="V", first_call=True, **kw):       # :184-186
        super().solve(levels, smoother, smooth_steps, max_iterations, error, initial_guess, cycle,
                      first_call, **kw)

    def interpolator(self, dimension, first_call=False):                    # :188-197
        if first_call:
            return self.l2_proj
        return super().interpolator(dimension, first_call)


class HierarchyMG(SemiGeometricMG):
    """Convenience spelling: HierarchyMG(A, rhs, [Q0, Q1, ...]) == SemiGeometricMG with
    hierarchy=[...] (one learned / projected transfer operator per coarsening)."""

    def __init__(self, matrix, rhs, hierarchy, **kw):
        hierarchy = list(hierarchy)
        super().__init__(matrix, rhs, hierarchy[0], hierarchy=hierarchy, **kw)
        self.label = "HierarchyMG"


class NeuralMG(Multigrid):
    """Multigrid.py:200-370: `NeuralMG(matrix, rhs, model, M, std, mean)`.  The transfer operator of
    every level is built from that level's mass matrix by the caller's model
    (`model.predict`, any object; see learned_q.py for the scatter) and the mass matrix is
    coarsened with it, M_c = Q^T M Q (:273-275).  The reference asks the model again in every
    cycle; a deterministic model returns the same Q each time, so the operators are built once
    per solve() here (setup / solve split) and the cycles run on the device."""

    def __init__(self, matrix, rhs, model, M, std, mean, **kw):
        super().__init__(matrix, rhs, **kw)
        self._log("Selected NN Multigrid")
        self.label = "NeuralMG"
        self.model, self.M, self.std, self.mean = model, M, std, mean

    def transfer_op(self, M):
        from ..learned_q import learned_transfer
        return learned_transfer(self.model, M, self.mean, self.std)                 # :306-311

    def _transfers(self, levels, first_call):
        out = []
        M = sp.csr_matrix(self.M)
        for _ in range(levels - 1):
            if M.shape[0] % 2 == 0:
                raise ValueError("mass matrix of even size %d: the learned 1-D transfer needs odd sizes "
                                 "(test/test_B_patch.py:52-53)" % M.shape[0])
            Q = sp.csr_matrix(self.transfer_op(np.asarray(M.toarray())))
            out.append(Q)
            # M_coarse = Q^T M Q (:273-275) by the device SpGEMM, evaluated (Q^T M) Q like SciPy does: the
            # same bits as the host product without forming dense n x n matrices for it
            dQ = ops.DeviceCSR.from_scipy(Q, self._device)
            dM = ops.DeviceCSR.from_scipy(M, self._device)
            M = ops.spgemm(ops.spgemm(dQ.transpose(), dM), dQ).to_scipy()
        return out
