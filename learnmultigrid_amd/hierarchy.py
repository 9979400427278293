"""Device-resident multigrid hierarchy and the V-cycle executor.

The reference rebuilds everything inside every cycle (Multigrid.py:91-106: transfer
operator lookup, `i.T @ A @ i`, SuperLU factorisation -- its own TODOs at :22-23).
Here that is the SETUP phase, done once per (matrix, transfers):
    P_l        uploaded as CSR (any scipy.sparse / ndarray input),
    R_l = P_l^T as an explicit CSR (restriction becomes a gather SpMV: deterministic,
               no atomics),
    A_{l+1} = (R_l A_l) P_l   by the device SpGEMM (lmg_spgemm_*), evaluated left to
               right like SciPy evaluates `i.T @ A @ i`,
    coarsest   A_L^-1 formed once on the device (dense, fp64) and applied per cycle by
               lmg_dense_gemv plus `coarse_refine` steps of iterative refinement
               (replaces `spsolve(A_coarse, res_coarse)` of Multigrid.py:106).
The SOLVE phase (Multigrid.py:77-124) then only launches bandwidth-bound kernels on
vectors that never leave HBM, and can be captured into a hipGraph.
"""
import math

import numpy as np
import scipy.sparse as sp
import torch

from . import ops
from .coarse import make_coarse_solver
from .ops import DeviceCSR, F64


def _to_csr_host(M):
    """csr_matrix(...) exactly like SemiGeometricMG.__init__ (Multigrid.py:182): dense
    inputs lose their zeros, sparse inputs keep explicit zeros."""
    M = sp.csr_matrix(M, dtype=np.float64)
    if not M.has_canonical_format:
        M = M.copy()
        M.sum_duplicates()
    return M


class Level:
    __slots__ = ("n", "A", "P", "R", "x", "b", "r", "tmp", "plan_RA", "plan_RAP", "RA",
                 "gs_sched", "host_pattern", "dinv", "M", "plan_RM", "plan_RMP", "RM")

    def __init__(self, A):
        self.n = A.shape[0]
        self.A = A
        self.P = self.R = None
        self.plan_RA = self.plan_RAP = self.RA = None
        dev = A.device
        self.x = torch.zeros(self.n, dtype=F64, device=dev)
        self.b = torch.zeros(self.n, dtype=F64, device=dev)
        self.r = torch.zeros(self.n, dtype=F64, device=dev)
        self.tmp = torch.zeros(self.n, dtype=F64, device=dev)
        self.gs_sched = {}
        self.host_pattern = None
        self.dinv = None
        self.M = None                  # mass matrix of the level (optional, see Hierarchy(mass=...))
        self.plan_RM = self.plan_RMP = self.RM = None


class Hierarchy:
    """levels[0] is the fine grid; transfers[l] (n_l x n_{l+1}) prolongates level l+1 -> l."""

    def __init__(self, A, transfers, device, coarse_refine="auto", verbose=False, ops_mod=None,
                 use_packed=True, coarse_solver="auto", spgemm_record="lazy", mass=None):
        """coarse_refine: steps of iterative refinement around every coarsest-level solve; "auto" (default) measures
        the solver once at setup -- ||b - A x|| / ||b|| of one application -- and refines only when that is not at
        rounding level (the explicit block inverses of coarse.py reach 1e-14 on the Galerkin operators of grid
        problems: no refinement, half the launches and bytes of the coarsest solve).
        mass: optional fine-level mass matrix; every level then also gets M_(l+1) = Q_l^T M_l Q_l by the
        same device SpGEMM (`M_coarse = i.T @ M @ i`, Multigrid.py:273-275, and `mass = Q.T @ mass @ Q` of
        NeuralMG_2D.define_hierarchy, :763): levels[l].M, refreshed by rebuild_mass_numeric()."""
        # `ops_mod` exists for the CPU-only host-logic tests (a test shim stands in for the
        # HIP kernels); the product always runs with learnmultigrid_amd.ops.
        self.ops = ops if ops_mod is None else ops_mod
        self.device = torch.device(device)
        # every kernel wrapper launches on the CURRENT device's stream: build on the device that holds the data
        if self.device.type == "cuda":
            with torch.cuda.device(self.device):
                self._build(A, transfers, coarse_refine, verbose, use_packed, coarse_solver, spgemm_record, mass)
        else:
            self._build(A, transfers, coarse_refine, verbose, use_packed, coarse_solver, spgemm_record, mass)

    def _build(self, A, transfers, coarse_refine, verbose, use_packed, coarse_solver, spgemm_record, mass):
        ops_ = self.ops
        self._coarse_refine_arg = coarse_refine
        self.coarse_refine = 1 if coarse_refine == "auto" else int(coarse_refine)
        self.coarse_strategy = coarse_solver
        self.verbose = verbose
        # every launch of this hierarchy goes to one explicit HIP stream (the legacy default
        # stream cannot be captured into a hipGraph)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        A0 = A if isinstance(A, DeviceCSR) else DeviceCSR.from_scipy(A, self.device)
        self.levels = [Level(A0)]
        if mass is not None:
            self.levels[0].M = mass if isinstance(mass, DeviceCSR) else DeviceCSR.from_scipy(_to_csr_host(mass), self.device)
            if self.levels[0].M.shape != A0.shape:
                raise ValueError("mass matrix %s does not match the operator %s" % (self.levels[0].M.shape, A0.shape))
        for P in transfers:
            lev = self.levels[-1]
            Ph = _to_csr_host(P)
            if Ph.shape[0] != lev.n:
                raise ValueError("transfer operator of level %d has %d rows, level has %d unknowns"
                                 % (len(self.levels) - 1, Ph.shape[0], lev.n))
            lev.P = DeviceCSR.from_scipy(Ph, self.device)
            lev.R = lev.P.transpose()
            lev.plan_RA = ops_.SpGEMMPlan(lev.R, lev.A, spgemm_record)
            lev.RA = lev.plan_RA.numeric(lev.R, lev.A)
            lev.plan_RAP = ops_.SpGEMMPlan(lev.RA, lev.P, spgemm_record)
            Ac = lev.plan_RAP.numeric(lev.RA, lev.P)
            self.levels.append(Level(Ac))
            if lev.M is not None:
                lev.plan_RM = ops_.SpGEMMPlan(lev.R, lev.M, spgemm_record)
                lev.RM = lev.plan_RM.numeric(lev.R, lev.M)
                lev.plan_RMP = ops_.SpGEMMPlan(lev.RM, lev.P, spgemm_record)
                self.levels[-1].M = lev.plan_RMP.numeric(lev.RM, lev.P)
        self.use_packed = bool(use_packed)
        self._pack_all()
        self._inverse_diagonals()
        self.partials = torch.empty(ops_.partials_count(self.levels[0].n), dtype=F64, device=self.device)
        self.outer_r = torch.zeros(self.levels[0].n, dtype=F64, device=self.device)
        self.norm2 = torch.zeros(1, dtype=F64, device=self.device)
        self._factor_coarsest()
        self._graphs = {}

    # ------------------------------------------------------------------ setup ----------
    @property
    def sizes(self):
        return [lev.n for lev in self.levels]

    def _pack_all(self):
        """Packed twins (lossless, fewer HBM bytes) of every operator the sweeps touch."""
        if not self.use_packed:
            return
        for lev in self.levels:
            for M in (lev.A, lev.P, lev.R):
                if M is not None and hasattr(M, "pack"):
                    M.pack()

    def _inverse_diagonals(self):
        # for the zero-initial-guess first sweep on coarse levels (allocated at setup so that
        # nothing has to be allocated while a hipGraph is being captured)
        for lev in self.levels[1:-1]:
            lev.dinv = self.ops.csr_inverse_diagonal(lev.A)

    def _factor_coarsest(self):
        """Direct solver of the coarsest operator (setup): dense inverse, or the banded block
        elimination of coarse.py when the operator is narrow-banded (grid problems)."""
        A = self.levels[-1].A
        old = getattr(self, "coarse", None)
        if old is not None and hasattr(old, "factor") and old.n == A.shape[0]:
            try:
                old.factor(A)                          # same pattern: numeric phase only
                self._measure_coarse()
                return
            except ValueError:
                pass
        self.coarse = make_coarse_solver(A, self.ops, self.coarse_strategy)
        self._measure_coarse()

    COARSE_AUTO_TOL = 1e-12

    def _measure_coarse(self):
        """coarse_refine="auto": one application of the fresh factors on a fixed right-hand side; its relative residual
        decides whether the cycles refine (setup only: one SpMV, one solve, two 8-byte reads)."""
        if self._coarse_refine_arg != "auto":
            return
        lev = self.levels[-1]
        n = lev.n
        g = torch.Generator().manual_seed(1234)
        bh = torch.rand(n, dtype=F64, generator=g) - 0.5
        nb = float(bh.norm())
        b = bh.to(self.device)
        x = torch.zeros(n, dtype=F64, device=self.device)
        self.coarse.apply(b, x)
        self.ops.csr_residual_norm2(lev.A, x, b, None, self.partials, self.norm2)       # own kernels only (no library load)
        rel = math.sqrt(max(float(self.norm2.item()), 0.0)) / nb
        self.coarse_residual = rel
        self.coarse_refine = 0 if rel <= self.COARSE_AUTO_TOL else 1

    def rebuild_numeric(self, new_vals):
        """Galerkin rebuild after the VALUES of the fine matrix changed (same pattern):
        numeric SpGEMM passes only, then the coarse factorisation (config #5)."""
        if self.device.type == "cuda" and torch.cuda.current_device() != self.device.index and self.device.index is not None:
            with torch.cuda.device(self.device):
                return self.rebuild_numeric(new_vals)
        A0 = self.levels[0].A
        if new_vals.numel() != A0.nnz:
            raise ValueError("rebuild_numeric needs the same sparsity pattern")
        A0.vals.copy_(new_vals)
        for l in range(len(self.levels) - 1):
            lev = self.levels[l]
            lev.plan_RA.numeric(lev.R, lev.A, out=lev.RA)
            lev.plan_RAP.numeric(lev.RA, lev.P, out=self.levels[l + 1].A)
        if self.use_packed:
            for lev in self.levels:
                lev.A.repack_values()            # same pattern: only the value streams change
        self._inverse_diagonals()
        self._factor_coarsest()
        self._graphs = {}

    def rebuild_mass_numeric(self, new_vals):
        """New values of the fine mass matrix on the same pattern: numeric SpGEMM passes only."""
        M0 = self.levels[0].M
        if M0 is None or new_vals.numel() != M0.nnz:
            raise ValueError("rebuild_mass_numeric needs a hierarchy built with mass= and the same sparsity pattern")
        M0.vals.copy_(new_vals)
        for l in range(len(self.levels) - 1):
            lev = self.levels[l]
            lev.plan_RM.numeric(lev.R, lev.M, out=lev.RM)
            lev.plan_RMP.numeric(lev.RM, lev.P, out=self.levels[l + 1].M)

    def gs_schedule(self, l, kind):
        lev = self.levels[l]
        if kind not in lev.gs_sched:
            if lev.host_pattern is None:
                lev.host_pattern = (lev.A.rowptr.cpu().numpy(), lev.A.colidx.cpu().numpy())
            rp, ci = lev.host_pattern
            pat = sp.csr_matrix((np.ones(ci.size, dtype=np.int8), ci, rp), shape=lev.A.shape)
            sched = self.ops.build_gs_schedule(pat, kind, self.device)
            # the schedule-ordered pattern copy of the one-workgroup executor is built HERE, eagerly:
            # it allocates and reads sizes back to the host, which must never happen while a
            # hipGraph is being captured (captured_cycle only calls gs_schedule before the capture)
            prep = getattr(self.ops, "gs_prepare", None)
            if prep is not None:
                prep(lev.A, sched)
            lev.gs_sched[kind] = sched
        return lev.gs_sched[kind]

    # ------------------------------------------------------------------ solve ----------
    def smooth(self, l, smoother, steps, omega, gs_mode, x_is_zero=False):
        """`steps` smoothing sweeps on level l.  x_is_zero: the iterate is known to be zero
        (coarse levels start from zeros, Multigrid.py:103): the first Jacobi sweep then is
        x = omega * (D^-1 b) -- same bits, a third of the bytes -- and nobody has to clear x."""
        lev = self.levels[l]
        if steps <= 0:
            if x_is_zero:
                self.ops.zero(lev.x)
            return
        if smoother == "GaussSeidel":
            if x_is_zero:
                self.ops.zero(lev.x)
            if self._wavefront_gs(l, gs_mode):
                # grid-stencil level: exact forward sweep as a pipelined wavefront, no schedule needed
                self.ops.stencil_gs(lev.A, lev.x, lev.b, steps)
            else:
                self.ops.csr_gs_schedule(lev.A, lev.x, lev.b, self.gs_schedule(l, gs_mode), steps)
        elif smoother == "Jacobi":
            if x_is_zero:
                if lev.dinv is None:
                    lev.dinv = self.ops.csr_inverse_diagonal(lev.A)
                self.ops.vmul(omega, lev.dinv, lev.b, lev.tmp)
                lev.x, lev.tmp = lev.tmp, lev.x
                steps -= 1
            for _ in range(steps):
                self.ops.csr_jacobi(lev.A, lev.x, lev.b, omega, lev.tmp)
                lev.x, lev.tmp = lev.tmp, lev.x
        else:
            raise ValueError("unknown smoother %r" % (smoother,))

    def _wavefront_gs(self, l, gs_mode):
        avail = getattr(self.ops, "stencil_gs_available", None)
        return gs_mode == "lexicographic" and avail is not None and avail(self.levels[l].A)

    def _fusable(self, l, smoother, steps):
        avail = getattr(self.ops, "stencil_smooth_available", None)
        return (smoother == "Jacobi" and steps >= 1 and avail is not None and avail(self.levels[l].A))

    def smooth_fused(self, l, steps, omega, x_is_zero=False, want_residual=False, correction=None, restrict_to=None):
        """`steps` Jacobi sweeps on level l (and r = b - A x afterwards) as fused passes of at most
        FUSED_MAX_SWEEPS sweeps each (lmg_stencil_smooth): same bits as smooth() + the residual launch,
        a third of the passes over the level's vectors.  correction = (P, e): the first pass starts from
        x + P e (Multigrid.py:115 folded in; the caller has checked stencil_smooth_prolong_available)."""
        lev = self.levels[l]
        left = steps
        mx = self.ops.FUSED_MAX_SWEEPS
        while left > 0:
            k = min(left, mx)
            left -= k
            if correction is not None:
                self.ops.stencil_smooth(lev.A, lev.x, lev.b, omega, k, lev.tmp, None, prolong=correction)
                correction = None
            elif restrict_to is not None and left == 0:
                # b_coarse = R (b - A x) formed in the pass, the residual itself is not written
                self.ops.stencil_smooth(lev.A, None if x_is_zero else lev.x, lev.b, omega, k, lev.tmp, None,
                                        restrict=restrict_to)
            else:
                self.ops.stencil_smooth(lev.A, None if x_is_zero else lev.x, lev.b, omega, k, lev.tmp,
                                        lev.r if (want_residual and left == 0) else None)
            lev.x, lev.tmp = lev.tmp, lev.x
            x_is_zero = False

    def coarse_solve(self):
        lev = self.levels[-1]
        self.coarse.apply(lev.b, lev.x)
        for _ in range(self.coarse_refine):
            self.ops.csr_residual_norm2(lev.A, lev.x, lev.b, lev.r, None, None)
            if getattr(self.coarse, "supports_accumulate", False):
                self.coarse.apply(lev.r, lev.x, accumulate=True)      # x += A^-1 r in the solver's last launch
            else:
                self.coarse.apply(lev.r, lev.tmp)
                self.ops.axpby(1.0, lev.tmp, 1.0, lev.x)

    def cycle(self, smoother, steps, omega=1.0, gs_mode="lexicographic", l=0, depth=None,
              after_presmooth=None, x_is_zero=False):
        """One V(steps, steps) cycle on level l: levels[l].x is the iterate, levels[l].b the
        right-hand side (Multigrid.py:77-124).  depth = number of grids used."""
        last = (len(self.levels) if depth is None else depth) - 1
        lev, nxt = self.levels[l], self.levels[l + 1]
        fused = self._fusable(l, smoother, steps)
        ravail = getattr(self.ops, "stencil_smooth_restrict_available", None)
        restricted = False
        if fused and ravail is not None and ravail(lev.A, lev.R):
            self.smooth_fused(l, steps, omega, x_is_zero, restrict_to=(lev.R, nxt.b))   # :88 + :90 + :93 in one pass
            restricted = True
            if after_presmooth is not None:
                after_presmooth(lev.x)
        elif fused:
            self.smooth_fused(l, steps, omega, x_is_zero, want_residual=True)  # :88 + :90 in one pass
            if after_presmooth is not None:
                after_presmooth(lev.x)
        else:
            self.smooth(l, smoother, steps, omega, gs_mode, x_is_zero)            # :88
            if after_presmooth is not None:
                after_presmooth(lev.x)
            self.ops.csr_residual_norm2(lev.A, lev.x, lev.b, lev.r, None, None)        # :90
        if not restricted:
            self.ops.csr_spmv(lev.R, lev.r, nxt.b, 1.0, 0.0)                       # :93
        if l + 1 == last:
            self.coarse_solve()                                               # :106
        else:
            self.cycle(smoother, steps, omega, gs_mode, l + 1, depth, x_is_zero=True)   # zeros, :103
        pavail = getattr(self.ops, "stencil_smooth_prolong_available", None)
        if fused and pavail is not None and pavail(lev.A, lev.P):
            self.smooth_fused(l, steps, omega, correction=(lev.P, nxt.x))     # :115 + :121 in one pass
            return
        self.ops.csr_spmv(lev.P, nxt.x, lev.x, 1.0, 1.0)                           # :115
        if fused:
            self.smooth_fused(l, steps, omega)                                # :121
        else:
            self.smooth(l, smoother, steps, omega, gs_mode)                       # :121

    def residual_norm(self, want_vector=True):
        """||b - A x||_2 on the fine level (Multigrid.py:62-63); one 8-byte D2H copy."""
        lev = self.levels[0]
        self.ops.csr_residual_norm2(lev.A, lev.x, lev.b, self.outer_r if want_vector else None,
                               self.partials, self.norm2)
        return math.sqrt(self.norm2.item())

    def check_smoothers(self):
        """Raise if a wavefront Gauss-Seidel band of any level ever gave up waiting for its predecessor
        (gs_wave.hip then leaves a wrong iterate behind and sets a flag).  One 4-byte D2H read per level that
        has run the wavefront kernel: call it where the host synchronises anyway (after the residual norm of an
        outer iteration, after a graph replay)."""
        chk = getattr(self.ops, "stencil_gs_check", None)
        if chk is None:
            return
        for lev in self.levels[:-1]:
            chk(lev.A)

    def prepare_smoother(self, smoother, gs_mode="lexicographic", l_from=0):
        """Everything a Gauss-Seidel cycle would otherwise do lazily on its first sweep -- the wavefront kernel's eligibility
        test (a device -> host read) and work buffer, the level schedules -- from level l_from down: nothing of it may happen
        while a hipGraph is being captured."""
        if smoother != "GaussSeidel":
            return
        for l in range(l_from, len(self.levels) - 1):
            if self._wavefront_gs(l, gs_mode):
                self.ops.stencil_gs(self.levels[l].A, self.levels[l].tmp, self.levels[l].b, 0)
            else:
                self.gs_schedule(l, gs_mode)

    def captured_cycle(self, smoother, steps, omega, gs_mode):
        """The same launch sequence as cycle(), captured once into a hipGraph and replayed."""
        key = (smoother, steps, omega, gs_mode)
        g = self._graphs.get(key)
        if g is None:
            self.prepare_smoother(smoother, gs_mode)
            before = [(lev.x, lev.tmp) for lev in self.levels]
            g = self.ops.CapturedGraph()
            with g:
                self.cycle(smoother, steps, omega, gs_mode)
            after = [(lev.x, lev.tmp) for lev in self.levels]
            if any(a[0] is not b[0] for a, b in zip(before, after)):
                raise RuntimeError("ping-pong buffers did not return to their slots")
            self._graphs[key] = g
        return g

    def memory_bytes(self):
        tot = 0
        for lev in self.levels:
            tot += lev.A.bytes() + 4 * 8 * lev.n
            for M in (lev.P, lev.R, lev.RA):
                if M is not None:
                    tot += M.bytes()
        return tot + self.coarse.bytes_per_apply()

    def cycle_bytes(self, steps):
        """Algorithmic HBM bytes of one V(steps,steps) cycle: per level
        (2*steps+1) sweeps + restriction + prolongation; coarsest dense apply separately."""
        tot = 0
        for lev in self.levels[:-1]:
            n, nnz = lev.n, lev.A.nnz
            nc = lev.P.shape[1]
            tot += (2 * steps + 1) * (12 * nnz + 4 * (n + 1) + 24 * n)
            tot += 12 * lev.R.nnz + 4 * (nc + 1) + 8 * n + 8 * nc
            tot += 12 * lev.P.nnz + 4 * (n + 1) + 8 * nc + 16 * n
        coarse = (1 + self.coarse_refine) * self.coarse.bytes_per_apply()
        return tot, coarse
