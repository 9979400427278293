"""Host logic of the coarsest-level solvers (coarse.py) on CPU tensors with the test-only
ops shim: the banded block elimination must reproduce SuperLU (`spsolve`, the reference's
Multigrid.py:106) to rounding, and the planner must only pick it for narrow-banded operators."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import torch

import cpu_ops_shim as shim
from learnmultigrid_amd import coarse, problems as P
from learnmultigrid_amd.ops import DeviceCSR


def galerkin_operator(m):
    A, _ = P.poisson_2d_structured(2 * m)
    Pm = P.tensor_interpolator_2d(2 * m + 1)
    Ac = sp.csr_matrix(Pm.T @ A @ Pm)
    Ac.sort_indices()
    return Ac


def test_plan_prefers_banding_only_when_it_pays():
    assert coarse.BandedBlockSolver.plan(129 * 129, 130) is not None
    k, s = coarse.BandedBlockSolver.plan(129 * 129, 130)
    assert s % 2 == 0 and k * s + (k - 1) * 130 <= 129 * 129
    assert 2 * k * s * s + (129 * 129 - k * s) ** 2 < 0.15 * (129 * 129) ** 2      # > 6x fewer bytes
    assert coarse.BandedBlockSolver.plan(1000, 400) is None                          # wide band: dense


def test_banded_block_solver_matches_superlu():
    Ac = galerkin_operator(64)                      # 65^2 = 4225 unknowns, 9-point, w = 66
    n = Ac.shape[0]
    assert coarse.half_bandwidth(Ac) == 66
    dA = DeviceCSR.from_scipy(Ac, "cpu")
    assert coarse.make_coarse_solver(dA, shim, "auto").kind == "grid-block"       # grid operators: blocks cut both ways
    solver = coarse.make_coarse_solver(dA, shim, "banded")
    assert solver.kind == "banded-block" and solver.k >= 4
    rng = np.random.default_rng(0)
    b = rng.standard_normal(n)
    x = torch.zeros(n, dtype=torch.float64)
    solver.apply(torch.from_numpy(b.copy()), x)
    want = spla.spsolve(sp.csc_matrix(Ac), b)
    err = np.linalg.norm(x.numpy() - want) / np.linalg.norm(want)
    assert err < 1e-11, err
    # one step of refinement against the true operator brings it to rounding level
    r = b - Ac @ x.numpy()
    d = torch.zeros(n, dtype=torch.float64)
    solver.apply(torch.from_numpy(r), d)
    x2 = x.numpy() + d.numpy()
    assert np.linalg.norm(x2 - want) / np.linalg.norm(want) < 1e-13
    dense = coarse.make_coarse_solver(dA, shim, "dense")
    assert dense.kind == "dense" and dense.bytes_per_apply() > 5 * solver.bytes_per_apply()
    y = torch.zeros(n, dtype=torch.float64)
    dense.apply(torch.from_numpy(b.copy()), y)
    assert np.linalg.norm(y.numpy() - want) / np.linalg.norm(want) < 1e-11


def test_refactor_with_new_values_on_the_same_pattern():
    Ac = galerkin_operator(64)
    n = Ac.shape[0]
    dA = DeviceCSR.from_scipy(Ac, "cpu")
    for strategy in ("banded", "dense"):
        solver = coarse.make_coarse_solver(dA, shim, strategy)
        rng = np.random.default_rng(5)
        A2 = Ac.copy()
        A2.data = A2.data * (1.0 + 0.2 * rng.random(A2.nnz))       # same pattern, other coefficients
        solver.factor(DeviceCSR.from_scipy(A2, "cpu"))
        b = rng.standard_normal(n)
        x = torch.zeros(n, dtype=torch.float64)
        solver.apply(torch.from_numpy(b.copy()), x)
        want = spla.spsolve(sp.csc_matrix(A2), b)
        assert np.linalg.norm(x.numpy() - want) / np.linalg.norm(want) < 1e-11, strategy
        fresh = coarse.make_coarse_solver(DeviceCSR.from_scipy(A2, "cpu"), shim, strategy)
        y = torch.zeros(n, dtype=torch.float64)
        fresh.apply(torch.from_numpy(b.copy()), y)
        assert torch.equal(x, y)                                   # refactoring == building anew
    with_other_pattern = DeviceCSR.from_scipy(galerkin_operator(48), "cpu")
    try:
        coarse.make_coarse_solver(dA, shim, "banded").factor(with_other_pattern)
        raise AssertionError("pattern change must be refused")
    except ValueError:
        pass


def test_unstructured_numbering_falls_back_to_dense():
    Ac = galerkin_operator(24)                       # 25^2 = 625 < 2048 -> dense by size
    assert coarse.make_coarse_solver(DeviceCSR.from_scipy(Ac, "cpu"), shim).kind == "dense"
    A, _ = P.poisson_2d_structured(48)               # 2401 unknowns, scrambled numbering: wide band
    rng = np.random.default_rng(1)
    p = rng.permutation(A.shape[0])
    Ap = sp.csr_matrix(A[p][:, p])
    Ap.sort_indices()
    assert coarse.make_coarse_solver(DeviceCSR.from_scipy(Ap, "cpu"), shim).kind == "dense"


def _l2_galerkin(side):
    """25-point Galerkin operator of an L2-type (learned-like) transfer on a jittered 7-point fine operator:
    what a 2-level learned-Q run hands to the coarse solver (half-bandwidth 2 * side + 2)."""
    A, _ = P.jittered_poisson_2d(2 * (side - 1), seed=42)
    l2 = P.pseudo_l2_interpolator_1d(2 * side - 1)
    Q = P.learned_like(sp.kron(l2, l2).tocsr(), 43)
    Ac = sp.csr_matrix(Q.T @ A @ Q)
    Ac.sort_indices()
    return Ac


def test_block_cyclic_reduction_matches_superlu():
    for Ac in (galerkin_operator(48), _l2_galerkin(41)):            # 9-point 49^2; 25-point 41^2 (odd block counts too)
        n = Ac.shape[0]
        dA = DeviceCSR.from_scipy(sp.csr_matrix(Ac), "cpu")
        solver = coarse.make_coarse_solver(dA, shim, "bcr")
        assert solver.kind == "block-cyclic-reduction" and solver.perm is None
        assert solver.b >= coarse.half_bandwidth(Ac) and len(solver.levels) == int(np.ceil(np.log2(solver.m)))
        assert solver.bytes_per_apply() <= coarse.BlockCyclicReduction.estimate_bytes(n, solver.b)
        rng = np.random.default_rng(1)
        b = rng.standard_normal(n)
        x = torch.zeros(n, dtype=torch.float64)
        solver.apply(torch.from_numpy(b.copy()), x)
        want = spla.spsolve(sp.csc_matrix(Ac), b)
        assert np.linalg.norm(x.numpy() - want) / np.linalg.norm(want) < 1e-10
        d = torch.zeros(n, dtype=torch.float64)
        solver.apply(torch.from_numpy(b - Ac @ x.numpy()), d)
        assert np.linalg.norm(x.numpy() + d.numpy() - want) / np.linalg.norm(want) < 1e-13
        # new values on the same pattern: numeric phase only
        A2 = Ac.copy()
        A2.data = A2.data * (1.0 + 0.1 * rng.random(A2.nnz))
        A2 = sp.csr_matrix(A2 + sp.identity(n) * abs(A2).sum(1).max())           # keep it nonsingular
        A2 = sp.csr_matrix((A2.data, A2.indices, A2.indptr), shape=A2.shape)
        if A2.nnz == Ac.nnz:
            d2 = DeviceCSR.from_scipy(A2, "cpu")
            solver.factor(d2)
            solver.apply(torch.from_numpy(b.copy()), x)
            want2 = spla.spsolve(sp.csc_matrix(A2), b)
            assert np.linalg.norm(x.numpy() - want2) / np.linalg.norm(want2) < 1e-10


def test_block_cyclic_reduction_reorders_scrambled_operators():
    Ac = galerkin_operator(40)
    n = Ac.shape[0]
    rng = np.random.default_rng(3)
    p = rng.permutation(n)
    As = sp.csr_matrix(Ac[p][:, p])                                    # same operator, numbering destroyed
    As.sort_indices()
    assert coarse.half_bandwidth(As) > n // 2
    dA = DeviceCSR.from_scipy(As, "cpu")
    solver = coarse.make_coarse_solver(dA, shim, "bcr")
    assert solver.kind == "block-cyclic-reduction" and solver.perm is not None and solver.b < 6 * 41
    b = rng.standard_normal(n)
    x = torch.zeros(n, dtype=torch.float64)
    solver.apply(torch.from_numpy(b.copy()), x)
    want = spla.spsolve(sp.csc_matrix(As), b)
    assert np.linalg.norm(x.numpy() - want) / np.linalg.norm(want) < 1e-10


def test_auto_strategy_switches_to_cyclic_reduction_when_banded_factors_get_large(monkeypatch):
    Ac = galerkin_operator(64)
    dA = DeviceCSR.from_scipy(Ac, "cpu")
    assert coarse.make_coarse_solver(dA, shim, "auto").kind == "grid-block"
    monkeypatch.setattr(coarse, "BANDED_MAX_BYTES", 1 << 20)
    monkeypatch.setattr(coarse, "DENSE_PREFERRED_BYTES", 1 << 20)
    assert coarse.make_coarse_solver(dA, shim, "auto").kind == "block-cyclic-reduction"


def test_grid_block_solver_matches_superlu():
    """Blocks cut in both grid directions (GridBlockSolver): 9-point Galerkin operator (r = 1), 25-point operator of an
    L2-type transfer (r = 2), a non-square grid; against SuperLU to rounding, with far fewer dense bytes than whole-line
    strips; numeric refactorisation on the same pattern; accumulate mode."""
    cases = [galerkin_operator(64), _l2_galerkin(41)]
    A, _ = P.poisson_2d_structured(64)
    Pm = P.tensor_interpolator_2d(65)
    G = sp.csr_matrix(Pm.T @ A @ Pm)                       # 33 x 33 -> cut a 33 x 24 sub-grid out of it (still a grid operator)
    keep = np.flatnonzero((np.arange(33 * 33) // 33) < 64)
    cases.append(sp.csr_matrix(sp.kron(sp.identity(3), G).tocsr()[:33 * 64][:, :33 * 64] + sp.diags(np.ones(33 * 64 - 33), 33) * 0.01
                               + sp.diags(np.ones(33 * 64 - 33), -33) * 0.01))
    for Ac in cases:
        Ac = sp.csr_matrix(Ac)
        Ac.sort_indices()
        n = Ac.shape[0]
        grid = coarse.GridBlockSolver.detect_grid(n, Ac)
        assert grid is not None, n
        dA = DeviceCSR.from_scipy(Ac, "cpu")
        solver = coarse.make_coarse_solver(dA, shim, "grid")
        assert solver.kind == "grid-block" and solver.k >= 4
        rng = np.random.default_rng(2)
        b = rng.standard_normal(n)
        x = torch.zeros(n, dtype=torch.float64)
        solver.apply(torch.from_numpy(b.copy()), x)
        want = spla.spsolve(sp.csc_matrix(Ac), b)
        err = np.linalg.norm(x.numpy() - want) / np.linalg.norm(want)
        assert err < 1e-12, (n, err)
        x2 = x.clone()
        solver.apply(torch.from_numpy(b.copy()), x2, accumulate=True)
        assert np.allclose(x2.numpy(), 2 * x.numpy(), rtol=1e-14, atol=0)
        # new values, same pattern
        A2 = Ac.copy()
        A2.data = A2.data * (1.0 + 0.05 * rng.random(A2.nnz))
        solver.factor(DeviceCSR.from_scipy(A2, "cpu"))
        solver.apply(torch.from_numpy(b.copy()), x)
        want2 = spla.spsolve(sp.csc_matrix(A2), b)
        assert np.linalg.norm(x.numpy() - want2) / np.linalg.norm(want2) < 1e-11
    # the planner prefers it to whole-line strips on a 129^2 grid and needs well under half their bytes
    plan = coarse.GridBlockSolver.plan(129, 129, 1)
    k, s = coarse.BandedBlockSolver.plan(129 * 129, 130)
    assert plan is not None and plan[2] < 0.5 * 8 * (2 * k * s * s + (129 * 129 - k * s) ** 2)


def test_grid_detection_refuses_non_grid_operators():
    A, _ = P.poisson_2d_structured(48)
    rng = np.random.default_rng(1)
    p = rng.permutation(A.shape[0])
    Ap = sp.csr_matrix(A[p][:, p])
    assert coarse.GridBlockSolver.detect_grid(Ap.shape[0], Ap) is None
    A1, _ = P.poisson_1d_fd(4096)
    assert coarse.GridBlockSolver.detect_grid(A1.shape[0], sp.csr_matrix(A1)) is None
