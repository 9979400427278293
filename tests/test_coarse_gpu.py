"""Coarsest-level solvers with the real kernels (-m gpu): the grid-block solver (blocks cut in both grid directions,
lmg_coarse_front_gather / lmg_coarse_back_gather) against SuperLU, the reference's per-cycle `spsolve`
(Multigrid.py:106); refinement switched off by the measured accuracy."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from learnmultigrid_amd import coarse, ops, problems as P                    # noqa: E402
from learnmultigrid_amd.hierarchy import Hierarchy                           # noqa: E402
from learnmultigrid_amd.ops import DeviceCSR                                 # noqa: E402

DEV = "cuda:0"


def galerkin_operator(m, coarsenings=1):
    A, _ = P.poisson_2d_structured(m * 2 ** coarsenings)
    side = m * 2 ** coarsenings + 1
    for _ in range(coarsenings):
        Pm = P.tensor_interpolator_2d(side)
        A = sp.csr_matrix(Pm.T @ A @ Pm)
        side = (side - 1) // 2 + 1
    A.sort_indices()
    return A


def l2_galerkin(side):
    A, _ = P.jittered_poisson_2d(2 * (side - 1), seed=42)
    l2 = P.pseudo_l2_interpolator_1d(2 * side - 1)
    Q = P.learned_like(sp.kron(l2, l2).tocsr(), 43)
    Ac = sp.csr_matrix(Q.T @ A @ Q)
    Ac.sort_indices()
    return Ac


@pytest.mark.parametrize("case", ["9pt_129", "9pt_65", "25pt_81"])
def test_grid_block_solver_on_the_device_matches_superlu(case):
    Ac = {"9pt_129": lambda: galerkin_operator(128, 2), "9pt_65": lambda: galerkin_operator(64),
          "25pt_81": lambda: l2_galerkin(81)}[case]()
    n = Ac.shape[0]
    dA = DeviceCSR.from_scipy(Ac, DEV)
    solver = coarse.make_coarse_solver(dA, ops, "auto")
    assert solver.kind == "grid-block", solver.kind
    banded = coarse.make_coarse_solver(dA, ops, "banded")
    assert solver.bytes_per_apply() < 0.6 * banded.bytes_per_apply()
    rng = np.random.default_rng(3)
    b = rng.standard_normal(n)
    want = spla.spsolve(sp.csc_matrix(Ac), b)
    db = torch.from_numpy(b).to(DEV)
    x = torch.full((n,), float("nan"), dtype=torch.float64, device=DEV)
    solver.apply(db, x)
    got = x.cpu().numpy()
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-12
    # x += A^-1 b in the solver's last launch (the refinement form)
    solver.apply(db, x, accumulate=True)
    np.testing.assert_allclose(x.cpu().numpy(), 2 * got, rtol=1e-13)
    # same answer as the strip solver to rounding
    y = torch.zeros(n, dtype=torch.float64, device=DEV)
    banded.apply(db, y)
    np.testing.assert_allclose(y.cpu().numpy(), got, rtol=0, atol=1e-11 * np.abs(want).max())
    # new values on the same pattern: numeric phase only, equal to a fresh build
    A2 = Ac.copy()
    A2.data = A2.data * (1.0 + 0.05 * rng.random(A2.nnz))
    d2 = DeviceCSR.from_scipy(A2, DEV)
    solver.factor(d2)
    solver.apply(db, x)
    fresh = coarse.make_coarse_solver(d2, ops, "grid")
    fresh.apply(db, y)
    assert torch.equal(x, y)
    want2 = spla.spsolve(sp.csc_matrix(A2), b)
    assert np.linalg.norm(x.cpu().numpy() - want2) / np.linalg.norm(want2) < 1e-11


def test_hierarchy_measures_the_coarse_solver_and_skips_the_refinement():
    m = 512
    A, rhs = P.poisson_2d_structured(m)
    H = Hierarchy(A, P.geometric_hierarchy_2d(m + 1, 3), DEV)
    assert H.coarse.kind == "grid-block" and H.levels[-1].n == 129 * 129
    assert H.coarse_residual < 1e-12 and H.coarse_refine == 0
    H1 = Hierarchy(A, P.geometric_hierarchy_2d(m + 1, 3), DEV, coarse_refine=1)
    assert H1.coarse_refine == 1
    for hh in (H, H1):
        with torch.cuda.stream(hh.stream):
            hh.levels[0].b.copy_(torch.from_numpy(rhs.ravel().copy()).to(DEV))
            ops.zero(hh.levels[0].x)
            hh.hist = [hh.residual_norm()]
            for _ in range(5):
                hh.cycle("Jacobi", 3, 0.8)
                hh.hist.append(hh.residual_norm())
    # (entries at the rounding floor of the run cannot agree to any relative accuracy: absolute floor as in test_solvers_gpu)
    np.testing.assert_allclose(H.hist, H1.hist, rtol=1e-10, atol=1e-14 * max(H1.hist))
