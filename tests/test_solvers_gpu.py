"""The drop-in solver classes on an MI355X against (a) golden vectors captured from the
reference and (b) the CPU oracle on seeded inputs (-m gpu).

Bar (BASELINE.json north_star): level indexing bit-exact; per-iteration residual norms
within 1e-10 relative.  Entries that sit at the rounding floor of a run (<= 1e-14 of the
largest entry) cannot be reproduced to any relative accuracy by anybody, including the
reference run twice with a different BLAS, hence the absolute floor.
"""
import numpy as np
import pytest
import scipy.sparse as sp

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from conftest import load_golden, coo_from                      # noqa: E402
from learnmultigrid_amd import problems as P                     # noqa: E402
from learnmultigrid_amd.solvers import (DirectSolver, Jacobi, GaussSeidel, GeometricMG,  # noqa: E402
                                        SemiGeometricMG, HierarchyMG, Multigrid)
from oracle import vcycle_ref as V                               # noqa: E402  (checker only)

RTOL = 1e-10


def assert_track(got, want, rtol=RTOL, floor=1e-14):
    assert got.shape == want.shape, (got.shape, want.shape)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=floor * float(np.max(want)))


def test_g1_small_solvers_match_reference():
    g = load_golden("g1_small_solvers")
    A, rhs = g["A"], g["rhs"]
    d = DirectSolver(A, rhs)
    d.solve()
    np.testing.assert_allclose(d.get_solution(), g["direct_solution"], rtol=1e-13)
    assert d.get_residual() <= 1e-14
    j = Jacobi(A, rhs)
    j.solve()
    assert j.get_iterations() == int(g["jacobi_iterations"])
    assert_track(j.get_track_res(), g["jacobi_track"])
    np.testing.assert_allclose(j.get_solution(), g["jacobi_solution"], rtol=1e-12)
    s = GaussSeidel(A, rhs)
    s.solve()
    assert s.get_iterations() == int(g["gs_iterations"])
    assert_track(s.get_track_res(), g["gs_track"], 1e-9)
    np.testing.assert_allclose(s.get_solution(), g["gs_solution"], rtol=1e-12)
    assert s.get_track_res().shape[1] == 1 and s.get_dimension() == 3


@pytest.mark.parametrize("ne", [16, 64, 1024])
def test_g2_residual_histories_match_reference(ne):
    g = load_golden("g2_poisson1d_ne%d" % ne)
    A, rhs = coo_from(g, "A"), g["rhs"]
    for kind in ("pseudo", "quasi"):
        Q = coo_from(g, "Q_" + kind)
        for levels, steps in ((2, 1), (2, 3), (3, 1)):
            key = "semi_%s_L%d_s%d" % (kind, levels, steps)
            m = SemiGeometricMG(A, rhs, Q)
            m.solve(smoother="GaussSeidel", smooth_steps=steps, levels=levels,
                    max_iterations=100, error=1e-11)
            assert m.get_iterations() == int(g[key + "_iterations"]), key
            assert_track(m.get_track_res(), g[key + "_track"])
            np.testing.assert_allclose(m.get_solution(), g[key + "_solution"], rtol=1e-9, atol=1e-13)
            assert m.level_dims[0] == ne + 1 and m.level_dims[1] == Q.shape[1]
    for levels, steps in ((2, 1), (3, 3), (4, 2)):
        key = "geo_L%d_s%d" % (levels, steps)
        m = GeometricMG(A, rhs)
        m.solve(smoother="GaussSeidel", smooth_steps=steps, levels=levels,
                max_iterations=100, error=1e-11)
        assert m.get_iterations() == int(g[key + "_iterations"]), key
        assert_track(m.get_track_res(), g[key + "_track"])
        assert m.level_dims == V.level_sizes(ne + 1, levels)            # bit-exact level indexing
    m = GeometricMG(A, rhs)
    m.solve()                                     # defaults: name "Jacobi" ignored as shipped
    assert m.get_iterations() == int(g["geo_default_iterations"])
    assert_track(m.get_track_res(), g["geo_default_track"])
    assert m.get_track_res()[0, 0] == np.sqrt(ne + 1)                   # iteration-1 quirk


@pytest.mark.parametrize("ne", [16, 64])
def test_g2_initial_guess_and_vcycle_signature(ne):
    g = load_golden("g2_poisson1d_ne%d" % ne)
    A, rhs = coo_from(g, "A"), g["rhs"]
    x0 = g["x0"].copy()
    m = GeometricMG(A, rhs)
    m.solve(levels=2, smooth_steps=2, max_iterations=3, error=1e-30, initial_guess=x0)
    assert_track(m.get_track_res(), g["geo_x0_track"])
    np.testing.assert_allclose(m.get_solution(), g["geo_x0_solution"], rtol=1e-10, atol=1e-14)
    assert np.array_equal(x0, g["x0"])                                   # not mutated by default
    m = GeometricMG(A, rhs)
    m.solve(levels=2, smooth_steps=2, max_iterations=3, error=1e-30, initial_guess=x0,
            mutate_initial_guess=True)
    np.testing.assert_allclose(x0, g["geo_x0_mutated_guess"], rtol=1e-11, atol=1e-15)
    m = GeometricMG(A, rhs)
    u0 = np.zeros((ne + 1, 1))
    u = m.v_cycle(m.get_matrix(), u0, rhs, "GaussSeidel", 2, 1e-8, 2)
    np.testing.assert_allclose(u, g["vcycle_u"], rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(u0, g["vcycle_u0_after"], rtol=1e-11, atol=1e-15)


@pytest.mark.parametrize("ne", [16, 64, 1024])
def test_g2_standalone_smoothers(ne):
    g = load_golden("g2_poisson1d_ne%d" % ne)
    A, rhs = coo_from(g, "A"), g["rhs"]
    j = Jacobi(A, rhs)
    j.solve(max_iterations=25)
    assert_track(j.get_track_res(), g["jacobi25_track"])
    np.testing.assert_allclose(j.get_solution(), g["jacobi25_solution"], rtol=1e-12, atol=1e-18)
    if ne <= 64:
        s = GaussSeidel(A, rhs)
        s.solve(max_iterations=25)
        assert_track(s.get_track_res(), g["gs25_track"])
        np.testing.assert_allclose(s.get_solution(), g["gs25_solution"], rtol=1e-11, atol=1e-18)


@pytest.mark.parametrize("ne", [32, 256])
def test_g3_learned_like_q_matches_reference(ne):
    g = load_golden("g3_fem1d_ne%d" % ne)
    A, rhs = coo_from(g, "A"), g["rhs"]
    for name in ("learned", "quasi"):
        Q = coo_from(g, "Q_" + name)
        m = SemiGeometricMG(A, rhs, Q)
        m.solve(levels=2, smoother="GaussSeidel", smooth_steps=3, error=1e-10, max_iterations=15)
        assert m.get_iterations() == int(g[name + "_iterations"])
        assert_track(m.get_track_res(), g[name + "_track"])
    # dense Q input (the reference scripts pass ndarrays) gives the same run
    m2 = SemiGeometricMG(A.toarray(), rhs, coo_from(g, "Q_learned").toarray())
    m2.solve(levels=2, smoother="GaussSeidel", smooth_steps=3, error=1e-10, max_iterations=15)
    assert_track(m2.get_track_res(), g["learned_track"])


def oracle_run(A, rhs, hier, **kw):
    ref = V.RefMultigrid(A, rhs.copy(), hierarchy=hier)
    ref.solve(**kw)
    return ref


@pytest.mark.parametrize("m,levels", [(64, 3), (128, 4)])
@pytest.mark.parametrize("mode", ["as_shipped", "jacobi", "gs_named"])
def test_2d_structured_hierarchy_matches_oracle(m, levels, mode):
    A, rhs = P.poisson_2d_structured(m)
    hier = P.geometric_hierarchy_2d(m + 1, levels)
    kw = dict(levels=levels, smooth_steps=3, max_iterations=12, error=1e-9)
    if mode == "as_shipped":
        okw = dict(smoother="Jacobi", semantics="as_shipped")
        gkw = dict(smoother="Jacobi")
    elif mode == "jacobi":
        okw = dict(smoother="Jacobi", semantics="as_named", omega=0.8)
        gkw = dict(smoother="Jacobi", smoother_semantics="as_named", omega=0.8)
    else:
        okw = dict(smoother="GaussSeidel", semantics="as_named")
        gkw = dict(smoother="GaussSeidel", smoother_semantics="as_named")
    ref = oracle_run(A, rhs, hier, **kw, **okw)
    mg = HierarchyMG(A, rhs.copy(), hier)
    mg.solve(**kw, **gkw)
    assert mg.get_iterations() == ref.iterations
    assert_track(mg.get_track_res(), ref.track_res)
    assert mg.level_dims == ref.level_dims == [s * s for s in P.level_sizes(m + 1, levels)]
    np.testing.assert_allclose(mg.get_solution(), ref.solution, rtol=1e-9, atol=1e-13)
    assert mg.get_track_res()[-1, 0] < 1e-4 * mg.get_track_res()[1, 0]   # it actually converges


def test_cfg2_513_three_levels_jacobi_and_graph_replay():
    """BASELINE config #2: 2-D structured P1 Poisson 512x512, 3-level V-cycle, geometric transfer."""
    m, levels = 512, 3
    A, rhs = P.poisson_2d_structured(m)
    hier = P.geometric_hierarchy_2d(m + 1, levels)
    kw = dict(levels=levels, smoother="Jacobi", smooth_steps=3, max_iterations=8, error=1e-9)
    ref = oracle_run(A, rhs, hier, semantics="as_named", omega=0.8, **kw)
    mg = HierarchyMG(A, rhs.copy(), hier)
    mg.solve(smoother_semantics="as_named", omega=0.8, **kw)
    assert_track(mg.get_track_res(), ref.track_res)
    assert mg.level_dims == [513 * 513, 257 * 257, 129 * 129]
    mg2 = HierarchyMG(A, rhs.copy(), hier)
    mg2.solve(smoother_semantics="as_named", omega=0.8, use_graph=True, **kw)
    assert np.array_equal(mg2.get_track_res(), mg.get_track_res())       # hipGraph replay == eager
    assert np.array_equal(mg2.get_solution(), mg.get_solution())


@pytest.mark.parametrize("m,levels", [(128, 4), (512, 3)])
@pytest.mark.parametrize("steps", [1, 2, 3, 4, 7])
def test_fused_smoothing_passes_equal_separate_sweeps(m, levels, steps):
    """The cycle with its Jacobi sweeps (and the residual) fused into single passes (lmg_stencil_smooth,
    chunks of <= 3 sweeps) against the same cycle with one launch per sweep: identical histories and
    iterates, eager and replayed from a hipGraph."""
    from learnmultigrid_amd import ops
    A, rhs = P.poisson_2d_structured(m)
    hier = P.geometric_hierarchy_2d(m + 1, levels)
    kw = dict(levels=levels, smoother="Jacobi", smooth_steps=steps, max_iterations=6, error=1e-30,
              smoother_semantics="as_named", omega=0.8)
    runs = {}
    min_rows, min_tr, min_tile = ops.FUSED_MIN_ROWS, ops.FUSED_TRANSFER_MIN_ROWS, ops.TILED_MIN_ROWS
    try:
        ops.TILED_MIN_ROWS = 0
        for fused in (False, "reg", "tile"):
            # "reg": the register-blocked passes (with the transfers folded in) on every level -- the product only
            # uses them on multi-million-row levels; "tile": the product's choice at these sizes, LDS-tiled passes
            ops.set_fused_enabled(bool(fused))
            ops.FUSED_MIN_ROWS = 0 if fused == "reg" else min_rows
            ops.FUSED_TRANSFER_MIN_ROWS = 0 if fused == "reg" else min_tr
            for graph in (False, True):
                mg = HierarchyMG(A, rhs.copy(), hier)
                mg.solve(use_graph=graph, **kw)
                kinds = [ops._fused_kind(lev.A) for lev in mg._hier.levels[:-1]]
                assert kinds == [fused or None] * len(kinds), kinds
                runs[fused, graph] = (mg.get_track_res(), mg.get_solution())
    finally:
        ops.set_fused_enabled(True)
        ops.FUSED_MIN_ROWS = min_rows
        ops.FUSED_TRANSFER_MIN_ROWS = min_tr
        ops.TILED_MIN_ROWS = min_tile
    t0, x0 = runs[False, False]
    for key, (t, x) in runs.items():
        assert np.array_equal(t, t0) and np.array_equal(x, x0), key
    if steps == 3:
        ref = oracle_run(A, rhs, hier, semantics="as_named", **{k: v for k, v in kw.items() if k != "smoother_semantics"})
        assert_track(t0, ref.track_res)


def test_default_smoother_graph_replay_2d():
    """solve(use_graph=True) with the DEFAULT semantics (forward Gauss-Seidel whatever the name says,
    Multigrid.py:88) on a 2-D grid whose level sets fit the one-workgroup executor: everything that
    executor builds lazily must exist before the capture starts; replay == eager, bit for bit."""
    m, levels = 64, 3
    A, rhs = P.poisson_2d_structured(m)
    hier = P.geometric_hierarchy_2d(m + 1, levels)
    kw = dict(levels=levels, smoother="GaussSeidel", smooth_steps=3, max_iterations=8, error=1e-12)
    mg = HierarchyMG(A, rhs.copy(), hier)
    mg.solve(**kw)
    mg2 = HierarchyMG(A, rhs.copy(), hier)
    mg2.solve(use_graph=True, **kw)
    assert np.array_equal(mg2.get_track_res(), mg.get_track_res())
    assert np.array_equal(mg2.get_solution(), mg.get_solution())
    ref = oracle_run(A, rhs, hier, semantics="as_shipped", **kw)
    assert_track(mg2.get_track_res(), ref.track_res)


def test_cfg1_weighted_jacobi_two_level_vcycle_ne1024():
    """BASELINE config #1 as worded: 1-D Poisson, 1024 elements, 2-level V-cycle, WEIGHTED JACOBI
    (smoother_semantics="as_named"), geometric and pseudo-L2 transfer, against the CPU oracle."""
    g = load_golden("g2_poisson1d_ne1024")
    A, rhs = coo_from(g, "A"), g["rhs"]
    Qp = coo_from(g, "Q_pseudo")
    for Q, cls_args in ((None, ()), (Qp, (Qp,))):
        for steps, omega in ((1, 2.0 / 3.0), (3, 0.8), (1, 1.0)):
            kw = dict(levels=2, smoother="Jacobi", smooth_steps=steps, max_iterations=30, error=1e-9)
            ref = V.RefMultigrid(A, rhs.copy(), l2_proj=Q)
            ref.solve(semantics="as_named", omega=omega, **kw)
            mg = GeometricMG(A, rhs.copy()) if Q is None else SemiGeometricMG(A, rhs.copy(), *cls_args)
            mg.solve(smoother_semantics="as_named", omega=omega, **kw)
            assert mg.get_iterations() == ref.iterations
            assert mg.level_dims == ref.level_dims == [1025, 513]
            # the operator is scaled by 1/h^2 ~ 1e6: the residual's own rounding floor is
            # eps * ||A|| * ||x|| ~ 1e-9, so the last entries (~1e-7) carry ~1e-12 of noise
            assert_track(mg.get_track_res(), ref.track_res, floor=1e-13)


def test_learned_like_hierarchy_2d_matches_oracle():
    m, levels = 96, 4                                                    # 97 -> 49 -> 25 -> 13
    A, rhs = P.jittered_poisson_2d(m, seed=42)
    sizes = P.level_sizes(m + 1, levels)
    hier = []
    for l, s in enumerate(sizes[:-1]):
        base = sp.kron(P.pseudo_l2_interpolator_1d(s), P.pseudo_l2_interpolator_1d(s)).tocsr()
        hier.append(P.learned_like(base, 43 + l))
    kw = dict(levels=levels, smoother="GaussSeidel", smooth_steps=3, max_iterations=10, error=1e-9)
    ref = oracle_run(A, rhs, hier, semantics="as_shipped", **kw)
    mg = HierarchyMG(A, rhs.copy(), hier)
    mg.solve(**kw)
    assert mg.get_iterations() == ref.iterations
    assert_track(mg.get_track_res(), ref.track_res)


def test_multicolor_gs_converges_and_is_deterministic():
    A, rhs = P.poisson_2d_structured(128)
    hier = P.geometric_hierarchy_2d(129, 4)
    runs = []
    for _ in range(2):
        mg = HierarchyMG(A, rhs.copy(), hier)
        mg.solve(levels=4, smoother="GaussSeidel", smooth_steps=2, max_iterations=12, error=1e-9,
                 smoother_semantics="as_named", gs_mode="multicolor")
        runs.append(mg.get_track_res())
    assert np.array_equal(runs[0], runs[1])
    assert runs[0][-1, 0] <= 1e-9 or runs[0][-1, 0] < 1e-7 * runs[0][1, 0]


def test_error_conventions():
    A, rhs = P.poisson_1d_fd(16)
    mg = GeometricMG(A, rhs)
    with pytest.raises(ValueError):
        mg.solve(levels=1)
    with pytest.raises(ValueError):
        mg.solve(cycle="W")
    with pytest.raises(ValueError):
        mg.solve(smoother="SOR")
    with pytest.raises(ValueError):
        mg.solve(smoother="CG", smoother_semantics="as_named")
    with pytest.raises(ValueError):
        SemiGeometricMG(A, rhs, np.ones((5, 3))).solve()
    assert isinstance(mg, Multigrid) and mg.label == "GeometricMG"


def test_galerkin_rebuild_numeric_only():
    """Config #5's "RAP rebuild": new coefficients, same pattern -> numeric SpGEMM only."""
    from learnmultigrid_amd.hierarchy import Hierarchy
    from learnmultigrid_amd.ops import DeviceCSR
    m, levels = 64, 3
    A1, _ = P.jittered_poisson_2d(m, seed=42, coeff_sigma=0.5, coeff_seed=44)
    A2, _ = P.jittered_poisson_2d(m, seed=42, coeff_sigma=0.5, coeff_seed=45)
    assert np.array_equal(A1.indices, A2.indices)
    hier = P.geometric_hierarchy_2d(m + 1, levels)
    H = Hierarchy(A1, hier, "cuda:0")
    H.rebuild_numeric(torch.from_numpy(A2.data.copy()).to("cuda:0"))
    want = A2
    for Pm in hier:
        want = sp.csr_matrix(Pm.T @ want @ Pm)
    got = H.levels[-1].A.to_scipy()
    assert abs(got - want).max() <= 1e-13 * abs(want).max()


def test_mass_matrix_coarsening_on_the_device():
    """M_coarse = Q^T M Q (Multigrid.py:273-275, :763) for every level by the hierarchy's SpGEMM plans: equal
    to SciPy's sparse product bit for bit; a numeric refresh after the mass values changed."""
    from learnmultigrid_amd.hierarchy import Hierarchy
    from learnmultigrid_amd.assembly import P1Mesh2D
    m, levels = 64, 3
    A, _ = P.poisson_2d_structured(m)
    mesh = P1Mesh2D.structured(m, "cuda:0") if hasattr(P1Mesh2D, "structured") else None
    rng = np.random.default_rng(3)
    # a mass-like matrix on A's pattern (positive, symmetric pattern)
    M = sp.csr_matrix((rng.random(A.nnz) + 0.1, A.indices, A.indptr), shape=A.shape)
    hier = P.geometric_hierarchy_2d(m + 1, levels)
    H = Hierarchy(A, hier, "cuda:0", mass=M)
    want = M
    for l, Pm in enumerate(hier):
        want = sp.csr_matrix(sp.csr_matrix(Pm.T @ want) @ Pm)
        got = H.levels[l + 1].M.to_scipy()
        assert got.shape == want.shape and abs(got - want).max() == 0.0, l
    M2 = sp.csr_matrix((rng.random(A.nnz) + 0.1, A.indices, A.indptr), shape=A.shape)
    H.rebuild_mass_numeric(torch.from_numpy(M2.data.copy()).to("cuda:0"))
    want = M2
    for l, Pm in enumerate(hier):
        want = sp.csr_matrix(sp.csr_matrix(Pm.T @ want) @ Pm)
        assert abs(H.levels[l + 1].M.to_scipy() - want).max() == 0.0
    with pytest.raises(ValueError):
        Hierarchy(A, hier, "cuda:0", mass=M[:100, :100])


def test_full_size_properties_4097():
    """cfg#4 size (4097^2, 6 levels), size-independent properties instead of an oracle run:
    monotone residual history with a multigrid-like factor, linearity in the rhs, and
    agreement of the fused norm with a norm computed from the downloaded residual."""
    m, levels = 4096, 6
    A, rhs = P.poisson_2d_structured(m)
    hier = P.geometric_hierarchy_2d(m + 1, levels)
    mg = HierarchyMG(A, rhs.copy(), hier)
    mg.solve(levels=levels, smoother="Jacobi", smooth_steps=3, max_iterations=8, error=1e-30,
             smoother_semantics="as_named", omega=0.8, use_graph=True)
    t = mg.get_track_res().ravel()
    assert mg.level_dims == [s * s for s in (4097, 2049, 1025, 513, 257, 129)]
    assert np.all(t[2:] < 0.25 * t[1:-1])                               # V(3,3) contraction
    r = mg.get_residual_vector()
    assert abs(np.linalg.norm(r) - t[-1]) <= 1e-12 * t[-1]
    mg2 = HierarchyMG(A, 2.0 * rhs, hier)
    mg2.solve(levels=levels, smoother="Jacobi", smooth_steps=3, max_iterations=8, error=1e-30,
              smoother_semantics="as_named", omega=0.8, use_graph=True)
    assert np.array_equal(mg2.get_solution(), 2.0 * mg.get_solution())   # exact: scaling by 2
    # the fine level runs its sweeps fused (lmg_stencil_smooth); one launch per sweep gives the same bits,
    # and the fused pass itself matches the CPU oracle's separate sweeps at full size
    from learnmultigrid_amd import ops
    from oracle import kernels as K
    assert ops.stencil_smooth_available(mg._hier.levels[0].A)
    assert [ops._fused_kind(lev.A) for lev in mg._hier.levels[:-1]] == ["reg", "tile", "tile", "tile", "tile"]
    try:
        ops.set_fused_enabled(False)
        mg3 = HierarchyMG(A, rhs.copy(), hier)
        mg3.solve(levels=levels, smoother="Jacobi", smooth_steps=3, max_iterations=8, error=1e-30,
                  smoother_semantics="as_named", omega=0.8)
    finally:
        ops.set_fused_enabled(True)
    assert np.array_equal(mg3.get_track_res(), mg.get_track_res())
    assert np.array_equal(mg3.get_solution(), mg.get_solution())
    n = A.shape[0]
    rng = np.random.default_rng(5)
    x0, b = rng.standard_normal(n), rng.standard_normal(n)
    Ac = K.as_csr(A)
    want = x0
    for _ in range(3):
        want = K.jacobi(Ac, want, b, 0.8)
    wr, _ = K.residual(Ac, want, b)
    dA = mg._hier.levels[0].A
    dx, db = torch.from_numpy(x0).to("cuda:0"), torch.from_numpy(b).to("cuda:0")
    out, r = torch.empty_like(dx), torch.empty_like(dx)
    ops.stencil_smooth(dA, dx, db, 0.8, 3, out, r)
    assert np.array_equal(out.cpu().numpy(), want) and np.array_equal(r.cpu().numpy(), wr)
    want0 = np.zeros(n)
    for _ in range(2):
        want0 = K.jacobi(Ac, want0, b, 0.8)
    ops.stencil_smooth(dA, None, db, 0.8, 2, out, None)
    assert np.array_equal(out.cpu().numpy(), want0)
    # ... and the post-smoothing pass with the coarse-grid correction folded in (what the cycle above ran on the
    # two finest levels): prolongation + 3 sweeps of the oracle, bitwise
    lev0, lev1 = mg._hier.levels[0], mg._hier.levels[1]
    assert all(ops.stencil_smooth_prolong_available(lev.A, lev.P) for lev in mg._hier.levels[:-1])
    Pm = K.as_csr(hier[0])
    e = rng.standard_normal(Pm.shape[1])
    want = K.spmv(Pm, e, x0.copy(), 1.0, 1.0)
    for _ in range(3):
        want = K.jacobi(Ac, want, b, 0.8)
    ops.stencil_smooth(dA, dx, db, 0.8, 3, out, None, prolong=(lev0.P, torch.from_numpy(e).to("cuda:0")))
    assert np.array_equal(out.cpu().numpy(), want)
    # ... and the pre-smoothing pass with the restriction folded in (no residual vector is written)
    assert all(ops.stencil_smooth_restrict_available(lev.A, lev.R) for lev in mg._hier.levels[:-1])
    want = x0
    for _ in range(3):
        want = K.jacobi(Ac, want, b, 0.8)
    wbc = K.spmv(K.as_csr(sp.csr_matrix(hier[0]).T), K.residual(Ac, want, b)[0])
    bc = torch.empty(Pm.shape[1], dtype=torch.float64, device="cuda:0")
    ops.stencil_smooth(dA, dx, db, 0.8, 3, out, None, restrict=(lev0.R, bc))
    assert np.array_equal(out.cpu().numpy(), want) and np.array_equal(bc.cpu().numpy(), wbc)


def test_cfg4_history_4097_six_levels_against_the_oracle():
    """cfg#4 itself -- 4097^2, 6 levels, V(3,3) weighted Jacobi, what bench.py times -- against the CPU oracle's
    hoisted cycle (oracle/vcycle_ref.py: R = P^T, Galerkin products and the coarse LU once, then the reference's
    arithmetic per cycle): five residual norms at 1e-10 relative (north_star's bar), ~10 s of CPU."""
    m, levels = 4096, 6
    A, rhs = P.poisson_2d_structured(m)
    hier = P.geometric_hierarchy_2d(m + 1, levels)
    mg = HierarchyMG(A, rhs.copy(), hier)
    mg.solve(levels=levels, smoother="Jacobi", smooth_steps=3, max_iterations=6, error=1e-30,
             smoother_semantics="as_named", omega=0.8, use_graph=True)
    got = mg.get_track_res().ravel()
    ref = V.HoistedVCycle(A, hier)
    b = rhs.ravel()
    x = np.zeros(A.shape[0])
    want = []
    for _ in range(6):
        want.append(np.linalg.norm(b - A @ x))
        x = ref.cycle(x, b, "Jacobi", 3, 0.8)
    want = np.array(want)
    # (entry 0 of the solver's track is the sqrt(n) quirk of Multigrid.py:64-66)
    assert got[0] == np.sqrt(float(A.shape[0]))
    np.testing.assert_allclose(got[1:], want[1:], rtol=RTOL, atol=1e-14 * want.max())
    xs = x_after(ref, b, 5)
    np.testing.assert_allclose(mg.get_solution().ravel(), xs, rtol=0, atol=1e-9 * np.abs(xs).max())


def x_after(ref, b, cycles):
    x = np.zeros(b.size)
    for _ in range(cycles):
        x = ref.cycle(x, b, "Jacobi", 3, 0.8)
    return x


def test_a_timed_out_wavefront_band_raises_instead_of_returning_a_wrong_iterate():
    """gs_wave.hip turns a stalled band into a wrong result and sets a sticky flag once its spin budget runs out; the
    solvers read the flag where they synchronise anyway (after the residual norm of each outer iteration) and raise."""
    from learnmultigrid_amd import ops
    from learnmultigrid_amd._lib import LmgError
    m = 256
    A, rhs = P.poisson_2d_structured(m)
    mg = HierarchyMG(A, rhs.copy(), P.geometric_hierarchy_2d(m + 1, 3))
    kw = dict(levels=3, smoother="GaussSeidel", smooth_steps=2, max_iterations=3, error=1e-30)
    mg.solve(**kw)                                             # as shipped: forward Gauss-Seidel on every level
    lev0 = mg._hier.levels[0].A
    assert ops.stencil_gs_available(lev0) and lev0.stencil._gs_work is not None
    ops.stencil_gs_check(lev0)                                 # clean run: nothing raised
    lev0.stencil._gs_work.view(torch.int32)[0] = 1             # what a timed-out band leaves behind
    try:
        with pytest.raises(LmgError):
            mg.solve(**kw)
        g = GaussSeidel(A, rhs.copy())
        g.solve(max_iterations=1)
        dm = g._device_matrix()
        if dm.stencil is not None and dm.stencil._gs_work is not None:
            dm.stencil._gs_work.view(torch.int32)[0] = 1
            with pytest.raises(LmgError):
                g.solve(max_iterations=2)
    finally:
        lev0.stencil._gs_work.view(torch.int32)[0] = 0


def test_g6_cg_matches_reference():
    from learnmultigrid_amd.solvers import CG
    g = load_golden("g6_cg_ne64")
    A, rhs = coo_from(g, "A"), g["rhs"]
    c = CG(A, rhs)
    c.solve(max_iterations=200, error=1e-10)
    assert c.get_iterations() == int(g["iterations"])
    # CG amplifies rounding differences (the reference scales A by alpha before the product,
    # CG.py:37): compare the history loosely and the solution tightly
    got, want = c.get_track_res(), g["track"]
    assert got.shape == want.shape
    np.testing.assert_allclose(got[:20], want[:20], rtol=1e-8)
    np.testing.assert_allclose(c.get_solution(), g["solution"], rtol=1e-9, atol=1e-12)


def test_multigrid_preconditioned_cg_converges_fast():
    from learnmultigrid_amd.solvers import CG
    from learnmultigrid_amd.hierarchy import Hierarchy
    m = 256
    A, rhs = P.poisson_2d_structured(m)
    # symmetric variant of the problem: eliminate the Dirichlet couplings (interior block)
    s = m + 1
    idx = np.arange(s * s)
    inter = ((idx % s) > 0) & ((idx % s) < m) & ((idx // s) > 0) & ((idx // s) < m)
    keep = sp.diags(inter.astype(float))
    As = sp.csr_matrix(keep @ A @ keep + sp.diags((~inter).astype(float)))
    plain = CG(As, rhs.copy())
    plain.solve(max_iterations=2000, error=1e-8)
    H = Hierarchy(As, P.geometric_hierarchy_2d(s, 5), "cuda:0")
    pcg = CG(As, rhs.copy())
    pcg.solve(max_iterations=100, error=1e-8, preconditioner=H)
    assert pcg.get_iterations() < 15 < plain.get_iterations()
    x_ref = plain.get_solution()
    assert np.linalg.norm(pcg.get_solution() - x_ref) <= 1e-6 * np.linalg.norm(x_ref)


def learned_like_hierarchy(side, levels, seed=43):
    hier = []
    for l, s in enumerate(P.level_sizes(side, levels)[:-1]):
        base = sp.kron(P.pseudo_l2_interpolator_1d(s), P.pseudo_l2_interpolator_1d(s)).tocsr()
        hier.append(P.learned_like(base, seed + l))
    return hier


def test_cfg3_unstructured_like_2M_dof_learned_q_five_levels():
    """BASELINE config #3: ~2 M DoF triangle mesh (1441^2 jittered nodes, 7-point P1 stiffness),
    5-level V-cycle with learned-like (row-stochastic, perturbed L2-type) transfer operators."""
    m, levels = 1440, 5
    A, rhs = P.jittered_poisson_2d(m, seed=42)
    assert A.shape[0] == 2076481
    hier = learned_like_hierarchy(m + 1, levels)
    kw = dict(levels=levels, smooth_steps=3, max_iterations=4, error=1e-30)
    ref = oracle_run(A, rhs, hier, smoother="Jacobi", semantics="as_named", omega=0.8, **kw)
    mg = HierarchyMG(A, rhs.copy(), hier)
    mg.solve(smoother="Jacobi", smoother_semantics="as_named", omega=0.8, **kw)
    assert_track(mg.get_track_res(), ref.track_res)
    assert mg.level_dims == ref.level_dims == [s * s for s in (1441, 721, 361, 181, 91)]
    ref = oracle_run(A, rhs, hier, smoother="GaussSeidel", semantics="as_shipped", **dict(kw, max_iterations=3))
    mg = HierarchyMG(A, rhs.copy(), hier)
    mg.solve(smoother="GaussSeidel", **dict(kw, max_iterations=3))
    assert_track(mg.get_track_res(), ref.track_res)


def test_cfg5_style_variable_coefficient_rebuild_and_solve():
    """BASELINE config #5 at 1025^2 (full size: test_cfg5_full_size_8193 below): variable-coefficient
    stiffness, learned-like Q, Galerkin rebuild (numeric SpGEMM only) after the coefficients
    change, then the solve -- against the oracle on the rebuilt problem."""
    from learnmultigrid_amd.hierarchy import Hierarchy
    m, levels = 1024, 5
    A1, rhs = P.variable_coeff_poisson_2d_structured(m, seed=44)
    A2, _ = P.variable_coeff_poisson_2d_structured(m, seed=45)
    hier = learned_like_hierarchy(m + 1, levels)
    H = Hierarchy(A1, hier, "cuda:0")
    H.rebuild_numeric(torch.from_numpy(A2.data.copy()).to("cuda:0"))
    fine = H.levels[0]
    fine.b.copy_(torch.from_numpy(rhs.ravel().copy()).to("cuda:0"))
    norms = []
    with torch.cuda.stream(H.stream):
        for _ in range(4):
            norms.append(H.residual_norm())
            H.cycle("Jacobi", 3, 0.8)
        norms.append(H.residual_norm())
    ref = oracle_run(A2, rhs, hier, smoother="Jacobi", semantics="as_named", omega=0.8, levels=levels,
                     smooth_steps=3, max_iterations=5, error=1e-30)
    want = ref.track_res.ravel()
    got = np.array(norms)
    np.testing.assert_allclose(got[1:], want[1:], rtol=1e-10)           # (entry 0 is the sqrt(n) quirk)


def test_two_level_learned_q_with_a_large_coarse_level():
    """The scripts' default shape -- SemiGeometricMG(A, rhs, Q).solve(levels=2, smoother="GaussSeidel",
    smooth_steps=3, error=1e-10) on a 2-D problem (test/test_B_patch.py:193-194, test/thesis_compare_2D.py:430-435)
    -- at a size where the coarse `spsolve` (Multigrid.py:106) sees 257^2 = 66 049 unknowns of a 25-point operator:
    too large for a dense inverse, 4 GB for the one-level banded solver, 1.4 GB for block cyclic reduction.
    Jittered triangulation (7-point fine operator, all-distinct values), learned-like L2-type Q; histories
    against the CPU oracle (which re-factorises with SuperLU in every cycle, like the reference)."""
    m = 512
    A, rhs = P.jittered_poisson_2d(m, seed=42)
    l2 = P.pseudo_l2_interpolator_1d(m + 1)
    Q = P.learned_like(sp.kron(l2, l2).tocsr(), 43)
    assert Q.shape == (513 * 513, 257 * 257)
    kw = dict(levels=2, smoother="GaussSeidel", smooth_steps=3, error=1e-10, max_iterations=4)
    mg = SemiGeometricMG(A, rhs.copy(), Q)
    mg.solve(**kw)
    assert mg._hier.coarse.kind in ("grid-block", "block-cyclic-reduction") and mg.level_dims == [513 * 513, 257 * 257]
    assert mg._hier.coarse.bytes_per_apply() < 2 << 30
    ref = V.RefMultigrid(A, rhs.copy(), l2_proj=Q)
    ref.solve(**kw)
    assert mg.get_iterations() == ref.iterations
    assert_track(mg.get_track_res(), ref.track_res)
    np.testing.assert_allclose(mg.get_solution(), ref.solution, rtol=1e-8, atol=1e-12)


def test_cfg5_full_size_8193():
    """BASELINE config #5 at FULL size on one GPU: 8193^2 = 67 125 249 DoF variable-coefficient stiffness
    (nnz = 335 M, close to the int32 limit), learned-like Q, 7 levels, ~25 GB resident plus the 32.7 GB
    SpGEMM replay map.  Size-independent properties + bitwise checks of single kernels against the C
    oracle (a full oracle solve would take minutes): level sizes, V(3,3) contraction over 4 cycles,
    fused norm vs downloaded residual, one fine-level packed sweep bitwise, numeric rebuild by replay
    bit-identical to the rebuild by sort."""
    from learnmultigrid_amd import ops
    from learnmultigrid_amd.hierarchy import Hierarchy
    from oracle import kernels as K
    m, levels = 8192, 7
    A, rhs = P.variable_coeff_poisson_2d_structured(m, seed=44)
    n = A.shape[0]
    assert n == 8193 * 8193 and A.nnz > 300e6 and A.indices.dtype == np.int32
    hier = learned_like_hierarchy(m + 1, levels)
    H = Hierarchy(A, hier, "cuda:0")
    assert H.sizes == [s * s for s in (8193, 4097, 2049, 1025, 513, 257, 129)]
    fine = H.levels[0]
    assert isinstance(fine.A.packed, ops.PackedCSR) and fine.A.patterns is None       # all-distinct values
    # (a) one fine-level Jacobi sweep and one residual, bitwise against the C oracle
    rng = np.random.default_rng(8193)
    x, b = rng.standard_normal(n), rng.standard_normal(n)
    dx, db = torch.from_numpy(x).to("cuda:0"), torch.from_numpy(b).to("cuda:0")
    Ac = K.as_csr(A)
    with torch.cuda.stream(H.stream):
        ops.csr_jacobi(fine.A, dx, db, 0.8, fine.tmp)
        got = fine.tmp.cpu().numpy()
    assert np.array_equal(got, K.jacobi(Ac, x, b, 0.8))
    with torch.cuda.stream(H.stream):
        ops.csr_residual_norm2(fine.A, dx, db, fine.tmp, H.partials, H.norm2)
        got, n2 = fine.tmp.cpu().numpy(), H.norm2.item()
    wr, wn2 = K.residual(Ac, x, b)
    assert np.array_equal(got, wr) and abs(n2 - wn2) <= 1e-13 * wn2
    del dx, db, x, b, got, wr
    # (b) 4 cycles: contraction and fused norm vs the norm of the downloaded residual
    fine.b.copy_(torch.from_numpy(rhs.ravel().copy()).to("cuda:0"))
    norms = []
    with torch.cuda.stream(H.stream):
        ops.zero(fine.x)
        for _ in range(5):
            norms.append(H.residual_norm())
            H.cycle("Jacobi", 3, 0.8)
        norms.append(H.residual_norm())
        r = H.outer_r.cpu().numpy()
    t = np.array(norms)
    # (the first cycle turns the smooth initial error into a small rough one -- the 5 % noise of the
    # learned-like transfers -- whose RESIDUAL is larger; from then on every cycle contracts)
    assert np.all(t[2:] < 0.25 * t[1:-1]), t
    assert abs(np.linalg.norm(r) - t[-1]) <= 1e-12 * t[-1]
    # (c) numeric Galerkin rebuild: 1st re-run sorts and records the product map, 2nd replays it
    newv = fine.A.vals.clone()
    H.rebuild_numeric(newv)
    sorted_vals = [lev.A.vals.clone() for lev in H.levels[1:]]
    H.rebuild_numeric(newv)
    assert all(lev.plan_RAP.recorded_bytes() > 0 for lev in H.levels[:-1])
    for lev, want in zip(H.levels[1:], sorted_vals):
        assert torch.equal(lev.A.vals, want)


@pytest.mark.parametrize("ne,levels,steps", [(15, 2, 1), (15, 3, 2), (16, 4, 1), (33, 3, 0), (64, 6, 1)])
def test_edge_cases_match_oracle(ne, levels, steps):
    """Even n (the interpolator quirk of Multigrid.py:139-142), coarsest grids of 2-3 unknowns,
    zero smoothing steps: same histories as the CPU oracle."""
    A, rhs = P.poisson_1d_fd(ne)
    for sem, name, om in (("as_shipped", "GaussSeidel", 1.0), ("as_named", "Jacobi", 0.7)):
        ref = V.RefMultigrid(A, rhs.copy())
        ref.solve(levels=levels, smoother=name, smooth_steps=steps, max_iterations=12, error=1e-12,
                  semantics=sem, omega=om)
        mg = GeometricMG(A, rhs.copy())
        mg.solve(levels=levels, smoother=name, smooth_steps=steps, max_iterations=12, error=1e-12,
                 smoother_semantics=sem, omega=om)
        assert mg.get_iterations() == ref.iterations
        assert mg.level_dims == ref.level_dims
        assert_track(mg.get_track_res(), ref.track_res, floor=1e-13)


def test_degenerate_calls():
    A, rhs = P.poisson_1d_fd(32)
    mg = GeometricMG(A, rhs.copy())
    mg.solve(levels=2, max_iterations=1)                      # one started iteration, no convergence
    assert mg.get_iterations() == 1 and mg.get_track_res().shape == (1, 1)
    assert mg.get_track_res()[0, 0] == np.sqrt(33) and np.array_equal(mg.get_residual_vector(), np.ones((33, 1)))
    mg = GeometricMG(A, rhs.copy())
    mg.solve(levels=2, error=100.0)                           # sqrt(33) <= 100: stops before any cycle
    assert mg.get_iterations() == 1 and not mg.get_solution().any()
    mg = GeometricMG(A, np.zeros((33, 1)))
    mg.solve(levels=3, max_iterations=5, error=1e-12)         # zero rhs: exact after the first check
    assert mg.get_iterations() == 2 and mg.get_track_res()[1, 0] == 0.0
    x0 = np.linspace(0, 1, 33).reshape(-1, 1)
    ref = V.RefMultigrid(A, rhs.copy())
    ref.solve(levels=3, smoother="Jacobi", smooth_steps=2, max_iterations=6, error=1e-30,
              initial_guess=x0.copy(), semantics="as_named", omega=0.8)
    mg = GeometricMG(A, rhs.copy())
    mg.solve(levels=3, smoother="Jacobi", smooth_steps=2, max_iterations=6, error=1e-30,
             initial_guess=x0.copy(), smoother_semantics="as_named", omega=0.8)
    assert_track(mg.get_track_res(), ref.track_res)
    # solving twice with the same object keeps counting iterations like the reference (no reset)
    mg.solve(levels=3, max_iterations=2, error=1e-30)
    assert mg.get_iterations() == 8


def test_g7_neuralmg_constructor_and_multilevel_run():
    from learnmultigrid_amd.solvers import NeuralMG
    from test_oracle_golden import ReplayModel
    g = load_golden("g7_neuralmg_ne64")
    A, rhs, M = coo_from(g, "A"), g["rhs"], g["M"]
    mg = NeuralMG(A, rhs, ReplayModel(g), M, np.ones(7), np.zeros(7))
    mg.solve(levels=3, smoother="GaussSeidel", smooth_steps=3, error=1e-10, max_iterations=12)
    assert mg.get_iterations() == int(g["iterations"])
    assert_track(mg.get_track_res(), g["track"])
    assert mg.level_dims == [65, 33, 17] and mg.label == "NeuralMG"
    with pytest.raises(ValueError):
        NeuralMG(A[:64, :64], rhs[:64], ReplayModel(g), M[:64, :64], np.ones(7), np.zeros(7)).solve(levels=2)
