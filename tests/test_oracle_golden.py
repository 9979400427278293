"""The CPU oracle (oracle/) against golden vectors captured from the reference
(tools/make_golden.py, run in the build container; fixtures in tests/golden/).

Tolerances: integer/index results bit-exact; residual histories 1e-12 relative
(the oracle calls the same SciPy routines the reference does, so most agree to
the last bit; the bound leaves room for a different SciPy build on the GPU box).
"""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden, coo_from
from oracle import kernels as K
from oracle import vcycle_ref as V

RTOL = 1e-12


def assert_track(got, want, rtol=RTOL, floor=1e-14):
    """Residual histories: relative tolerance, plus an absolute floor of
    `floor` x the largest entry (entries at the rounding floor of the run are
    not reproducible to any relative accuracy, even by the reference itself)."""
    assert got.shape == want.shape, (got.shape, want.shape)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=floor * float(np.max(want)))


def test_g1_small_solvers():
    g = load_golden("g1_small_solvers")
    A, rhs = g["A"], g["rhs"]
    d = V.RefDirect(A, rhs)
    d.solve()
    np.testing.assert_allclose(d.solution, g["direct_solution"], rtol=1e-14)
    j = V.RefJacobi(A, rhs)
    j.solve()
    assert j.iterations == int(g["jacobi_iterations"]) == 66
    assert_track(j.track_res, g["jacobi_track"], 1e-10)   # tail entries are ~1e-13 absolute
    np.testing.assert_allclose(j.solution, g["jacobi_solution"], rtol=1e-13)
    s = V.RefGaussSeidel(A, rhs)
    s.solve()
    assert s.iterations == int(g["gs_iterations"]) == 24
    assert_track(s.track_res, g["gs_track"], 1e-9)
    np.testing.assert_allclose(s.solution, g["gs_solution"], rtol=1e-13)


def test_g2_interpolator_bit_exact():
    g = load_golden("g2_interpolators")
    for n in (2, 3, 9, 10, 17, 64, 1025):
        want = g["interp_%d" % n]
        got = V.geometric_interpolator_1d(n)
        assert got.shape == want.shape
        assert np.array_equal(got, want), n                 # level indexing: bit-exact
    assert V.level_sizes(1025, 7) == [int(v) for v in g["level_sizes_from_1025"]]


@pytest.mark.parametrize("ne", [16, 64, 1024])
def test_g2_poisson_1d_histories(ne):
    g = load_golden("g2_poisson1d_ne%d" % ne)
    A, rhs = coo_from(g, "A"), g["rhs"]
    for kind in ("pseudo", "quasi"):
        Q = coo_from(g, "Q_" + kind)
        for levels, steps in ((2, 1), (2, 3), (3, 1)):
            key = "semi_%s_L%d_s%d" % (kind, levels, steps)
            m = V.RefMultigrid(A, rhs, l2_proj=Q)
            m.solve(smoother="GaussSeidel", smooth_steps=steps, levels=levels,
                    max_iterations=100, error=1e-11)
            assert m.iterations == int(g[key + "_iterations"]), key
            assert_track(m.track_res, g[key + "_track"], 1e-9)
            np.testing.assert_allclose(m.solution, g[key + "_solution"], rtol=1e-10, atol=1e-14)
    for levels, steps in ((2, 1), (3, 3), (4, 2)):
        key = "geo_L%d_s%d" % (levels, steps)
        m = V.RefMultigrid(A, rhs)
        m.solve(smoother="GaussSeidel", smooth_steps=steps, levels=levels,
                max_iterations=100, error=1e-11)
        assert m.iterations == int(g[key + "_iterations"]), key
        assert_track(m.track_res, g[key + "_track"], 1e-9)
        assert m.level_dims == V.level_sizes(ne + 1, levels)
    m = V.RefMultigrid(A, rhs)
    m.solve()                                             # all defaults, smoother name ignored
    assert m.iterations == int(g["geo_default_iterations"])
    assert_track(m.track_res, g["geo_default_track"], 1e-9)
    assert m.track_res[0, 0] == np.sqrt(ne + 1)           # iteration-1 quirk, Multigrid.py:64-66


@pytest.mark.parametrize("ne", [16, 64])
def test_g2_initial_guess_is_mutated_and_vcycle_signature(ne):
    g = load_golden("g2_poisson1d_ne%d" % ne)
    A, rhs = coo_from(g, "A"), g["rhs"]
    x0 = g["x0"].copy()
    m = V.RefMultigrid(A, rhs)
    m.solve(levels=2, smooth_steps=2, max_iterations=3, error=1e-30, initial_guess=x0)
    assert_track(m.track_res, g["geo_x0_track"], 1e-11)
    np.testing.assert_allclose(m.solution, g["geo_x0_solution"], rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(x0, g["geo_x0_mutated_guess"], rtol=1e-11, atol=1e-15)
    m = V.RefMultigrid(A, rhs)
    u0 = np.zeros((ne + 1, 1))
    u = m.v_cycle(m.matrix, u0, rhs, "GaussSeidel", 2, 1e-8, 2)
    np.testing.assert_allclose(u, g["vcycle_u"], rtol=1e-12, atol=1e-16)
    np.testing.assert_allclose(u0, g["vcycle_u0_after"], rtol=1e-12, atol=1e-16)


@pytest.mark.parametrize("ne", [16, 64, 1024])
def test_g2_standalone_smoothers(ne):
    g = load_golden("g2_poisson1d_ne%d" % ne)
    A, rhs = coo_from(g, "A"), g["rhs"]
    j = V.RefJacobi(A, rhs)
    j.solve(max_iterations=25)
    assert_track(j.track_res, g["jacobi25_track"])
    np.testing.assert_allclose(j.solution, g["jacobi25_solution"], rtol=1e-13, atol=1e-18)
    # the C sweep used on the device-parity side must give the same Jacobi iterate
    x = np.zeros(ne + 1)
    for _ in range(25):
        x = K.jacobi(A, x, rhs, 1.0)
    np.testing.assert_allclose(x, g["jacobi25_solution"].ravel(), rtol=1e-13, atol=1e-18)
    if ne <= 64:
        s = V.RefGaussSeidel(A, rhs)
        s.solve(max_iterations=25)
        assert_track(s.track_res, g["gs25_track"], 1e-11)
        np.testing.assert_allclose(s.solution, g["gs25_solution"], rtol=1e-12, atol=1e-18)


@pytest.mark.parametrize("ne", [32, 256])
def test_g3_fem1d_learnedlike(ne):
    g = load_golden("g3_fem1d_ne%d" % ne)
    A, rhs = coo_from(g, "A"), g["rhs"]
    for name in ("learned", "quasi"):
        Q = coo_from(g, "Q_" + name)
        np.testing.assert_allclose(np.asarray(Q.sum(axis=1)).ravel(), 1.0, rtol=1e-13)
        m = V.RefMultigrid(A, rhs, l2_proj=Q)
        m.solve(levels=2, smoother="GaussSeidel", smooth_steps=3, error=1e-10, max_iterations=15)
        assert m.iterations == int(g[name + "_iterations"])
        assert_track(m.track_res, g[name + "_track"], 1e-8)


def test_c_kernels_match_scipy_bitwise():
    rng = np.random.default_rng(0)
    A = sp.random(300, 300, density=0.03, random_state=1, format="csr") + sp.identity(300) * 4
    A = K.as_csr(A)
    x, b = rng.standard_normal(300), rng.standard_normal(300)
    assert np.array_equal(K.matvec(A, x), A @ x)
    r, n2 = K.residual(A, x, b)
    assert np.array_equal(r, b - A @ x)
    assert abs(n2 - float(r @ r)) <= 1e-13 * n2
    assert np.array_equal(K.jacobi(A, x, b, 1.0), x + (1.0 / A.diagonal()) * (b - A @ x))
    # Gauss-Seidel sweep == (D+L)^-1 form of the reference's GaussSeidel.py:23-37
    xs = x.copy()
    K.gs_forward(A, xs, b, 1)
    DL = sp.tril(A).tocsr()
    want = x + sp.linalg.spsolve_triangular(DL, b - A @ x, lower=True)
    np.testing.assert_allclose(xs, want, rtol=1e-12, atol=1e-14)
    # a row list in natural order is the forward sweep
    xr = x.copy()
    K.gs_rows(A, xr, b, np.arange(300))
    assert np.array_equal(xr, xs)


def test_hoisted_cycle_matches_faithful_cycle():
    g = load_golden("g2_poisson1d_ne64")
    A, rhs = coo_from(g, "A"), g["rhs"]
    sizes = V.level_sizes(65, 3)
    hier = [sp.csr_matrix(V.geometric_interpolator_1d(n)) for n in sizes[:-1]]
    m = V.RefMultigrid(A, rhs, hierarchy=hier)
    u0 = np.zeros((65, 1))
    u = m.v_cycle(m.matrix, u0, rhs, "GaussSeidel", 2, 1e-8, 3)
    h = V.HoistedVCycle(A, hier)
    uh = h.cycle(np.zeros(65), rhs, "GaussSeidel", 2)
    np.testing.assert_allclose(uh, u.ravel(), rtol=1e-11, atol=1e-15)


class ReplayModel:
    """Feeds back the predictions recorded from the reference run (fixture g7), keyed by the
    number of patches, and checks that it is shown the same features."""

    def __init__(self, g):
        self.table = {g["features_l0"].shape[0]: (g["features_l0"], g["pred_l0"]),
                      g["features_l1"].shape[0]: (g["features_l1"], g["pred_l1"])}
        self.calls = 0

    def predict(self, data):
        feats, pred = self.table[data.shape[0]]
        np.testing.assert_allclose(data, feats, rtol=1e-11, atol=1e-16)
        self.calls += 1
        return pred


def test_g7_neuralmg_multilevel():
    from conftest import load_golden, coo_from
    g = load_golden("g7_neuralmg_ne64")
    A, rhs, M = coo_from(g, "A"), g["rhs"], g["M"]
    model = ReplayModel(g)
    m = V.RefNeuralMG(A, rhs, model, M, np.ones(7), np.zeros(7))
    m.solve(levels=3, smoother="GaussSeidel", smooth_steps=3, error=1e-10, max_iterations=12)
    assert m.iterations == int(g["iterations"])
    assert_track(m.track_res, g["track"], 1e-9)
    assert model.calls == int(g["n_predict_calls"])          # asked again in every cycle, like the reference
    # the product's vectorised scatter is bit-identical to the reference's
    from learnmultigrid_amd import learned_q as LQ
    assert np.array_equal(LQ.patch_features(M), g["features_l0"])
    g3 = load_golden("g3_fem1d_ne32")
    assert np.array_equal(LQ.transfer_from_predictions(g3["fake_pred"], g3["M"]), coo_from(g3, "Q_learned").toarray())
