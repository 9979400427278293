"""Host-side generators (learnmultigrid_amd/problems.py) against the reference's own
assemblies captured in tests/golden (bit-exact where the arithmetic is integer or
exactly representable)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from conftest import load_golden, coo_from
from learnmultigrid_amd import problems as P


@pytest.mark.parametrize("ne", [16, 64, 1024])
def test_poisson_1d_matches_reference(ne):
    g = load_golden("g2_poisson1d_ne%d" % ne)
    A, rhs = P.poisson_1d_fd(ne)
    Aref = coo_from(g, "A")
    assert abs(A - Aref).max() <= 1e-12 * abs(Aref).max()
    assert A.nnz == Aref.nnz
    np.testing.assert_allclose(rhs, g["rhs"], rtol=1e-13, atol=1e-18)


def test_geometric_interpolator_bit_exact():
    g = load_golden("g2_interpolators")
    for n in (2, 3, 9, 10, 17, 64, 1025):
        got = P.geometric_interpolator_1d(n)
        assert np.array_equal(got.toarray(), g["interp_%d" % n]), n
    assert P.level_sizes(1025, 7) == [int(v) for v in g["level_sizes_from_1025"]]


@pytest.mark.parametrize("k", [4, 16])
def test_structured_2d_matches_reference_assembly(k):
    g = load_golden("g4_structured2d_k%d" % k)
    A, rhs = P.poisson_2d_structured(k)
    Aref = coo_from(g, "A")
    assert abs(A - Aref).max() <= 1e-12
    # the reference assembly leaves ~1e-17 hypotenuse couplings; without them it is 5-point
    Aref.data[np.abs(Aref.data) < 1e-12] = 0
    Aref.eliminate_zeros()
    assert A.nnz == Aref.nnz
    np.testing.assert_allclose(rhs, g["rhs"], rtol=1e-12, atol=1e-18)
    # the general P1 assembler reproduces the unconstrained stiffness as well
    s = k + 1
    gx = np.linspace(0, 1, s)
    Afree, det = P.p1_stiffness_2d(np.tile(gx, s), np.repeat(gx, s), k)
    assert abs(Afree - coo_from(g, "A_free")).max() <= 1e-12
    assert abs(float(coo_from(g, "M").sum()) - 1.0) < 1e-12


def test_tensor_interpolator_partition_of_unity_and_rap_stencil():
    s = 17
    Pm = P.tensor_interpolator_2d(s)
    assert Pm.shape == (s * s, 81)
    assert np.array_equal(np.asarray(Pm.sum(axis=1)).ravel(), np.ones(s * s))
    A, _ = P.poisson_2d_structured(s - 1)
    Ac = sp.csr_matrix(Pm.T @ A @ Pm)
    assert Ac.shape == (81, 81)
    assert np.diff(Ac.indptr).max() <= 9


def test_learned_like_is_row_stochastic_and_seeded():
    Pm = P.tensor_interpolator_2d(9)
    Q1, Q2 = P.learned_like(Pm, 43), P.learned_like(Pm, 43)
    assert (Q1 != Q2).nnz == 0
    np.testing.assert_allclose(np.asarray(Q1.sum(axis=1)).ravel(), 1.0, rtol=1e-14)
    assert Q1.nnz == Pm.nnz


def test_jittered_problem_is_solvable():
    A, rhs = P.jittered_poisson_2d(16, seed=42)
    assert np.diff(A.indptr).max() <= 7
    x = spla.spsolve(sp.csc_matrix(A), rhs)
    assert np.all(np.isfinite(x)) and x.min() < 0
    Av, _ = P.jittered_poisson_2d(16, seed=42, coeff_sigma=0.5)
    assert abs(Av - A).max() > 1e-3


def test_variable_coefficient_generator_matches_general_assembler():
    m = 12
    s = m + 1
    rng = np.random.default_rng(3)
    coeff = np.exp(0.5 * rng.standard_normal(2 * m * m))
    g = np.linspace(0, 1, s)
    Afree, _ = P.p1_stiffness_2d(np.tile(g, s), np.repeat(g, s), m, coeff)
    A, rhs = P.variable_coeff_poisson_2d_structured(m, coeff=coeff)
    idx = np.arange(s * s)
    inter = ((idx % s) > 0) & ((idx % s) < m) & ((idx // s) > 0) & ((idx // s) < m)
    Ad, _ = P.apply_dirichlet_identity_rows(Afree, rhs * 0, ~inter)
    assert abs(A - Ad).max() < 1e-13
    assert np.diff(A.indptr).max() == 5
    A1, _ = P.variable_coeff_poisson_2d_structured(m, seed=44)
    A2, _ = P.variable_coeff_poisson_2d_structured(m, seed=45)
    assert np.array_equal(A1.indices, A2.indices) and abs(A1 - A2).max() > 1e-3   # same pattern: RAP rebuild
