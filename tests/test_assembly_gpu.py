"""Device P1 assembly (csrc/assembly.hip) against the reference's own assembly (goldens g4:
StiffnessMatrix / MassMatrix / LoadVector on Mesh2D(k*k)) and against the vectorised host
assembler on a jittered, variable-coefficient mesh (-m gpu)."""
import numpy as np
import pytest
import scipy.sparse as sp

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from conftest import load_golden, coo_from                       # noqa: E402
from learnmultigrid_amd import problems as P                      # noqa: E402
from learnmultigrid_amd.assembly import P1Mesh2D                  # noqa: E402


@pytest.mark.parametrize("k", [4, 16])
def test_matches_reference_assembly(k):
    g = load_golden("g4_structured2d_k%d" % k)
    mesh = P1Mesh2D(g["p"], g["conn"], "cuda:0")
    A, M, rhs = mesh.assemble(load=-1.0)
    Aref, Mref = coo_from(g, "A_free"), coo_from(g, "M")
    assert abs(A.to_scipy() - Aref).max() <= 1e-13
    assert abs(M.to_scipy() - Mref).max() <= 1e-15
    # g["rhs"] has the Dirichlet rows zeroed (thesis_structured_2d.py:414): compare interior nodes
    p = g["p"]
    interior = (p[:, 0] > 0) & (p[:, 0] < 1) & (p[:, 1] > 0) & (p[:, 1] < 1)
    np.testing.assert_allclose(rhs.cpu().numpy()[interior], g["rhs"].ravel()[interior], rtol=1e-12)
    assert abs(float(M.vals.sum()) - 1.0) < 1e-12                # area of the unit square


def test_jittered_variable_coefficient_mesh_and_determinism():
    m = 96
    s = m + 1
    h = 1.0 / m
    gx = np.linspace(0, 1, s)
    px, py = np.tile(gx, s), np.repeat(gx, s)
    rng = np.random.default_rng(5)
    idx = np.arange(s * s)
    inter = ((idx % s) > 0) & ((idx % s) < m) & ((idx // s) > 0) & ((idx // s) < m)
    px = np.where(inter, px + rng.uniform(-0.25 * h, 0.25 * h, s * s), px)
    py = np.where(inter, py + rng.uniform(-0.25 * h, 0.25 * h, s * s), py)
    coeff = np.exp(0.5 * rng.standard_normal(2 * m * m))
    sq = (np.arange(m)[:, None] * s + np.arange(m)[None, :]).ravel()
    conn = np.concatenate([np.stack([sq, sq + 1, sq + s + 1], 1), np.stack([sq, sq + s + 1, sq + s], 1)], 0)
    want, det = P.p1_stiffness_2d(px, py, m, coeff)
    mesh = P1Mesh2D(np.stack([px, py], 1), conn, "cuda:0")
    A, M, rhs = mesh.assemble(load=-1.0, coeff=coeff)
    got = A.to_scipy()
    assert abs(got - want).max() <= 1e-12 * abs(want).max()
    A2, _, _ = mesh.assemble(mass=False, coeff=coeff)
    assert torch.equal(A.vals, A2.vals)                          # atomic-free: identical bits
    wrhs = np.zeros(s * s)
    np.add.at(wrhs, conn.ravel(), np.repeat(-det / 6.0, 3))
    np.testing.assert_allclose(rhs.cpu().numpy(), wrhs, rtol=1e-11, atol=1e-18)
    # the assembled operator drives the solver like any other matrix
    from learnmultigrid_amd.solvers import HierarchyMG
    Ad, bd = P.apply_dirichlet_identity_rows(got, rhs.cpu().numpy().reshape(-1, 1), ~inter)
    mg = HierarchyMG(Ad, bd, P.geometric_hierarchy_2d(s, 4))
    mg.solve(levels=4, smoother="Jacobi", smooth_steps=3, max_iterations=30, error=1e-9,
             smoother_semantics="as_named", omega=0.8)
    assert mg.get_residual() <= 1e-9


# ---- 1-D L2-projection coupling operator on the device (csrc/l2proj.hip) ---------------------------
@pytest.mark.parametrize("ne", [16, 64, 1024])
def test_l2_projection_device_regular_nested(ne):
    """Q "pseudo" / "quasi" of L2Projection(...).compute_transfer_1d() (L2Projection.py:74-90) built by the HIP
    kernels against the operators the reference itself produced (goldens g2), and B against the host restatement."""
    from conftest import load_golden, coo_from
    from learnmultigrid_amd import l2_projection as L2
    g = load_golden("g2_poisson1d_ne%d" % ne)
    xf, xc = np.linspace(0, 1, ne + 1), np.linspace(0, 1, ne // 2 + 1)
    for kind in ("pseudo", "quasi"):
        Q = L2.transfer_1d_device(kind, xf, xc, "cuda:0").to_scipy()
        assert abs(Q - coo_from(g, "Q_" + kind)).max() <= 1e-13, kind
        assert Q.has_sorted_indices and np.diff(Q.indptr).max() <= 3
    B = L2.transfer_1d_device("B", xf, xc, "cuda:0").to_scipy()
    assert abs(B - L2.coupling_operator_1d(xf, xc)).max() <= 1e-16


@pytest.mark.parametrize("case", ["g3_ne32", "g3_ne256", "non_nested", "coarse_finer_than_fine"])
def test_l2_projection_device_irregular(case):
    from conftest import load_golden, coo_from
    from learnmultigrid_amd import l2_projection as L2
    rng = np.random.default_rng(9)
    if case.startswith("g3"):
        g = load_golden("g3_fem1d_" + case[3:])
        xf = g["x"]
        xc = xf[0::2]
    elif case == "non_nested":                       # no coincident interior nodes at all
        xf = np.sort(np.concatenate([[0.0, 1.0], rng.random(300)]))
        xc = np.sort(np.concatenate([[0.0, 1.0], rng.random(77)]))
    else:                                            # locally the "coarse" mesh is the finer one: long rows of B
        xf = np.sort(np.concatenate([[0.0, 1.0], rng.random(20)]))
        xc = np.sort(np.concatenate([[0.0, 1.0], rng.random(400)]))
    for kind in ("B", "pseudo", "quasi"):
        got = L2.transfer_1d_device(kind, xf, xc, "cuda:0").to_scipy()
        want = L2.coupling_operator_1d(xf, xc) if kind == "B" else L2.transfer_1d(kind, xf, xc)
        assert abs(got - want).max() <= 1e-13 * max(1.0, abs(want).max()), (case, kind)
    if case.startswith("g3"):
        Q = L2.transfer_1d_device("quasi", xf, xc, "cuda:0").to_scipy()
        assert abs(Q - coo_from(g, "Q_quasi")).max() <= 1e-13
        np.testing.assert_allclose(np.asarray(Q.sum(axis=1)).ravel(), 1.0, rtol=1e-13)
    with pytest.raises(ValueError):
        L2.transfer_1d_device("L2", xf, xc, "cuda:0")
    with pytest.raises(ValueError):
        L2.transfer_1d_device("quasi", xf[::-1], xc, "cuda:0")
