"""learnmultigrid_amd/l2_projection.py against the transfer operators the reference's
L2Projection produced (goldens g2: regular nested meshes; g3: seeded irregular mesh)."""
import numpy as np
import pytest

from conftest import load_golden, coo_from
from learnmultigrid_amd import l2_projection as L2


@pytest.mark.parametrize("ne", [16, 64, 1024])
def test_regular_nested_meshes(ne):
    g = load_golden("g2_poisson1d_ne%d" % ne)
    xf, xc = np.linspace(0, 1, ne + 1), np.linspace(0, 1, ne // 2 + 1)
    for kind in ("pseudo", "quasi"):
        Q = L2.transfer_1d(kind, xf, xc)
        Qref = coo_from(g, "Q_" + kind)
        assert abs(Q - Qref).max() <= 1e-13, kind
        # same sparsity class as the reference's Q: 3 nnz on coincident rows, 2 in between
        Q.data[np.abs(Q.data) < 1e-14] = 0
        Q.eliminate_zeros()
        assert np.diff(Q.indptr).max() <= 3


def test_irregular_mesh_quasi_and_l2():
    g = load_golden("g3_fem1d_ne32")
    xf = g["x"]
    xc = xf[0::2]
    Q = L2.transfer_1d("quasi", xf, xc)
    assert abs(Q - coo_from(g, "Q_quasi")).max() <= 1e-13
    np.testing.assert_allclose(np.asarray(Q.sum(axis=1)).ravel(), 1.0, rtol=1e-13)
    M = L2.mass_matrix_1d(xf)
    assert abs(M.toarray() - g["M"]).max() <= 1e-16
    # the true L2 projection reproduces coarse functions exactly: Q @ 1 = 1 and Q @ x_c = x_f
    Ql2 = L2.transfer_1d("L2", xf, xc)
    np.testing.assert_allclose(Ql2 @ np.ones(xc.size), 1.0, atol=1e-12)
    np.testing.assert_allclose(Ql2 @ xc, xf, atol=1e-12)
    with pytest.raises(ValueError):
        L2.transfer_1d("cubic", xf, xc)
