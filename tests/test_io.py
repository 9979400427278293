"""learnmultigrid_amd/io.py against files written by the reference's own save() methods
(tests/golden/g5_saved_*, produced by tools/make_golden.py) and round trips."""
import os

import numpy as np
import scipy.sparse as sp

from conftest import GOLDEN, load_golden, coo_from
from learnmultigrid_amd import io as lio, problems as P


def test_loads_files_written_by_the_reference():
    A = lio.load_matrix(os.path.join(GOLDEN, "g5_saved_A.npz"))
    g = load_golden("g4_structured2d_k4")
    assert A.shape == (25, 25)
    assert abs(A - coo_from(g, "A_free")).max() <= 1e-15        # same assembly as the k=4 golden
    rhs = lio.load_rhs(os.path.join(GOLDEN, "g5_saved_rhs.npy"))
    assert rhs.shape == (25, 1) and abs(rhs.sum() + 1.0) < 1e-12  # integral of f = -1 over the unit square


def test_round_trips(tmp_path):
    A, rhs = P.poisson_2d_structured(8)
    lio.save_matrix(str(tmp_path / "A"), A)
    B = lio.load_matrix(str(tmp_path / "A.npz"))
    assert (A != B).nnz == 0
    y = np.load(str(tmp_path / "A.npz"))
    assert sorted(y.files) == ["col", "data", "row", "shape"]      # the reference's key names
    lio.save_rhs(str(tmp_path / "rhs"), rhs)
    assert np.array_equal(lio.load_rhs(str(tmp_path / "rhs.npy")), rhs)
    Q = P.tensor_interpolator_2d(9)
    lio.save_transfer(str(tmp_path / "Q"), Q)
    Q2 = lio.load_transfer(str(tmp_path / "Q.npy"))
    assert (Q != Q2).nnz == 0 and sp.isspmatrix_csr(Q2)


def test_mat_problem(tmp_path):
    from scipy.io import savemat
    A, rhs = P.poisson_2d_structured(4)
    Q = P.tensor_interpolator_2d(5)
    p = np.random.default_rng(0).random((25, 2))
    conn1 = np.array([[1, 2, 7], [1, 7, 6]])                      # 1-based like MATLAB
    savemat(str(tmp_path / "prob.mat"), {"A": A, "M": sp.identity(25), "rhs": rhs, "Q": Q,
                                          "mesh": {"p": p, "conn": conn1}})
    d = lio.load_mat_problem(str(tmp_path / "prob.mat"))
    assert (d["A"] != A).nnz == 0 and (d["Q"] != Q).nnz == 0
    assert np.array_equal(d["rhs"], rhs) and np.array_equal(d["conn"], conn1 - 1)
    assert np.allclose(d["p"], p)
