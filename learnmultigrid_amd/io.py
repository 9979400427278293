"""On-disk formats of the reference, host side:

  * matrices: COO in an .npz with keys row, col, data, shape
    (assembly/StiffnessMatrix.py:38-51, assembly/MassMatrix.py:37-50);
  * right-hand sides: a plain .npy column (assembly/LoadVector.py:36-43);
  * transfer operators: a dense .npy matrix (test/thesis_prepare_virtual.py:438);
  * MATLAB problems: a .mat with A, M, rhs, Q and mesh{p, conn} (conn is 1-based;
    test/thesis_compare_2D.py:379-386).
Files written by the reference load here and vice versa; nothing executes from the files
(np.load without pickle, scipy.io.loadmat)."""
import numpy as np
import scipy.sparse as sp


def save_matrix(path, A):
    """np.savez(path, row=, col=, data=, shape=) of the COO form, like StiffnessMatrix.save."""
    C = sp.coo_matrix(A)
    np.savez(path, row=C.row, col=C.col, data=C.data, shape=C.shape)


def load_matrix(path, fmt="csr"):
    y = np.load(path, allow_pickle=False)
    M = sp.coo_matrix((y["data"], (y["row"], y["col"])), shape=tuple(int(v) for v in y["shape"]))
    return M.asformat(fmt)


def save_rhs(path, rhs):
    np.save(path, np.asarray(rhs, dtype=np.float64))


def load_rhs(path):
    rhs = np.load(path, allow_pickle=False)
    return np.asarray(rhs, dtype=np.float64).reshape(-1, 1)


def save_transfer(path, Q):
    """Dense .npy like the scripts that export learned Q."""
    np.save(path, Q.toarray() if sp.issparse(Q) else np.asarray(Q, dtype=np.float64))


def load_transfer(path):
    """Dense .npy -> CSR (zeros dropped, exactly what csr_matrix(l2_proj) does, Multigrid.py:182)."""
    return sp.csr_matrix(np.load(path, allow_pickle=False))


def load_mat_problem(path):
    """{'A','M','rhs','Q','p','conn'} from a MATLAB file; conn converted to 0-based."""
    from scipy.io import loadmat
    m = loadmat(path)
    out = {}
    for k in ("A", "M", "Q"):
        if k in m:
            out[k] = sp.csr_matrix(m[k])
    if "rhs" in m:
        out["rhs"] = np.asarray(m["rhs"], dtype=np.float64).reshape(-1, 1)
    if "mesh" in m:
        mesh = m["mesh"]
        out["p"] = np.asarray(mesh["p"][0, 0], dtype=np.float64)
        out["conn"] = np.asarray(mesh["conn"][0, 0], dtype=np.int64) - 1
    return out
