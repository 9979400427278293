"""Host-side construction of the 1-D learned transfer operator from a model's predictions --
the scatter of NeuralMG.transfer_op / prepare_nn_input / construct_B
(learn_multigrid/solvers/Multigrid.py:306-370), vectorised.  The network itself is the
caller's object: anything with `.predict(array (npatch, 7)) -> array (npatch, >= 9)`; no
TensorFlow is imported here (the reference ships no model files either).

Patch k is centred on fine node 2k+2 and reads 7 mass-matrix entries (rows 2k+1..2k+3 of the
tridiagonal M); of the predicted coupling stencil the reference keeps 5 numbers:
    B[2k+2, k]   = patch[2]        B[2k+1 : 2k+4, k+1] = patch[4:7]        B[2k+2, k+2] = patch[8]
The first / last two rows are closed by the row-sum constraint sum_j B_ij = sum_j M_ij
(:359-363) and Q = B / rowsum(B) (:367-368)."""
import numpy as np


def patch_features(M):
    """data_M of prepare_nn_input (Multigrid.py:313-334): (npatch, 7), npatch = (n-1)/2 - 1."""
    M = np.asarray(M.todense()) if hasattr(M, "todense") else np.asarray(M)
    n = M.shape[0]
    i = np.arange(1, n - 2, 2)
    i = i[: (n - 1) // 2 - 1]
    return np.stack([M[i, i - 1], M[i, i], M[i, i + 1], M[i + 1, i + 1], M[i + 1, i + 2],
                     M[i + 2, i + 2], M[i + 2, i + 3]], axis=1)


def transfer_from_predictions(pred, M):
    """Q (n x (n-1)/2+1, dense like the reference's) from the model output `pred`."""
    M = np.asarray(M.todense()) if hasattr(M, "todense") else np.asarray(M)
    n = M.shape[0]
    nc = (n - 1) // 2 + 1
    B = np.zeros((n, nc))
    k = np.arange(pred.shape[0])
    B[2 * k + 2, k] = pred[:, 2]
    for off in range(3):
        B[2 * k + 1 + off, k + 1] = pred[:, 4 + off]
    B[2 * k + 2, k + 2] = pred[:, 8]
    diff = M.sum(axis=1) - B.sum(axis=1)
    B[0:2, 0] = diff[0:2]
    B[-2:, -1] = diff[-2:]
    return B / B.sum(axis=1)[:, np.newaxis]


def learned_transfer(model, M, mean, std):
    """NeuralMG.transfer_op (Multigrid.py:306-311)."""
    data = (patch_features(M) - mean) / std
    return transfer_from_predictions(np.asarray(model.predict(data)), M)
