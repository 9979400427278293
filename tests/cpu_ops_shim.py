"""TEST-ONLY stand-in for learnmultigrid_amd.ops on CPU tensors.

It lets the -m "not gpu" suite exercise the HOST logic that sits above the kernels
(hierarchy recursion, row-block partitioning, ghost layouts, halo plans, collectives over
gloo) by routing every kernel call to the CPU oracle.  It lives under tests/ on purpose:
the product package never imports it and has no CPU path.
"""
import numpy as np
import scipy.sparse as sp
import torch

from learnmultigrid_amd.ops import DeviceCSR, F64, I32   # pure-torch container, device agnostic
from oracle import kernels as K


def _np(t):
    return t.numpy()


def _sp(A):
    return sp.csr_matrix((_np(A.vals), _np(A.colidx), _np(A.rowptr)), shape=A.shape)


def partials_count(n):
    return max(1024, (n + 255) // 256)


def _raw(A):
    """Raw CSR arrays in STORAGE order (no canonicalisation: the local matrices of the
    distributed solver keep the global entry order after their columns are renumbered)."""
    return A.shape[0], _np(A.rowptr), _np(A.colidx), _np(A.vals)


def csr_residual_norm2(A, x, b, r, partials, norm2):
    n, rp, ci, va = _raw(A)
    rr = np.empty(n)
    n2 = K.lib().orc_csr_residual(n, rp, ci, va, np.ascontiguousarray(_np(x)),
                                  np.ascontiguousarray(_np(b)[:n]), rr)
    if r is not None:
        _np(r)[:n] = rr
    if norm2 is not None:
        _np(norm2)[0] = n2


def csr_jacobi(A, x_in, b, omega, x_out):
    # rectangular local matrices (owned rows x [owned | ghosts]): the diagonal is column == row
    n, rp, ci, va = _raw(A)
    out = np.empty(n)
    K.lib().orc_csr_jacobi(n, rp, ci, va, np.ascontiguousarray(_np(x_in)),
                           np.ascontiguousarray(_np(b)[:n]), float(omega), out)
    _np(x_out)[:n] = out


def csr_spmv(A, x, y, alpha=1.0, beta=0.0):
    n, rp, ci, va = _raw(A)
    out = np.ascontiguousarray(_np(y)[:n]).copy()
    K.lib().orc_csr_spmv(n, rp, ci, va, np.ascontiguousarray(_np(x)), out, float(alpha), float(beta))
    _np(y)[:n] = out


def axpby(alpha, x, beta, y):
    yy = _np(y)
    yy[:] = alpha * _np(x) if beta == 0.0 else alpha * _np(x) + beta * yy


def vmul(alpha, x, y, out):
    _np(out)[:] = alpha * (_np(x) * _np(y))


def csr_inverse_diagonal(A):
    n, rp, ci, va = _raw(A)
    rows = np.repeat(np.arange(n), np.diff(rp))
    d = np.zeros(n)
    on = ci == rows
    np.add.at(d, rows[on], va[on])
    out = np.zeros(n)
    np.divide(1.0, d, out=out, where=d != 0)
    return torch.from_numpy(out)


def dot(x, y, partials, out):
    _np(out)[0] = float(np.dot(_np(x), _np(y)))


def copy(src, dst):
    dst.copy_(src)


def zero(x):
    x.zero_()


def gather(idx, x, buf):
    _np(buf)[: idx.numel()] = _np(x)[_np(idx)]


def scatter(idx, buf, x):
    _np(x)[_np(idx)] = _np(buf)[: idx.numel()]


def dense_gemv(M, x, y):
    _np(y)[:] = K.dense_gemv(_np(M), _np(x))


def dense_gemv_blockdiag(M, x, y):
    k, s, _ = M.shape
    for i in range(k):
        _np(y)[i * s:(i + 1) * s] = K.dense_gemv(_np(M[i]), _np(x)[i * s:(i + 1) * s])


def dense_gemv_windows(M, x, x_stride, y, y_stride, z=None, z_stride=0, alpha=1.0):
    nb, rows, cols = M.shape
    Mn, xn, yn = _np(M), _np(x), _np(y)
    zn = None if z is None else _np(z)
    out = [(zn[k * z_stride:k * z_stride + rows] if zn is not None else 0.0)
           + alpha * K.dense_gemv(np.ascontiguousarray(Mn[k]), np.ascontiguousarray(xn[k * x_stride:k * x_stride + cols]))
           for k in range(nb)]
    for k in range(nb):
        yn[k * y_stride:k * y_stride + rows] = out[k]


def dense_gemv_windows_off(M, x, x_offsets, y, y_stride, z=None, z_stride=0, alpha=1.0):
    nb, rows, cols = M.shape
    Mn, xn, yn, off = _np(M), _np(x), _np(y), _np(x_offsets)
    zn = None if z is None else _np(z)
    out = [(zn[k * z_stride:k * z_stride + rows] if zn is not None else 0.0)
           + alpha * K.dense_gemv(np.ascontiguousarray(Mn[k]), np.ascontiguousarray(xn[off[k]:off[k] + cols]))
           for k in range(nb)]
    for k in range(nb):
        yn[k * y_stride:k * y_stride + rows] = out[k]


def coarse_front_gather(M, b, idx, y, tail_idx, tail_out):
    k, s, _ = M.shape
    bn, gi = _np(b), _np(idx)
    seg = np.where(gi >= 0, bn[np.maximum(gi, 0)], 0.0).reshape(k, s)
    for i in range(k):
        _np(y)[i * s:(i + 1) * s] = K.dense_gemv(np.ascontiguousarray(_np(M[i])), np.ascontiguousarray(seg[i]))
    _np(tail_out)[:] = bn[_np(tail_idx)]


def coarse_back_gather(Wm, x_tail, xidx, z, alpha, oidx, tail_idx, out, accumulate):
    k, s, cwp = Wm.shape
    xn, wi, zn, gi, on = _np(x_tail), _np(xidx).reshape(k, cwp), _np(z), _np(oidx), _np(out)
    for i in range(k):
        v = zn[i * s:(i + 1) * s] + alpha * K.dense_gemv(np.ascontiguousarray(_np(Wm[i])), np.ascontiguousarray(xn[wi[i]]))
        g = gi[i * s:(i + 1) * s]
        ok = g >= 0
        on[g[ok]] = v[ok] + on[g[ok]] if accumulate else v[ok]
    ti = _np(tail_idx)
    on[ti] = xn[:ti.size] + on[ti] if accumulate else xn[:ti.size]


def block_copy(nblocks, bs, src, src_stride, dst, dst_stride):
    sn, dn = _np(src), _np(dst)
    for k in range(nblocks):
        dn[k * dst_stride:k * dst_stride + bs] = sn[k * src_stride:k * src_stride + bs]


class SpGEMMPlan:
    def __init__(self, A, B, record="lazy"):
        self.shape = (A.shape[0], B.shape[1])

    def numeric(self, A, B, out=None):
        C = sp.csr_matrix(_sp(A) @ _sp(B))
        C.sort_indices()
        new = DeviceCSR.from_scipy(C, "cpu")
        if out is not None:
            out.vals.copy_(new.vals)
            return out
        return new


class _Sched:
    def __init__(self, rows):
        self.d_rows = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32))


def build_gs_schedule(A_scipy_csr, kind, device):
    """The CPU stand-in executes a lexicographic schedule as what it is: the rows in ascending order."""
    if kind != "lexicographic":
        raise NotImplementedError(kind)
    return _Sched(np.arange(A_scipy_csr.shape[0], dtype=np.int32))


def csr_gs_schedule(A, x, b, sched, sweeps=1):
    n, rp, ci, va = _raw(A)
    xx = np.ascontiguousarray(_np(x))
    rows = np.sort(_np(sched.d_rows)).astype(np.int32)
    for _ in range(int(sweeps)):
        K.lib().orc_csr_gs_rows(rp, ci, va, xx, np.ascontiguousarray(_np(b)), rows, rows.size)
    _np(x)[:] = xx
