"""ctypes front-end of oracle/lmg_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under learnmultigrid_amd/ does.  See lmg_oracle.c for the
reference file:line each function restates.
"""
import ctypes
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblmg_oracle.so")
_lib = None

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "lmg_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liblmg_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i64, i32, f64 = ctypes.c_int64, ctypes.c_int32, ctypes.c_double
        L.orc_csr_matvec.argtypes = [i64, _i32p, _i32p, _f64p, _f64p, _f64p]
        L.orc_csr_matvec.restype = None
        L.orc_csr_spmv.argtypes = [i64, _i32p, _i32p, _f64p, _f64p, _f64p, f64, f64]
        L.orc_csr_spmv.restype = None
        L.orc_csr_residual.argtypes = [i64, _i32p, _i32p, _f64p, _f64p, _f64p, _f64p]
        L.orc_csr_residual.restype = f64
        L.orc_csr_jacobi.argtypes = [i64, _i32p, _i32p, _f64p, _f64p, _f64p, f64, _f64p]
        L.orc_csr_jacobi.restype = None
        L.orc_csr_gs_forward.argtypes = [i64, _i32p, _i32p, _f64p, _f64p, _f64p, i32]
        L.orc_csr_gs_forward.restype = None
        L.orc_csr_gs_rows.argtypes = [_i32p, _i32p, _f64p, _f64p, _f64p, _i32p, i64]
        L.orc_csr_gs_rows.restype = None
        L.orc_dense_gemv.argtypes = [i64, i64, _f64p, _f64p, _f64p]
        L.orc_dense_gemv.restype = None
        _lib = L
    return _lib


def as_csr(A):
    """Sorted, duplicate-free CSR with int32 indices and float64 values (what
    pyamg hands its C++ kernel after its CSC->CSR conversion, Multigrid.py:88)."""
    A = sp.csr_matrix(A, dtype=np.float64)
    if not A.has_canonical_format:
        A = A.copy()
        A.sum_duplicates()
    if A.indptr.dtype != np.int32 or A.indices.dtype != np.int32:
        A = sp.csr_matrix((A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32)),
                          shape=A.shape)
    return A


def _vec(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))


def matvec(A, x):
    A = as_csr(A)
    y = np.empty(A.shape[0])
    lib().orc_csr_matvec(A.shape[0], A.indptr, A.indices, A.data, _vec(x), y)
    return y


def spmv(A, x, y=None, alpha=1.0, beta=0.0):
    A = as_csr(A)
    out = np.zeros(A.shape[0]) if y is None else _vec(y).copy()
    lib().orc_csr_spmv(A.shape[0], A.indptr, A.indices, A.data, _vec(x), out, alpha, beta)
    return out


def residual(A, x, b):
    """returns (r, sum r_i^2)."""
    A = as_csr(A)
    r = np.empty(A.shape[0])
    n2 = lib().orc_csr_residual(A.shape[0], A.indptr, A.indices, A.data, _vec(x), _vec(b), r)
    return r, n2


def jacobi(A, x, b, omega=1.0):
    A = as_csr(A)
    out = np.empty(A.shape[0])
    lib().orc_csr_jacobi(A.shape[0], A.indptr, A.indices, A.data, _vec(x), _vec(b), omega, out)
    return out


def gs_forward(A, x, b, iterations=1):
    """In place on x (must be a float64 array whose ravel() is a view), like pyamg."""
    A = as_csr(A)
    xr = x.reshape(-1)
    if xr.dtype != np.float64 or not xr.flags.c_contiguous or not np.shares_memory(xr, x):
        raise ValueError("x must be contiguous float64 (updated in place)")
    lib().orc_csr_gs_forward(A.shape[0], A.indptr, A.indices, A.data, xr, _vec(b), int(iterations))
    return x


def gs_rows(A, x, b, rows):
    A = as_csr(A)
    xr = x.reshape(-1)
    if xr.dtype != np.float64 or not xr.flags.c_contiguous or not np.shares_memory(xr, x):
        raise ValueError("x must be contiguous float64 (updated in place)")
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    lib().orc_csr_gs_rows(A.indptr, A.indices, A.data, xr, _vec(b), rows, rows.size)
    return x


def dense_gemv(M, x):
    M = np.ascontiguousarray(M, dtype=np.float64)
    y = np.empty(M.shape[0])
    lib().orc_dense_gemv(M.shape[0], M.shape[1], M, _vec(x), y)
    return y
