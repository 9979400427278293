"""1-D L2-projection transfer operators, vectorised (host side).

Restates learn_multigrid/L2_projection: `L2Projection(type, fine_mesh, coarse_mesh)
.compute_transfer_1d()` (L2Projection.py:26-57) = mesh intersection (Intersection.py:35-80, an
O(ne * ne_c) double loop in the reference) + coupling operator B by 3-point Gauss quadrature on
every intersection segment (CouplingOperator.py:32-69) + mass matrix (MassMatrix.py:61-82), then
    "L2"     Q = M^-1 B                         (L2Projection.py:67-72)
    "pseudo" Q = diag(colsum M)^-1 B            (lumped mass, :74-82)
    "quasi"  Q = B / rowsum(B)                  (:84-90)
Here the segments come from one sorted merge of the two node sets (O(n log n)) and B, M are
assembled with array operations; results agree with the reference's to rounding
(tests/test_l2_projection.py against goldens g2 / g3).  Returned sparse (CSR); the reference
returns the same matrix dense."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# Quadrature(3) of the reference (assembly/Quadrature.py:52-66), 14-digit constants included
_GP = np.array([0.11270166537926, 0.50000000000000, 0.88729833462074])
_GW = np.array([0.27777777777778, 0.44444444444444, 0.27777777777778])


def mass_matrix_1d(x):
    """P1 mass matrix of the 1-D mesh with nodes x (MassMatrix.compute_mass_1d)."""
    x = np.asarray(x, dtype=np.float64)
    h = np.diff(x)
    # loc_m[i, j] = h * sum_k phi_i(p_k) phi_j(p_k) w_k  with phi = (1 - p, p)
    phi = np.stack([1.0 - _GP, _GP])                      # (2, 3)
    ref = np.einsum("ik,jk,k->ij", phi, phi, _GW)         # (2, 2)
    n = x.size
    i = np.arange(n - 1)
    rows = np.concatenate([i, i, i + 1, i + 1])
    cols = np.concatenate([i, i + 1, i, i + 1])
    vals = np.concatenate([h * ref[0, 0], h * ref[0, 1], h * ref[1, 0], h * ref[1, 1]])
    return sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()


def coupling_operator_1d(x_fine, x_coarse):
    """B (n_fine x n_coarse): integral of fine basis x coarse basis over every intersection of a
    fine with a coarse element."""
    xf = np.asarray(x_fine, dtype=np.float64)
    xc = np.asarray(x_coarse, dtype=np.float64)
    union = np.union1d(xf, xc)                            # Intersection.py:57
    xa, xb = union[:-1], union[1:]
    mid = 0.5 * (xa + xb)
    fe = np.clip(np.searchsorted(xf, mid, side="right") - 1, 0, xf.size - 2)    # fine element of each segment
    ce = np.clip(np.searchsorted(xc, mid, side="right") - 1, 0, xc.size - 2)
    phys = xa[:, None] + _GP[None, :] * (xb - xa)[:, None]                      # g_function
    fr = (phys - xf[fe][:, None]) / (xf[fe + 1] - xf[fe])[:, None]              # inv_g_function
    cr = (phys - xc[ce][:, None]) / (xc[ce + 1] - xc[ce])[:, None]
    pf = np.stack([1.0 - fr, fr], axis=1)                 # (nseg, 2, 3)
    pc = np.stack([1.0 - cr, cr], axis=1)
    loc = np.einsum("sik,sjk,k->sij", pf, pc, _GW) * (xb - xa)[:, None, None]
    rows = np.stack([fe, fe, fe + 1, fe + 1], axis=1).ravel()
    cols = np.stack([ce, ce + 1, ce, ce + 1], axis=1).ravel()
    return sp.coo_matrix((loc.reshape(-1, 4).ravel(), (rows, cols)), shape=(xf.size, xc.size)).tocsr()


def transfer_1d(kind, x_fine, x_coarse):
    """Q of L2Projection(kind, fine, coarse).compute_transfer_1d(); kind in L2 | pseudo | quasi."""
    B = coupling_operator_1d(x_fine, x_coarse)
    if kind == "quasi":
        rs = np.asarray(B.sum(axis=1)).ravel()
        return sp.csr_matrix(sp.diags(1.0 / rs) @ B)
    M = mass_matrix_1d(x_fine)
    if kind == "pseudo":
        lumped = np.asarray(M.sum(axis=0)).ravel()
        return sp.csr_matrix(sp.diags(1.0 / lumped) @ B)
    if kind == "L2":
        return sp.csr_matrix(spla.spsolve(sp.csc_matrix(M), sp.csc_matrix(B)))
    raise ValueError("Invalid order %r (L2 | pseudo | quasi)" % (kind,))


def transfer_1d_device(kind, x_fine, x_coarse, device):
    """The same Q on an MI355X (csrc/l2proj.hip: one thread per fine node, segments by binary search,
    rows in CSR): kind in B | pseudo | quasi ("L2" is a dense M^-1 B: use transfer_1d).  Returns an
    ops.DeviceCSR; nodes must be strictly increasing."""
    import torch
    from . import ops, _lib
    code = {"B": 0, "pseudo": 1, "quasi": 2}.get(kind)
    if code is None:
        raise ValueError("device transfer: kind must be B | pseudo | quasi, got %r" % (kind,))
    xf = np.ascontiguousarray(np.asarray(x_fine, dtype=np.float64).ravel())
    xc = np.ascontiguousarray(np.asarray(x_coarse, dtype=np.float64).ravel())
    if xf.size < 2 or xc.size < 2 or np.any(np.diff(xf) <= 0) or np.any(np.diff(xc) <= 0):
        raise ValueError("node coordinates must be strictly increasing (at least two nodes per mesh)")
    dxf, dxc = torch.from_numpy(xf).to(device), torch.from_numpy(xc).to(device)
    nf, nc = xf.size, xc.size
    L = _lib.lib()
    s = torch.cuda.current_stream(dxf.device).cuda_stream
    rownnz = torch.empty(nf, dtype=torch.int32, device=device)
    _lib.check(L.lmg_l2_coupling_count(nf, nc, dxf.data_ptr(), dxc.data_ptr(), rownnz.data_ptr(), s), "lmg_l2_coupling_count")
    rowptr = torch.empty(nf + 1, dtype=torch.int32, device=device)
    ops.exclusive_scan_i32(rownnz, rowptr)
    nnz = int(rowptr[-1])
    colidx = torch.empty(nnz, dtype=torch.int32, device=device)
    vals = torch.empty(nnz, dtype=torch.float64, device=device)
    _lib.check(L.lmg_l2_coupling_fill(code, nf, nc, dxf.data_ptr(), dxc.data_ptr(), rowptr.data_ptr(), colidx.data_ptr(),
                                      vals.data_ptr(), s), "lmg_l2_coupling_fill")
    return ops.DeviceCSR(rowptr, colidx, vals, (nf, nc))
