"""P1 finite-element assembly on the device -- what
learn_multigrid/assembly/{StiffnessMatrix,MassMatrix,LoadVector}.py do with per-element Python
loops, as one atomic-free HIP kernel (csrc/assembly.hip).  The mesh graph (CSR pattern and the
node->element adjacency) is integer work done once per mesh on the host; values are assembled
on the GPU and can be re-assembled for new coordinates or coefficients without touching the
pattern."""
import numpy as np
import scipy.sparse as sp
import torch

from . import _lib
from ._lib import check
from .ops import DeviceCSR, F64, I32, _p, _s


class P1Mesh2D:
    """Triangle mesh (p: (n,2) coordinates, conn: (ne,3) 0-based vertex ids, like Mesh2D.p /
    Mesh2D.conn of the reference) prepared for device assembly."""

    def __init__(self, p, conn, device):
        p = np.asarray(p, dtype=np.float64)
        conn = np.ascontiguousarray(conn, dtype=np.int32)
        self.n, self.ne = p.shape[0], conn.shape[0]
        self.device = torch.device(device)
        # node -> (element, local vertex), elements ascending (= the reference's assembly order)
        flat = conn.ravel()
        order = np.argsort(flat, kind="stable")
        counts = np.bincount(flat, minlength=self.n)
        ptr = np.zeros(self.n + 1, dtype=np.int32)
        np.cumsum(counts, out=ptr[1:])
        # pattern: node i is coupled to every vertex of its elements
        rows = np.repeat(conn, 3, axis=1).ravel()
        cols = np.tile(conn, (1, 3)).ravel()
        pat = sp.coo_matrix((np.ones(rows.size, dtype=np.int8), (rows, cols)), shape=(self.n, self.n)).tocsr()
        pat.sum_duplicates()
        pat.sort_indices()
        d = self.device
        self.px = torch.from_numpy(np.ascontiguousarray(p[:, 0])).to(d)
        self.py = torch.from_numpy(np.ascontiguousarray(p[:, 1])).to(d)
        self.conn = torch.from_numpy(conn.ravel()).to(d)
        self.n2e_ptr = torch.from_numpy(ptr).to(d)
        self.n2e_elem = torch.from_numpy((order // 3).astype(np.int32)).to(d)
        self.n2e_loc = torch.from_numpy((order % 3).astype(np.int32)).to(d)
        self.rowptr = torch.from_numpy(pat.indptr.astype(np.int32)).to(d)
        self.colidx = torch.from_numpy(pat.indices.astype(np.int32)).to(d)
        self.nnz = int(pat.nnz)

    def assemble(self, stiffness=True, mass=True, load=None, coeff=None):
        """Returns (A, M, rhs): DeviceCSR / DeviceCSR / (n,) tensor, None for parts not asked for.
        load: constant right-hand-side function value f (the scripts use f = -1)."""
        d = self.device
        a = torch.empty(self.nnz, dtype=F64, device=d) if stiffness else None
        m = torch.empty(self.nnz, dtype=F64, device=d) if mass else None
        r = torch.empty(self.n, dtype=F64, device=d) if load is not None else None
        c = None
        if coeff is not None:
            c = torch.as_tensor(np.asarray(coeff, dtype=np.float64)).to(d) if not torch.is_tensor(coeff) else coeff
            if c.numel() != self.ne:
                raise ValueError("one coefficient per element expected")
        check(_lib.lib().lmg_p1_assemble_2d(self.n, _p(self.px), _p(self.py), _p(self.conn), _p(self.n2e_ptr),
                                            _p(self.n2e_elem), _p(self.n2e_loc), _p(c),
                                            float(load if load is not None else 0.0), _p(self.rowptr),
                                            _p(self.colidx), _p(a), _p(m), _p(r), _s()), "lmg_p1_assemble_2d")
        A = DeviceCSR(self.rowptr, self.colidx, a, (self.n, self.n)) if stiffness else None
        M = DeviceCSR(self.rowptr, self.colidx, m, (self.n, self.n)) if mass else None
        return A, M, r
