"""Coarsest-level direct solve on the device (replaces `spsolve(A_coarse, res_coarse)`,
learn_multigrid/solvers/Multigrid.py:106, which re-factorises with SuperLU in every cycle).

Setup happens once; every application is a handful of HBM-bound GEMV / SpMV launches
(graph-capturable, no host round trip).  Two strategies:

* `DenseInverse` -- A^-1 as one dense fp64 matrix applied by lmg_dense_gemv: 8 n^2 bytes per
  application (129^2 unknowns: 2.2 GB, ~0.55 ms on MI355X).  Used for small or non-banded
  operators.

* `BandedBlockSolver` -- for operators with half-bandwidth w << n in their natural ordering
  (every Galerkin operator of a row-major grid: w = side + 1).  The unknowns are cut into k
  strips separated by k-1 separator blocks of >= w unknowns, so strips only couple through
  separators (one-level nested dissection in the given ordering, EXACT block elimination):

        [ A_II  A_IS ] [x_I]   [b_I]        A_II = blockdiag(strip_0 .. strip_{k-1})
        [ A_SI  A_SS ] [x_S] = [b_S]        S    = A_SS - A_SI A_II^-1 A_IS

        y_I = A_II^-1 b_I ;  x_S = S^-1 (b_S - A_SI y_I) ;  x_I = A_II^-1 (b_I - A_IS x_S)

  Dense storage: k strip inverses (s x s) + S^-1, i.e. about n^2/k + ((k-1)w)^2 doubles
  instead of n^2: 129^2 unknowns, k = 16 -> 2 x 108 MB + 30 MB per application instead of
  2.2 GB.  Pivoting happens inside the strips only (fine for the M-matrix-like Galerkin
  operators); `refine` steps of iterative refinement against the true sparse operator are
  applied by the caller and make the result as accurate as a pivoted factorisation.
"""
import numpy as np
import scipy.sparse as sp
import torch

from .ops import DeviceCSR, F64

MAX_DENSE = 46000          # 46000^2 * 8 B = 17 GB of the 288 GB HBM
_INV_LEAF = 512


def csr_to_dense(A):
    n, m = A.shape
    dense = torch.zeros((n, m), dtype=F64, device=A.device)
    if A.vals.is_cuda:
        from . import ops as _ops
        _ops.csr_to_dense(A, dense)
        return dense
    rows = torch.repeat_interleave(torch.arange(n, device=A.device), (A.rowptr[1:] - A.rowptr[:-1]).long())
    dense.index_put_((rows, A.colidx.long()), A.vals, accumulate=True)      # (duplicates are summed: non-canonical input)
    return dense


GRID_G_FORCE = None           # grid-block solver: blocks per line instead of the byte-minimal choice (A/B: tools/scan_coarse.py)
FOLD_PERMUTATION = True       # banded solver: gather / scatter folded into the first / last product (A/B switch)
BACKSUB_ONE_LAUNCH = True     # banded solver: x_I = y_I - (A_II^-1 A_IS) x_S instead of A_II^-1 (b_I - A_IS x_S)


def _inv_schur(A, leaf=_INV_LEAF):
    """Recursive 2x2 Schur-complement inversion of A (..., n, n): only small (batched) leaf
    inversions and plain GEMMs.  No pivoting across blocks; the callers verify the result (dense_inverse)."""
    if A.is_cuda:
        A3 = A if A.dim() == 3 else A.unsqueeze(0)
        return _inv_schur_dev(A3.contiguous(), 128).reshape(A.shape)
    n = A.shape[-1]
    if n <= leaf:
        return torch.linalg.inv(A)
    h = n // 2
    A11, A12, A21, A22 = A[..., :h, :h], A[..., :h, h:], A[..., h:, :h], A[..., h:, h:]
    I11 = _inv_schur(A11.contiguous(), leaf)
    T = I11 @ A12
    IS = _inv_schur((A22 - A21 @ T).contiguous(), leaf)
    W = IS @ (A21 @ I11)
    out = torch.empty_like(A)
    out[..., h:, h:] = IS
    out[..., h:, :h] = -W
    out[..., :h, h:] = -(T @ IS)
    out[..., :h, :h] = I11 + T @ W
    return out


def _inv_schur_dev(A, leaf):
    """The same recursion on a device batch (b, n, n) with the library's own kernels only: batched Gauss-Jordan leaves
    (lmg_batched_inverse), lmg_batched_gemm / lmg_copy2d on strided sub-block views -- no rocBLAS, no rocSOLVER, no library
    elementwise kernels (each costs 50 - 200 ms of code-object loading the first time a process uses it)."""
    from . import ops as _ops
    b, n, _ = A.shape
    dev = A.device
    if n <= leaf:
        M = _ops.batched_inverse(A)
        if M is None:
            raise RuntimeError("singular leaf")
        return M
    h = n // 2
    new = lambda r, c: torch.empty((b, r, c), dtype=F64, device=dev)
    A11, A12, A21, A22 = A[:, :h, :h], A[:, :h, h:], A[:, h:, :h], A[:, h:, h:]
    I11 = _inv_schur_dev(_ops.copy2d(A11, new(h, h)), leaf)
    T = _ops.gemm(I11, A12, new(h, n - h))
    S = _ops.copy2d(A22, new(n - h, n - h))
    _ops.gemm(A21, T, S, alpha=-1.0, beta=1.0)                        # S = A22 - A21 A11^-1 A12
    IS = _inv_schur_dev(S, leaf)
    out = new(n, n)
    _ops.copy2d(IS, out[:, h:, h:])
    V = _ops.gemm(A21, I11, new(n - h, h))
    _ops.gemm(IS, V, out[:, h:, :h], alpha=-1.0)                      # -W,  W = S^-1 A21 A11^-1
    _ops.gemm(T, IS, out[:, :h, h:], alpha=-1.0)
    _ops.copy2d(I11, out[:, :h, :h])
    _ops.gemm(T, out[:, h:, :h], out[:, :h, :h], alpha=-1.0, beta=1.0)    # A11^-1 + T W
    return out


def _defect_dev(dense, M):
    """|| I - A M ||_F of a device batch (>= the largest entry of the defect), own kernels only."""
    from . import ops as _ops
    d3 = dense if dense.dim() == 3 else dense.unsqueeze(0)
    m3 = M if M.dim() == 3 else M.unsqueeze(0)
    b, n, _ = d3.shape
    D = torch.zeros((b, n, n), dtype=F64, device=dense.device)
    ones = torch.ones((b, n, 1), dtype=F64, device=dense.device)
    _ops.copy2d(ones, torch.as_strided(D, (b, n, 1), (n * n, n + 1, 1)))       # the diagonal as a strided column
    _ops.gemm(d3, m3, D, alpha=-1.0, beta=1.0)
    flat = D.reshape(-1)
    part = torch.empty(_ops.partials_count(flat.numel()), dtype=F64, device=dense.device)
    out = torch.zeros(1, dtype=F64, device=dense.device)
    _ops.dot(flat, flat, part, out)
    v = float(out.item())
    return float("inf") if v != v else v ** 0.5


def dense_inverse(dense, polish=2, tol=1e-9):
    """A^-1 on the device (SETUP phase) for one matrix (n, n) or a batch (k, n, n).
    1. block Schur recursion on GEMMs with small pivoted leaf inversions (own kernels on the device: see
       _inv_schur_dev);
    2. if the defect I - A M is not below tol (a pivot the recursion could not see): torch.linalg.inv, then
       `polish` Newton-Schulz steps M <- M (2I - A M);
    3. as a last resort a pure Newton-Schulz iteration from A^T/(|A|_1 |A|_inf), which converges
       for every nonsingular A.
    Raises if the operator is numerically singular."""
    n = dense.shape[-1]
    eye = None

    def defect(M):
        nonlocal eye
        if dense.is_cuda:
            return _defect_dev(dense, M)
        if eye is None:
            eye = torch.eye(n, dtype=F64, device=dense.device)
        return float((eye - dense @ M).abs().max())

    def good(M):
        return M is not None and defect(M) < tol            # (a NaN / inf anywhere in M makes the defect NaN / inf: not < tol)

    M = None
    try:
        M = _inv_schur(dense, 128 if (dense.dim() == 3 or dense.is_cuda) else _INV_LEAF)
    except RuntimeError:                              # a singular leaf
        M = None
    if good(M):
        return M.contiguous()
    eye = torch.eye(n, dtype=F64, device=dense.device)
    try:
        M = torch.linalg.inv(dense)
    except RuntimeError:
        M = None
    if M is not None and bool(torch.isfinite(M).all()):
        for _ in range(polish):
            if float((eye - dense @ M).abs().max()) < 1e-13:
                break
            M = M @ (2.0 * eye - dense @ M)
        if good(M):
            return M.contiguous()
    nrm = dense.abs().sum(-2).max() * dense.abs().sum(-1).max()
    M = dense.transpose(-1, -2).contiguous() / nrm
    for _ in range(200):
        M = M @ (2.0 * eye - dense @ M)
        if float((eye - dense @ M).abs().max()) < 1e-12:
            break
    if not good(M):
        raise ValueError("coarsest operator is numerically singular (cannot be inverted)")
    return M.contiguous()


class DenseInverse:
    kind = "dense"

    def __init__(self, A, ops_mod):
        self.ops = ops_mod
        n = A.shape[0]
        if n > MAX_DENSE:
            raise ValueError("coarsest level has %d unknowns (> %d): use more levels" % (n, MAX_DENSE))
        self.n = n
        self.factor(A)

    def factor(self, A):
        if A.shape[0] != self.n:
            raise ValueError("coarse operator changed its size")
        self.inv = dense_inverse(csr_to_dense(A))

    def apply(self, b, x):
        self.ops.dense_gemv(self.inv, b, x)

    def bytes_per_apply(self):
        return 8 * self.n * self.n


def half_bandwidth(A_host):
    coo = A_host.tocoo()
    return int(np.abs(coo.row - coo.col).max()) if coo.nnz else 0


class BandedBlockSolver:
    kind = "banded-block"

    @staticmethod
    def plan(n, w):
        """(k strips, strip size s) minimising the dense bytes per application
        k s^2 + k s c_w + n_S^2 (strip inverses, the strips' A_II^-1 A_IS on their separator windows of c_w = two
        separators, Schur-complement inverse) with equal even strip sizes, or None if banding does not pay."""
        best = None
        for k in range(2, 129):
            s = (n - (k - 1) * w) // k
            s -= s % 2
            if s < 2 * w or s < 32:
                break
            ns = n - k * s
            cost = k * s * s + k * s * 2 * (-(-ns // (k - 1))) + ns * ns
            if best is None or cost < best[0]:
                best = (cost, k, s)
        if best is None or best[0] > 0.5 * n * n:
            return None
        return best[1], best[2]

    def __init__(self, A, ops_mod, k, s):
        self.ops = ops_mod
        self._symbolic(A, k, s)
        self.factor(A)

    # ---- symbolic: index sets and entry maps, once per sparsity pattern (host) ----------
    def _symbolic(self, A, k, s):
        dev = A.device
        n = A.shape[0]
        rp, ci = A.rowptr.cpu().numpy(), A.colidx.cpu().numpy()
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
        cols = ci.astype(np.int64)
        w = int(np.abs(rows - cols).max()) if cols.size else 0
        ns = n - k * s                                # all separator unknowns
        base, extra = divmod(ns, k - 1)               # separator sizes, each >= w
        if base < w:
            raise ValueError("separators thinner than the bandwidth")
        strips, seps, sep_off, pos = [], [], [0], 0
        for i in range(k):
            strips.append(np.arange(pos, pos + s))
            pos += s
            if i < k - 1:
                sz = base + (1 if i < extra else 0)
                seps.append(np.arange(pos, pos + sz))
                sep_off.append(sep_off[-1] + sz)
                pos += sz
        assert pos == n
        I = np.concatenate(strips)
        S = np.concatenate(seps)
        nI, nS = I.size, S.size
        self.n, self.nI, self.nS, self.k, self.s, self.w = n, nI, nS, k, s, w
        perm = np.concatenate([I, S])
        inv = np.empty(n, dtype=np.int64)
        inv[perm] = np.arange(n)
        pr, pc = inv[rows], inv[cols]                 # permuted coordinates of every entry
        eid = np.arange(cols.size, dtype=np.int64)
        rI, cI = pr < nI, pc < nI
        # strips must be mutually decoupled: A_II is block diagonal with s x s blocks
        m = rI & cI
        if np.any(pr[m] // s != pc[m] // s):
            raise ValueError("strips are coupled: the operator is not banded in this ordering")
        # window of separator unknowns a strip couples to: separators i-1 and i (contiguous in S)
        sep_off = np.asarray(sep_off, dtype=np.int64)
        ws = np.array([sep_off[max(i - 1, 0)] for i in range(k)], dtype=np.int64)
        we = np.array([sep_off[min(i + 1, k - 1)] for i in range(k)], dtype=np.int64)
        cw = int((we - ws).max())
        self.cw = cw
        t = lambda a, dt=torch.int64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
        if max(k * s * s, k * s * cw, nS * nS, cols.size) >= 2 ** 31:
            raise ValueError("banded coarse solver: dense factors beyond int32 indexing")
        t32 = lambda a: t(a, torch.int32)
        self._src_II = t32(eid[m])
        self._dst_II = t32((pr[m] // s) * s * s + (pr[m] % s) * s + (pc[m] % s))
        m = rI & ~cI                                  # A_IS: strip rows x separator columns
        strip = pr[m] // s
        loc = (pc[m] - nI) - ws[strip]
        if np.any(loc < 0) or np.any(loc >= (we - ws)[strip]):
            raise ValueError("a strip couples to a distant separator: not banded in this ordering")
        self._src_IS = t32(eid[m])
        self._dst_IS = t32(strip * s * cw + (pr[m] % s) * cw + loc)
        IS_pat = sp.csr_matrix((eid[m] + 1.0, (pr[m], pc[m] - nI)), shape=(nI, nS))
        m = ~rI & cI                                  # A_SI: separator rows x strip columns
        strip = pc[m] // s
        loc = (pr[m] - nI) - ws[strip]
        if np.any(loc < 0) or np.any(loc >= (we - ws)[strip]):
            raise ValueError("a strip couples to a distant separator: not banded in this ordering")
        self._src_SI = t32(eid[m])
        self._dst_SI = t32(strip * cw * s + loc * s + (pc[m] % s))
        SI_pat = sp.csr_matrix((eid[m] + 1.0, (pr[m] - nI, pc[m])), shape=(nS, nI))
        m = ~rI & ~cI
        self._src_SS = t32(eid[m])
        self._dst_SS = t32((pr[m] - nI) * nS + (pc[m] - nI))
        # where the k local Schur updates land in S (flat indices, padded entries masked out)
        ar = np.arange(cw, dtype=np.int64)
        gi = ws[:, None, None] + ar[None, :, None]
        gj = ws[:, None, None] + ar[None, None, :]
        ok = (gi < we[:, None, None]) & (gj < we[:, None, None])
        flat = np.arange(k * cw * cw, dtype=np.int64).reshape(k, cw, cw)
        dst = gi * nS + gj
        # strips i and i+1 both update separator i: even strips first, then odd ones, so every
        # pass writes each entry of S at most once (fixed order of additions => reproducible bits)
        self._upd = []
        for parity in (0, 1):
            sel = ok.copy()
            sel[np.arange(k) % 2 != parity] = False
            self._upd.append((t32(flat[sel]), t32(dst[sel])))
        # sparse off-diagonal blocks for apply(): CSR patterns once, values refreshed by factor()
        self._IS_src = t32(np.rint(IS_pat.data).astype(np.int64) - 1)
        self._SI_src = t32(np.rint(SI_pat.data).astype(np.int64) - 1)
        z64 = lambda m_: torch.zeros(m_, dtype=F64, device=dev)
        self.A_IS = DeviceCSR(t(IS_pat.indptr, torch.int32), t(IS_pat.indices, torch.int32), z64(IS_pat.nnz), (nI, nS))
        self.A_SI = DeviceCSR(t(SI_pat.indptr, torch.int32), t(SI_pat.indices, torch.int32), z64(SI_pat.nnz), (nS, nI))
        self.perm = t(perm, torch.int32)
        # back-substitution in one launch: x_I = y_I - W x_S with W = A_II^-1 A_IS per strip (s x cw, cw padded to even;
        # window starts ws[i] in x_S; x_S is followed by zeros so that the last window stays inside)
        self.cwp = cw + (cw & 1)
        self._ws = t32(ws)
        self.bp, self.xp, self.y, self.t = z64(n), z64(n + self.cwp + 2), z64(nI), z64(nI)
        self._acc = z64(n)
        self.W = None
        self.blocks = torch.zeros((k, s, s), dtype=F64, device=dev)
        self.Sinv = None
        self._nnz = int(cols.size)

    # ---- numeric: device only (repeated for every Galerkin rebuild) -----------------------
    def _place(self, v, src, dst, size):
        """zeros(size) with out[dst[k]] = v[src[k]] (destinations are distinct: sorted, duplicate-free CSR) --
        own gather / scatter kernels: torch's index_put_ costs 0.4 s to load in a fresh process."""
        out = torch.zeros(size, dtype=F64, device=v.device)
        if src.numel():
            tmp = torch.empty(src.numel(), dtype=F64, device=v.device)
            self.ops.gather(src, v, tmp)
            self.ops.scatter(dst, tmp, out)
        return out

    def factor(self, A):
        """Strip inverses, Schur complement S = A_SS - A_SI A_II^-1 A_IS and S^-1 from the
        current values of A (same pattern as at construction)."""
        if A.nnz != self._nnz:
            raise ValueError("coarse operator changed its sparsity pattern")
        k, s, cw, nS = self.k, self.s, self.cw, self.nS
        dev = A.vals.device
        v = A.vals
        dense = self._place(v, self._src_II, self._dst_II, k * s * s).view(k, s, s)
        self.blocks = dense_inverse(dense)            # all strips as one batch
        ais = self._place(v, self._src_IS, self._dst_IS, k * s * cw)
        asi = self._place(v, self._src_SI, self._dst_SI, k * cw * s)
        w_ = torch.bmm(self.blocks, ais.view(k, s, cw))                                      # k x s x cw: A_II^-1 A_IS
        upd = torch.bmm(asi.view(k, cw, s), w_)                                              # k x cw x cw
        if BACKSUB_ONE_LAUNCH and hasattr(self.ops, "dense_gemv_windows_off"):
            self.W = torch.zeros((k, s, self.cwp), dtype=F64, device=dev)
            self.W[:, :, :cw] = w_
        else:
            self.W = None
        del w_
        Sc = self._place(v, self._src_SS, self._dst_SS, nS * nS)
        upd = upd.reshape(-1)
        for sel, dst in self._upd:                       # S[dst] -= upd[sel], every destination once per pass
            if sel.numel():
                tmp = torch.empty(sel.numel(), dtype=F64, device=dev)
                cur = torch.empty(sel.numel(), dtype=F64, device=dev)
                self.ops.gather(sel, upd, tmp)
                self.ops.gather(dst, Sc, cur)
                self.ops.axpby(-1.0, tmp, 1.0, cur)
                self.ops.scatter(dst, cur, Sc)
        self.Sinv = dense_inverse(Sc.view(nS, nS))
        self.ops.gather(self._IS_src, v, self.A_IS.vals)
        self.ops.gather(self._SI_src, v, self.A_SI.vals)
        self.A_IS.invalidate_packed()
        self.A_SI.invalidate_packed()

    supports_accumulate = True

    def apply(self, b, x, accumulate=False):
        """x = A^-1 b; accumulate: x += A^-1 b (the refinement step of Hierarchy.coarse_solve without a temporary)."""
        o = self.ops
        nI = self.nI
        folded = self.W is not None and BACKSUB_ONE_LAUNCH and FOLD_PERMUTATION and hasattr(o, "coarse_front")
        bI, bS = self.bp[:nI], self.bp[nI:]
        xI, xS = self.xp[:nI], self.xp[nI:nI + self.nS]
        if folded:
            # the permutation [strips | separators] is folded into the first and the last product: 4 launches
            o.coarse_front(self.blocks, b, self.perm, self.y, bS)     # y_I = A_II^-1 b_I, b_S gathered on the side
            o.csr_spmv(self.A_SI, self.y, bS, -1.0, 1.0)              # g_S = b_S - A_SI y_I   (in place)
            o.dense_gemv(self.Sinv, bS, xS)                           # x_S = S^-1 g_S
            o.coarse_back(self.W, self.xp[nI:], self._ws, self.y, -1.0, self.perm, x, accumulate, self.nS)
            return
        if accumulate:
            self.apply(b, self._acc)
            o.axpby(1.0, self._acc, 1.0, x)
            return
        o.gather(self.perm, b, self.bp)                           # permuted rhs [b_I | b_S]
        o.dense_gemv_blockdiag(self.blocks, bI, self.y)           # y_I = A_II^-1 b_I
        o.csr_spmv(self.A_SI, self.y, bS, -1.0, 1.0)              # g_S = b_S - A_SI y_I   (in place)
        o.dense_gemv(self.Sinv, bS, xS)                           # x_S = S^-1 g_S
        if self.W is not None and BACKSUB_ONE_LAUNCH:
            # x_I = y_I - (A_II^-1 A_IS) x_S: one launch over 8 k s cw bytes instead of an SpMV and 8 k s^2
            o.dense_gemv_windows_off(self.W, self.xp[nI:], self._ws, xI, self.s, z=self.y, z_stride=self.s, alpha=-1.0)
        else:
            o.csr_residual_norm2(self.A_IS, xS, bI, self.t, None, None)      # t_I = b_I - A_IS x_S (one launch)
            o.dense_gemv_blockdiag(self.blocks, self.t, xI)           # x_I = A_II^-1 t_I
        o.scatter(self.perm, self.xp[:self.n], x)

    def bytes_per_apply(self):
        back = self.k * self.s * self.cwp if self.W is not None else self.k * self.s * self.s
        return 8 * (self.k * self.s * self.s + back + self.nS * self.nS)



class GridBlockSolver:
    """One-level nested dissection in TWO directions for operators of a row-major grid (n = lines x W unknowns,
    every coupling within r lines and r columns: the 9-point Galerkin operators of the geometric transfers, r = 1;
    the 25-point ones of L2-type / learned transfers, r = 2).  Separator lines AND columns, r unknowns thick, cut
    the grid into Gy x Gx rectangular blocks that only couple through separators; then exactly the block
    elimination of BandedBlockSolver,

        y_I = A_II^-1 b_I ;  x_S = S^-1 (b_S - A_SI y_I) ;  x_I = y_I - (A_II^-1 A_IS) x_S,

    with dense block inverses (padded to one size s), dense W_k = A_II^-1 A_IS on the cw separator unknowns block k
    touches and a dense S^-1.  Cutting in both directions is what makes the factors small: strips of whole grid lines
    need k s^2 with s ~ 5 W, squares (W / G)^2 -- 129^2 unknowns: 63 MB per application (G = 10: 100 blocks of <= 144,
    2 241 separator unknowns) against 159 MB for 21 strips.  Four launches per application, the permutation folded
    into the first and the last one (lmg_coarse_front_gather / lmg_coarse_back_gather)."""
    kind = "grid-block"
    supports_accumulate = True

    @staticmethod
    def detect_grid(n, A_host):
        """(W, lines, r) when every entry (i, j) of the pattern has |line(i) - line(j)| <= r and |col(i) - col(j)| <= r
        for a line stride W with n = lines * W and a small r, else None."""
        coo = A_host.tocoo()
        if coo.nnz == 0:
            return None
        d = coo.col.astype(np.int64) - coo.row.astype(np.int64)
        w = int(np.abs(d).max())
        for r in (1, 2, 3):
            for dd in range(r, -1, -1):
                if (w - dd) % r:
                    continue
                W = (w - dd) // r
                if W < 8 or n % W or n // W < 8 or r * 8 > W:
                    continue
                ri, rj = coo.row // W, coo.col // W
                ci, cj = coo.row % W, coo.col % W
                if np.all(np.abs(ri - rj) <= r) and np.all(np.abs(ci - cj) <= r):
                    return int(W), int(n // W), r
        return None

    @staticmethod
    def _cuts(m, G, r):
        """Block coordinate of each of m lines (-1: separator line) for G blocks separated by G - 1 separators of r lines."""
        lab = np.full(m, -1, dtype=np.int64)
        free = m - (G - 1) * r
        if G < 1 or free < G:
            return None
        base, extra = divmod(free, G)
        pos = 0
        for g in range(G):
            sz = base + (1 if g < extra else 0)
            lab[pos:pos + sz] = g
            pos += sz + r
        return lab

    @classmethod
    def plan(cls, W, lines, r):
        """(Gy, Gx, dense bytes per application) minimising k s^2 + k s cw + nS^2, or None."""
        best = None
        for G in (range(2, 65) if GRID_G_FORCE is None else (int(GRID_G_FORCE),)):
            Gx = G
            Gy = max(2, int(round(G * lines / W)))
            lx, ly = cls._cuts(W, Gx, r), cls._cuts(lines, Gy, r)
            if lx is None or ly is None:
                break
            bx, by = -(-(W - (Gx - 1) * r) // Gx), -(-(lines - (Gy - 1) * r) // Gy)
            if bx < 2 * r or by < 2 * r:
                break
            s = bx * by
            s += s % 2
            k = Gx * Gy
            nS = W * lines - int((lx >= 0).sum()) * int((ly >= 0).sum())
            cw = 2 * r * (bx + by) + 4 * r * r
            cost = 8 * (k * s * s + k * s * cw + nS * nS)
            if best is None or cost < best[2]:
                best = (Gy, Gx, cost)
        return best

    def __init__(self, A, ops_mod, W, lines, r, Gy, Gx):
        self.ops = ops_mod
        self._symbolic(A, W, lines, r, Gy, Gx)
        self.factor(A)

    def _symbolic(self, A, W, lines, r, Gy, Gx):
        dev = A.device
        n = A.shape[0]
        rp, ci = A.rowptr.cpu().numpy(), A.colidx.cpu().numpy()
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
        cols = ci.astype(np.int64)
        ly, lx = self._cuts(lines, Gy, r), self._cuts(W, Gx, r)
        if ly is None or lx is None:
            raise ValueError("grid too small for this many blocks")
        by, bx = ly[np.arange(n) // W], lx[np.arange(n) % W]
        interior = (by >= 0) & (bx >= 0)
        blk = np.where(interior, by * Gx + bx, -1)
        k = Gy * Gx
        I = np.flatnonzero(interior)
        order = I[np.argsort(blk[I], kind="stable")]                 # block-major, natural order inside a block
        sizes = np.bincount(blk[order], minlength=k)
        if sizes.min() < 1:
            raise ValueError("empty block")
        s = int(sizes.max())
        s += s % 2
        start = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        pos = np.arange(order.size) - start[blk[order]]              # position inside the block
        slot = np.full(n, -1, dtype=np.int64)                        # padded interior slot of every unknown
        slot[order] = blk[order] * s + pos
        S = np.flatnonzero(~interior)
        nS = S.size
        sidx = np.full(n, -1, dtype=np.int64)
        sidx[S] = np.arange(nS)
        self.n, self.nS, self.k, self.s = n, int(nS), int(k), s
        self.W_, self.lines, self.r, self.Gy, self.Gx = W, lines, r, Gy, Gx
        eid = np.arange(cols.size, dtype=np.int64)
        rI, cI = interior[rows], interior[cols]
        m = rI & cI
        if np.any(blk[rows[m]] != blk[cols[m]]):
            raise ValueError("blocks are coupled: the operator is not a grid operator of this radius")
        if max(k * s * s, nS * nS, cols.size) >= 2 ** 31:
            raise ValueError("grid coarse solver: dense factors beyond int32 indexing")
        t = lambda a, dt=torch.int64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
        t32 = lambda a: t(a, torch.int32)
        self._src_II = t32(eid[m])
        self._dst_II = t32(blk[rows[m]] * s * s + (slot[rows[m]] % s) * s + (slot[cols[m]] % s))
        # padding rows of the blocks are identity rows
        pad_blk = np.repeat(np.arange(k), s - sizes)
        pad_pos = np.concatenate([np.arange(sz, s) for sz in sizes]) if (s - sizes).sum() else np.zeros(0, dtype=np.int64)
        self._pad_dst = t32(pad_blk * s * s + pad_pos * s + pad_pos)
        # separator unknowns every block touches (through A_IS or A_SI): sorted lists, padded to cwp with nS (a zero slot)
        mIS, mSI = rI & ~cI, ~rI & cI
        pair_b = np.concatenate([blk[rows[mIS]], blk[cols[mSI]]])
        pair_s = np.concatenate([sidx[cols[mIS]], sidx[rows[mSI]]])
        key = np.unique(pair_b * nS + pair_s)
        kb, ks = key // nS, key % nS
        cnt = np.bincount(kb, minlength=k)
        cw = int(cnt.max()) if key.size else 0
        if cw == 0:
            raise ValueError("blocks without separators")
        cwp = cw + (cw & 1)
        wstart = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        wloc = np.arange(key.size) - wstart[kb]
        widx = np.full((k, cwp), nS, dtype=np.int64)
        widx[kb, wloc] = ks
        self.cw, self.cwp = cw, cwp
        if k * s * cwp >= 2 ** 31:
            raise ValueError("grid coarse solver: dense factors beyond int32 indexing")

        def loc_of(b_, s_):                                          # position of separator s_ in block b_'s list
            return wloc[np.searchsorted(key, b_ * nS + s_)]
        self._src_IS = t32(eid[mIS])
        self._dst_IS = t32(blk[rows[mIS]] * s * cwp + (slot[rows[mIS]] % s) * cwp + loc_of(blk[rows[mIS]], sidx[cols[mIS]]))
        self._src_SI = t32(eid[mSI])
        self._dst_SI = t32(blk[cols[mSI]] * cwp * s + loc_of(blk[cols[mSI]], sidx[rows[mSI]]) * s + (slot[cols[mSI]] % s))
        SI_pat = sp.csr_matrix((eid[mSI] + 1.0, (sidx[rows[mSI]], slot[cols[mSI]])), shape=(nS, k * s))
        m = ~rI & ~cI
        self._src_SS = t32(eid[m])
        self._dst_SS = t32(sidx[rows[m]] * nS + sidx[cols[m]])
        # where the k local Schur updates (cwp x cwp each) land in S: four passes by the parity of the block
        # coordinates -- blocks of one parity class share no separator unknown, so a pass writes every entry of S at
        # most once (fixed order of additions => reproducible bits)
        valid = widx < nS
        flat = np.arange(k * cwp * cwp, dtype=np.int64).reshape(k, cwp, cwp)
        dst = widx[:, :, None] * nS + widx[:, None, :]
        ok = valid[:, :, None] & valid[:, None, :]
        color = ((np.arange(k) // Gx) % 2) * 2 + (np.arange(k) % Gx) % 2
        self._upd = []
        for c in range(4):
            sel = ok & (color == c)[:, None, None]
            if sel.any():
                d_ = dst[sel]
                if np.unique(d_).size != d_.size:
                    raise ValueError("blocks of one parity class share separator unknowns")
                self._upd.append((t32(flat[sel]), t32(d_)))
        self._SI_src = t32(np.rint(SI_pat.data).astype(np.int64) - 1)
        z64 = lambda m_: torch.zeros(m_, dtype=F64, device=dev)
        self.A_SI = DeviceCSR(t(SI_pat.indptr, torch.int32), t(SI_pat.indices, torch.int32), z64(SI_pat.nnz), (nS, k * s))
        gI = np.full(k * s, -1, dtype=np.int64)
        gI[slot[order]] = order
        self._gI = t32(gI)                                           # padded interior slot -> unknown (-1: padding)
        self._gS = t32(S)                                            # separator slot -> unknown
        self._widx = t32(widx.reshape(-1))
        self.y = z64(k * s)
        self.bS = z64(nS)
        self.xS = z64(nS + 2)                                        # (zero slot nS for the padded window entries)
        self._acc = z64(n)
        self.blocks = None
        self.Wm = None
        self.Sinv = None
        self._nnz = int(cols.size)

    _place = BandedBlockSolver._place

    def factor(self, A):
        if A.nnz != self._nnz:
            raise ValueError("coarse operator changed its sparsity pattern")
        k, s, cwp, nS = self.k, self.s, self.cwp, self.nS
        dev = A.vals.device
        v = A.vals
        dense = self._place(v, self._src_II, self._dst_II, k * s * s)
        if self._pad_dst.numel():
            self.ops.scatter(self._pad_dst, torch.ones(self._pad_dst.numel(), dtype=F64, device=dev), dense)
        self.blocks = dense_inverse(dense.view(k, s, s))
        ais = self._place(v, self._src_IS, self._dst_IS, k * s * cwp)
        asi = self._place(v, self._src_SI, self._dst_SI, k * cwp * s)
        if v.is_cuda and hasattr(self.ops, "gemm"):
            self.Wm = self.ops.gemm(self.blocks, ais.view(k, s, cwp), torch.empty((k, s, cwp), dtype=F64, device=dev))
            upd = self.ops.gemm(asi.view(k, cwp, s), self.Wm, torch.empty((k, cwp, cwp), dtype=F64, device=dev)).reshape(-1)
        else:
            self.Wm = torch.bmm(self.blocks, ais.view(k, s, cwp)).contiguous()           # k x s x cwp: A_II^-1 A_IS
            upd = torch.bmm(asi.view(k, cwp, s), self.Wm).reshape(-1)                    # k x cwp x cwp
        Sc = self._place(v, self._src_SS, self._dst_SS, nS * nS)
        for sel, dst in self._upd:                       # S[dst] -= upd[sel], every destination once per pass
            tmp = torch.empty(sel.numel(), dtype=F64, device=dev)
            cur = torch.empty(sel.numel(), dtype=F64, device=dev)
            self.ops.gather(sel, upd, tmp)
            self.ops.gather(dst, Sc, cur)
            self.ops.axpby(-1.0, tmp, 1.0, cur)
            self.ops.scatter(dst, cur, Sc)
        self.Sinv = dense_inverse(Sc.view(nS, nS))
        self.ops.gather(self._SI_src, v, self.A_SI.vals)
        self.A_SI.invalidate_packed()

    def apply(self, b, x, accumulate=False):
        """x = A^-1 b; accumulate: x += A^-1 b."""
        o = self.ops
        o.coarse_front_gather(self.blocks, b, self._gI, self.y, self._gS, self.bS)       # y_I = A_II^-1 b_I ; b_S on the side
        o.csr_spmv(self.A_SI, self.y, self.bS, -1.0, 1.0)                                # g_S = b_S - A_SI y_I
        o.dense_gemv(self.Sinv, self.bS, self.xS[:self.nS])                              # x_S = S^-1 g_S
        o.coarse_back_gather(self.Wm, self.xS, self._widx, self.y, -1.0, self._gI, self._gS, x, accumulate)

    def bytes_per_apply(self):
        return 8 * (self.k * self.s * self.s + self.k * self.s * self.cwp + self.nS * self.nS)


class BlockCyclicReduction:
    """Exact block elimination of a banded operator by block cyclic reduction: the unknowns (in their
    natural order, or in reverse Cuthill-McKee order when that is narrower) are cut into m blocks of b >=
    half-bandwidth unknowns, which makes the operator block tridiagonal; every level eliminates the odd
    blocks

        x_i = D_i^-1 (b_i - L_i x_{i-1} - U_i x_{i+1})                                  (i odd)
        D'_j = D_j - L_j D_{j-1}^-1 U_{j-1} - U_j D_{j+1}^-1 L_{j+1},  L'_j = -L_j D_{j-1}^-1 L_{j-1}, ...  (j even)

    until one block is left (log2 m levels).  All inverses are formed explicitly at setup (batched GEMMs),
    so an application is 3 batched GEMV launches per level on dense b x b / b x 2b blocks -- about 5 n b
    doubles of HBM traffic and no sequential triangular solve: 257^2 unknowns of a 25-point operator
    (b = 516) 1.4 GB, 513^2 (b = 1028) 11 GB.  This is what lifts the size limit of the other two solvers
    for the scripts' 2-level runs (coarse `spsolve` on 66 k - 263 k unknowns, test/thesis_compare_2D.py:430-435).
    No pivoting across blocks (like BandedBlockSolver); the caller's refinement step and dense_inverse's
    own verification guard the accuracy."""
    kind = "block-cyclic-reduction"

    @staticmethod
    def estimate_bytes(n, b):
        m = -(-n // b)
        tot = 0
        while m > 1:
            mo, me = m // 2, m - m // 2
            tot += mo * b * b + me * 2 * b * b + mo * 2 * b * b
            m = me
        return 8 * (tot + b * b)

    def __init__(self, A, ops_mod, perm=None, w=None):
        self.ops = ops_mod
        self._symbolic(A, perm, w)
        self.factor(A)

    def _symbolic(self, A, perm, w):
        dev = A.device
        n = A.shape[0]
        rp, ci = A.rowptr.cpu().numpy(), A.colidx.cpu().numpy()
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
        cols = ci.astype(np.int64)
        if perm is not None:
            inv = np.empty(n, dtype=np.int64)
            inv[np.asarray(perm, dtype=np.int64)] = np.arange(n)
            rows, cols = inv[rows], inv[cols]
        if w is None:
            w = int(np.abs(rows - cols).max()) if cols.size else 1
        b = max(int(w), 2)
        b += b % 2
        m = -(-n // b)
        self.n, self.b, self.m, self.npad = n, b, m, m * b
        I, J = rows // b, cols // b
        if np.any(np.abs(I - J) > 1):
            raise ValueError("block size below the bandwidth: the operator is not block tridiagonal")
        t = lambda a, dt=torch.int64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
        loc = (rows % b) * b + (cols % b)
        eid = np.arange(cols.size, dtype=np.int64)
        self._maps = []
        if m * b * b >= 2 ** 31:
            raise ValueError("block cyclic reduction: dense blocks beyond int32 indexing")
        for d in (0, -1, 1):                                   # D, L (block column I-1), U (I+1)
            msk = (J - I) == d
            self._maps.append((t(eid[msk], torch.int32), t(I[msk] * b * b + loc[msk], torch.int32)))
        # padding rows (unknowns n .. npad-1) are identity rows
        pad = np.arange(n, self.npad, dtype=np.int64)
        self._pad_dst = t((pad // b) * b * b + (pad % b) * b + (pad % b), torch.int32)
        self.perm = None if perm is None else t(perm, torch.int32)
        self._nnz = int(cols.size)
        # vectors of every level: right-hand sides B, D_odd^-1 b_odd C (one zero block in front and behind),
        # solutions X (one zero block behind)
        z = lambda k: torch.zeros(k * b, dtype=F64, device=dev)
        self.B, self.C, self.X, sizes = [], [], [], []
        mm = m
        while True:
            sizes.append(mm)
            self.B.append(z(mm))
            self.X.append(z(mm + 1))
            if mm == 1:
                break
            self.C.append(z(mm // 2 + 2))
            mm = mm - mm // 2
        self.sizes = sizes
        self.levels = []
        self.root = None

    def factor(self, A):
        if A.nnz != self._nnz:
            raise ValueError("coarse operator changed its sparsity pattern")
        b, m = self.b, self.m
        dev = A.vals.device
        v = A.vals
        stacks = []
        for src, dst in self._maps:
            st = torch.zeros(m * b * b, dtype=F64, device=dev)
            if src.numel():
                tmp = torch.empty(src.numel(), dtype=F64, device=dev)
                self.ops.gather(src, v, tmp)
                self.ops.scatter(dst, tmp, st)
            stacks.append(st)
        if self._pad_dst.numel():
            self.ops.scatter(self._pad_dst, torch.ones(self._pad_dst.numel(), dtype=F64, device=dev), stacks[0])
        D, L, U = (st.view(m, b, b) for st in stacks)
        self.levels = []
        while D.shape[0] > 1:
            mm = D.shape[0]
            mo, me = mm // 2, mm - mm // 2
            Dinv = dense_inverse(D[1::2].contiguous())
            Hm, Hp = torch.bmm(Dinv, L[1::2]), torch.bmm(Dinv, U[1::2])
            Le, Ue = L[0::2], U[0::2]
            zero = torch.zeros((1, b, b), dtype=F64, device=dev)
            Hp_prev = torch.cat([zero, Hp], 0)[:me]                    # H+ of the odd block left of even block k
            Hm_next = torch.cat([Hm, zero], 0)[:me]                    # H- of the odd block right of it
            Dn = D[0::2] - torch.bmm(Le, Hp_prev) - torch.bmm(Ue, Hm_next)
            Hm_prev = torch.cat([zero, Hm], 0)[:me]
            Hp_next = torch.cat([Hp, zero], 0)[:me]
            Ln = -torch.bmm(Le, Hm_prev)
            Un = -torch.bmm(Ue, Hp_next)
            self.levels.append((Dinv.contiguous(), torch.cat([Le, Ue], 2).contiguous(), torch.cat([Hm, Hp], 2).contiguous()))
            D, L, U = Dn.contiguous(), Ln.contiguous(), Un.contiguous()
        self.root = dense_inverse(D[0].contiguous()).view(1, b, b).contiguous()

    def apply(self, rhs, x):
        o, b, n = self.ops, self.b, self.n
        B0 = self.B[0]
        if self.perm is None:
            o.copy(rhs, B0[:n])
        else:
            o.gather(self.perm, rhs, B0[:n])
        for l, (Dinv, LU, _H) in enumerate(self.levels):
            mm = self.sizes[l]
            mo, me = mm // 2, mm - mm // 2
            Bl, Cl, Bn = self.B[l], self.C[l], self.B[l + 1]
            o.dense_gemv_windows(Dinv, Bl[b:], 2 * b, Cl[b:], b)                               # c_k = D^-1 b_(2k+1)
            o.dense_gemv_windows(LU, Cl, b, Bn, b, z=Bl, z_stride=2 * b, alpha=-1.0)           # b'_k = b_2k - [L U][c_k-1; c_k]
        last = len(self.levels)
        o.dense_gemv_windows(self.root, self.B[last], b, self.X[last], b)
        for l in range(last - 1, -1, -1):
            mm = self.sizes[l]
            mo, me = mm // 2, mm - mm // 2
            Xl, Xn, Cl = self.X[l], self.X[l + 1], self.C[l]
            H = self.levels[l][2]
            o.dense_gemv_windows(H, Xn, b, Xl[b:], 2 * b, z=Cl[b:], z_stride=b, alpha=-1.0)     # odd blocks
            o.block_copy(me, b, Xn, b, Xl, 2 * b)                                             # even blocks
        if self.perm is None:
            o.copy(self.X[0][:n], x)
        else:
            o.scatter(self.perm, self.X[0][:n], x)

    def bytes_per_apply(self):
        return 8 * (sum(int(t.numel()) for lev in self.levels for t in lev) + int(self.root.numel()))


BANDED_MAX_BYTES = 1 << 30          # above this the one-level banded solver gives way to cyclic reduction
DENSE_PREFERRED_BYTES = 256 << 20   # non-banded operators this small keep the plain dense inverse


def _rcm_order(Ah):
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    pat = sp.csr_matrix((np.ones(Ah.nnz, dtype=np.int8), Ah.indices, Ah.indptr), shape=Ah.shape)
    pat = (pat + pat.T).tocsr()
    perm = np.asarray(reverse_cuthill_mckee(pat, symmetric_mode=True), dtype=np.int64)
    inv = np.empty(perm.size, dtype=np.int64)
    inv[perm] = np.arange(perm.size)
    coo = Ah.tocoo()
    w = int(np.abs(inv[coo.row] - inv[coo.col]).max()) if coo.nnz else 0
    return perm, w


def make_coarse_solver(A, ops_mod, strategy="auto"):
    """strategy: "auto" | "dense" | "banded" | "bcr".
    auto: dense inverse below 2048 unknowns; the one-level banded solver while its factors stay below 1 GB;
    block cyclic reduction (natural or reverse Cuthill-McKee order, whichever is narrower) beyond that;
    the dense inverse as the last resort (up to MAX_DENSE unknowns)."""
    n = A.shape[0]
    if strategy not in ("auto", "dense", "banded", "bcr", "grid"):
        raise ValueError("unknown coarse solver strategy %r" % (strategy,))
    if strategy == "dense" or n < 2048 and strategy == "auto":
        return DenseInverse(A, ops_mod)
    Ah = sp.csr_matrix((A.vals.cpu().numpy(), A.colidx.cpu().numpy(), A.rowptr.cpu().numpy()), shape=A.shape)
    w = max(half_bandwidth(Ah), 1)
    plan = BandedBlockSolver.plan(n, w) if strategy in ("auto", "banded") else None
    if strategy in ("auto", "grid") and hasattr(ops_mod, "coarse_front_gather"):
        # grid operators: blocks cut in both directions (a third of the bytes of whole-line strips)
        grid = GridBlockSolver.detect_grid(n, Ah)
        gplan = GridBlockSolver.plan(*grid) if grid is not None else None
        banded_est = None if plan is None else 8 * (2 * plan[0] * plan[1] ** 2 + (n - plan[0] * plan[1]) ** 2)
        if gplan is not None and (strategy == "grid" or (gplan[2] <= BANDED_MAX_BYTES and
                                                         (banded_est is None or gplan[2] < banded_est))):
            try:
                return GridBlockSolver(A, ops_mod, grid[0], grid[1], grid[2], gplan[0], gplan[1])
            except (ValueError, RuntimeError):
                if strategy == "grid":
                    raise
        elif strategy == "grid":
            raise ValueError("operator is not a grid operator of small radius: cannot use the grid coarse solver")
    if strategy in ("auto", "banded"):
        if plan is not None:
            k, s = plan
            est = 8 * (2 * k * s * s + (n - k * s) ** 2)
            if strategy == "banded" or est <= BANDED_MAX_BYTES:
                try:
                    return BandedBlockSolver(A, ops_mod, k, s)
                except (ValueError, RuntimeError):
                    if strategy == "banded":
                        raise
        elif strategy == "banded":
            raise ValueError("operator is not narrow-banded: cannot use the banded coarse solver")
    if strategy == "auto" and 8 * n * n <= DENSE_PREFERRED_BYTES:
        return DenseInverse(A, ops_mod)                       # small and not banded: one GEMV beats 3 log2(m) launches
    # block cyclic reduction, in the narrower of the natural and the reverse Cuthill-McKee ordering
    perm, wr = (None, w)
    if w > 4 * int(np.sqrt(n)) + 8:
        p2, w2 = _rcm_order(Ah)
        if w2 < w:
            perm, wr = p2, max(w2, 1)
    free = torch.cuda.mem_get_info(A.device)[0] if A.device.type == "cuda" else 1 << 62
    est = BlockCyclicReduction.estimate_bytes(n, wr + wr % 2)
    if 4 * wr <= n and 3 * est <= free:                       # (factor() holds about three copies while it runs)
        try:
            return BlockCyclicReduction(A, ops_mod, perm, wr)
        except (ValueError, RuntimeError):
            if strategy == "bcr":
                raise
    elif strategy == "bcr":
        raise ValueError("block cyclic reduction does not fit: bandwidth %d of %d unknowns, %d MB" % (wr, n, est >> 20))
    return DenseInverse(A, ops_mod)
