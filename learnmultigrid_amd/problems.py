"""Synthetic problem generators (host side, vectorised NumPy/SciPy; seeded).

They produce the inputs of BASELINE.json's configs -- matrices, right-hand sides and
transfer operators -- without the reference's per-element Python loops.  Conventions
follow the reference so its own small assemblies (tests/golden/g2*, g4*) are
reproduced exactly:
  * 1-D FD Poisson with Dirichlet rows as identity: utilities/laplacian.py:22-59,
    right-hand side assembly/LoadVector.py:53-62 with f == 1.
  * 2-D structured P1 Poisson on the unit square, row-major node numbering
    k = j*(m+1)+i (mesh/Mesh2D.py:63-93), squares split along the (k, k+m+2)
    diagonal => 5-point rows [-1,-1,4,-1,-1]; boundary rows overwritten by identity
    rows WITHOUT touching the columns (test/thesis_structured_2d.py:407-414).
  * transfer operators: the reference only has the 1-D geometric interpolator
    (solvers/Multigrid.py:126-147); its tensor product is the build-defined 2-D
    geometric transfer, and `learned_like` is a row-stochastic perturbation standing
    in for the absent networks (unit row sums: Multigrid.py:367-368,:758-759).
"""
import numpy as np
import scipy.sparse as sp


def level_sizes(n, levels):
    """n_{l+1} = floor((n_l - 1)/2) + 1  (Multigrid.py:130)."""
    out = [int(n)]
    for _ in range(int(levels) - 1):
        out.append((out[-1] - 1) // 2 + 1)
    return out


# ------------------------------------------------------------------ 1-D ------------
def poisson_1d_fd(ne):
    """A (csr, n = ne+1) and rhs (n,1) of laplacian_1d_fd_bc(Mesh1D(True, ne), f==1)."""
    n = ne + 1
    h = (1.0 - 0.0) / (n - 1)
    s = 1.0 / h ** 2
    main = np.full(n, 2.0 * s)
    lo = np.full(n - 1, -1.0 * s)
    up = np.full(n - 1, -1.0 * s)
    A = sp.diags([lo, main, up], [-1, 0, 1], format="lil")
    A[1, 0] = 0
    A[n - 2, n - 1] = 0
    A[0, :] = 0
    A[n - 1, :] = 0
    A[0, 0] = 1
    A[n - 1, n - 1] = 1
    A = sp.csr_matrix(A)
    A.eliminate_zeros()
    # LoadVector.compute_rhs_1d with f == 1: interior nodes get h_left/2 + h_right/2
    x = np.linspace(0, 1, n)
    hs = np.diff(x)
    rhs = np.zeros((n, 1))
    rhs[:-1, 0] += 0.5 * hs
    rhs[1:, 0] += 0.5 * hs
    rhs[0] = 0
    rhs[-1] = 0
    return A, rhs


def geometric_interpolator_1d(n):
    """Sparse CSR form of Multigrid.interpolator(n) (Multigrid.py:126-147), same values:
    interior coarse node j -> (1/2, 1, 1/2) on fine rows 2j-1..2j+1; first column (1, 1/2)
    on rows 0,1; last column (1/2, 1) on the LAST two rows (even n: Appendix B quirk)."""
    n = int(n)
    nc = (n - 1) // 2 + 1
    j = np.arange(1, nc - 1)
    rows = np.concatenate([2 * j - 1, 2 * j, 2 * j + 1])
    cols = np.concatenate([j, j, j])
    vals = np.concatenate([np.full(j.size, 0.5), np.full(j.size, 1.0), np.full(j.size, 0.5)])
    P = sp.lil_matrix((n, nc))
    P[rows, cols] = vals
    for (r, c, v) in ((0, 0, 1.0), (1, 0, 0.5), (n - 1, nc - 1, 1.0), (n - 2, nc - 1, 0.5)):
        P[r, c] = v                      # assigned after the interior, in reference order
    return sp.csr_matrix(P)


# ------------------------------------------------------------------ 2-D ------------
def poisson_2d_structured(m, dirichlet=True):
    """P1 stiffness of -Laplace on the structured triangulation with m elements per
    side ((m+1)^2 nodes) and the load vector of f == -1, built directly as CSR.

    Returns (A csr int32/fp64, rhs (n,1)).  Interior rows are [-1,-1,4,-1,-1] at
    columns k-(m+1), k-1, k, k+1, k+(m+1); with dirichlet=True boundary rows are
    identity rows and rhs is 0 there (columns untouched => A is not symmetric)."""
    s = m + 1
    n = s * s
    h = 1.0 / m
    idx = np.arange(n, dtype=np.int64)
    i, j = idx % s, idx // s
    boundary = (i == 0) | (i == m) | (j == 0) | (j == m)
    if not dirichlet:
        raise NotImplementedError("free (Neumann) rows are only needed for golden checks")
    nnz_row = np.where(boundary, 1, 5).astype(np.int64)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(nnz_row, out=rowptr[1:])
    nnz = int(rowptr[-1])
    colidx = np.empty(nnz, dtype=np.int32)
    vals = np.empty(nnz, dtype=np.float64)
    b = np.flatnonzero(boundary)
    colidx[rowptr[b]] = b
    vals[rowptr[b]] = 1.0
    it = np.flatnonzero(~boundary)
    p = rowptr[it]
    for off, (dc, v) in enumerate(((-s, -1.0), (-1, -1.0), (0, 4.0), (1, -1.0), (s, -1.0))):
        colidx[p + off] = it + dc
        vals[p + off] = v
    A = sp.csr_matrix((vals, colidx, rowptr.astype(np.int32)), shape=(n, n))
    rhs = np.where(boundary, 0.0, -(h * h)).reshape(n, 1)
    return A, rhs


def tensor_interpolator_2d(s):
    """Build-defined 2-D geometric transfer: tensor product of the reference's 1-D
    interpolator with itself on an s x s row-major grid -> ((s*s), (sc*sc)) CSR."""
    P1 = geometric_interpolator_1d(s)
    P = sp.kron(P1, P1, format="csr")
    P.sort_indices()
    return P


def geometric_hierarchy_2d(s, levels):
    """[P_0, ..., P_{levels-2}], P_l maps level l+1 (coarser) to level l."""
    sizes = level_sizes(s, levels)
    return [tensor_interpolator_2d(sz) for sz in sizes[:-1]]


def geometric_hierarchy_1d(n, levels):
    return [geometric_interpolator_1d(sz) for sz in level_sizes(n, levels)[:-1]]


def pseudo_l2_interpolator_1d(n):
    """1-D "pseudo-L2"-shaped transfer on a regular nested mesh (Appendix B of SURVEY.md:
    coincident rows (1/12, 5/6, 1/12), in-between rows (1/2, 1/2), rows renormalised
    at the ends): 3 nnz on even rows, 2 on odd rows -- the sparsity class of the
    reference's L2Projection("pseudo") / learned Q."""
    n = int(n)
    nc = (n - 1) // 2 + 1
    P = sp.lil_matrix((n, nc))
    for r in range(n):
        c = r // 2
        if r % 2 == 0:
            c = min(c, nc - 1)
            for dc, v in ((-1, 1.0 / 12), (0, 5.0 / 6), (1, 1.0 / 12)):
                if 0 <= c + dc < nc:
                    P[r, c + dc] = v
        else:
            for dc in (0, 1):
                if 0 <= c + dc < nc:
                    P[r, c + dc] = 0.5
    P = sp.csr_matrix(P)
    return row_normalise(P)


def row_normalise(Q):
    Q = sp.csr_matrix(Q, dtype=np.float64)
    rs = np.asarray(Q.sum(axis=1)).ravel()
    rs[rs == 0] = 1.0
    return sp.csr_matrix(sp.diags(1.0 / rs) @ Q)


def learned_like(P, seed, jitter=0.05):
    """Row-stochastic perturbation of a transfer operator: entries x (1 + jitter*N(0,1)),
    then Q /= rowsum -- same sparsity and unit-row-sum contract as the learned Q."""
    P = sp.csr_matrix(P, dtype=np.float64).copy()
    rng = np.random.default_rng(seed)
    P.data = P.data * (1.0 + jitter * rng.standard_normal(P.data.size))
    return row_normalise(P)


def p1_stiffness_2d(px, py, m, coeff=None):
    """Vectorised P1 stiffness on the structured triangulation topology (m x m squares,
    each split into (k, k+1, k+m+2) and (k, k+m+2, k+m+1) like Mesh2D.construct) with
    arbitrary node coordinates (jittered => 7-point, "unstructured-like") and an
    optional per-element coefficient (variable-coefficient -div(k grad u))."""
    s = m + 1
    sq = (np.arange(m)[:, None] * s + np.arange(m)[None, :]).ravel()
    t1 = np.stack([sq, sq + 1, sq + s + 1], axis=1)
    t2 = np.stack([sq, sq + s + 1, sq + s], axis=1)
    tri = np.concatenate([t1, t2], axis=0)
    x, y = px[tri], py[tri]
    # gradients of the barycentric basis
    bx = np.stack([y[:, 1] - y[:, 2], y[:, 2] - y[:, 0], y[:, 0] - y[:, 1]], axis=1)
    by = np.stack([x[:, 2] - x[:, 1], x[:, 0] - x[:, 2], x[:, 1] - x[:, 0]], axis=1)
    det = (x[:, 1] - x[:, 0]) * (y[:, 2] - y[:, 0]) - (x[:, 2] - x[:, 0]) * (y[:, 1] - y[:, 0])
    scale = 1.0 / (2.0 * det)
    if coeff is not None:
        scale = scale * coeff
    loc = (bx[:, :, None] * bx[:, None, :] + by[:, :, None] * by[:, None, :]) * scale[:, None, None]
    rows = np.repeat(tri, 3, axis=1).ravel()
    cols = np.tile(tri, (1, 3)).ravel()
    A = sp.coo_matrix((loc.ravel(), (rows, cols)), shape=(s * s, s * s)).tocsr()
    A.sum_duplicates()
    return A, np.abs(det)


def apply_dirichlet_identity_rows(A, rhs, boundary):
    """Boundary rows -> identity rows, rhs -> 0 there; columns untouched
    (test/thesis_structured_2d.py:407-414)."""
    A = sp.csr_matrix(A)
    keep = sp.diags((~boundary).astype(np.float64))
    A = sp.csr_matrix(keep @ A + sp.diags(boundary.astype(np.float64)))
    A.eliminate_zeros()
    A.sort_indices()
    rhs = rhs.copy()
    rhs[boundary] = 0
    return A, rhs


def jittered_poisson_2d(m, seed=42, jitter=0.25, coeff_sigma=None, coeff_seed=44):
    """cfg#3 / cfg#5 style problem: structured topology with interior nodes moved by
    U(-jitter*h, jitter*h) (7-point P1 stiffness), optional log-normal element
    coefficient k = exp(sigma * N(0,1)); load vector of f == -1; Dirichlet identity rows."""
    s = m + 1
    h = 1.0 / m
    g = np.linspace(0.0, 1.0, s)
    px = np.tile(g, s)
    py = np.repeat(g, s)
    idx = np.arange(s * s)
    i, j = idx % s, idx // s
    boundary = (i == 0) | (i == m) | (j == 0) | (j == m)
    rng = np.random.default_rng(seed)
    dx = rng.uniform(-jitter * h, jitter * h, s * s)
    dy = rng.uniform(-jitter * h, jitter * h, s * s)
    px = np.where(boundary, px, px + dx)
    py = np.where(boundary, py, py + dy)
    coeff = None
    if coeff_sigma is not None:
        coeff = np.exp(coeff_sigma * np.random.default_rng(coeff_seed).standard_normal(2 * m * m))
    A, det = p1_stiffness_2d(px, py, m, coeff)
    # P1 load vector of f == -1: each triangle gives -area/3 = -det/6 to its 3 nodes
    sq = (np.arange(m)[:, None] * s + np.arange(m)[None, :]).ravel()
    tri = np.concatenate([np.stack([sq, sq + 1, sq + s + 1], 1), np.stack([sq, sq + s + 1, sq + s], 1)], 0)
    rhs = np.zeros(s * s)
    np.add.at(rhs, tri.ravel(), np.repeat(-det / 6.0, 3))
    A, rhs = apply_dirichlet_identity_rows(A, rhs.reshape(-1, 1), boundary)
    return A, rhs


def variable_coeff_poisson_2d_structured(m, seed=44, sigma=0.5, coeff=None):
    """cfg#5 problem: -div(k grad u) with one coefficient per triangle, P1 on the structured
    triangulation, Dirichlet identity rows, load vector of f == -1 -- built directly as CSR
    (5-point rows) so that 8193^2 nodes need no 1.2 G-entry COO detour.  The reference has no
    variable-coefficient assembly (test/thesis_variableCoeff_stiff.py is empty); this is
    StiffnessMatrix.loc_a_2d (StiffnessMatrix.py:53-59) with each element matrix scaled by
    k_e, cross-checked against p1_stiffness_2d(coeff=...) in the tests.

    coeff: optional (2*m*m,) array, first the lower-right triangles (k, k+1, k+s+1) of all
    squares row-major, then the upper-left ones (k, k+s+1, k+s); default exp(sigma*N(0,1))."""
    s = m + 1
    n = s * s
    h = 1.0 / m
    if coeff is None:
        coeff = np.exp(sigma * np.random.default_rng(seed).standard_normal(2 * m * m))
    k1 = np.zeros((m + 1, m + 1))
    k2 = np.zeros((m + 1, m + 1))
    k1[:m, :m] = coeff[: m * m].reshape(m, m)          # [j, i]: triangle with legs bottom/right
    k2[:m, :m] = coeff[m * m:].reshape(m, m)           # [j, i]: triangle with legs left/top
    # edge weights: a leg shared by two triangles carries -(k_a + k_b)/2, hypotenuses carry 0
    k2_below = np.zeros((s, s))
    k2_below[1:, :] = k2[:m, :]                        # square (i, j-1)
    k1_left = np.zeros((s, s))
    k1_left[:, 1:] = k1[:, :m]                         # square (i-1, j)
    w_right = -(k1 + k2_below) / 2.0                   # edge (i,j)-(i+1,j), defined for i < m
    w_up = -(k2 + k1_left) / 2.0                       # edge (i,j)-(i,j+1), defined for j < m
    idx = np.arange(n, dtype=np.int64)
    i, j = idx % s, idx // s
    boundary = (i == 0) | (i == m) | (j == 0) | (j == m)
    nnz_row = np.where(boundary, 1, 5).astype(np.int64)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(nnz_row, out=rowptr[1:])
    nnz = int(rowptr[-1])
    colidx = np.empty(nnz, dtype=np.int32)
    vals = np.empty(nnz, dtype=np.float64)
    b = np.flatnonzero(boundary)
    colidx[rowptr[b]] = b
    vals[rowptr[b]] = 1.0
    it = np.flatnonzero(~boundary)
    ii, jj = i[it], j[it]
    wd = w_up[jj - 1, ii]
    wl = w_right[jj, ii - 1]
    wr = w_right[jj, ii]
    wu = w_up[jj, ii]
    p = rowptr[it]
    for off, (dc, v) in enumerate(((-s, wd), (-1, wl), (0, -(wd + wl + wr + wu)), (1, wr), (s, wu))):
        colidx[p + off] = it + dc
        vals[p + off] = v
    A = sp.csr_matrix((vals, colidx, rowptr.astype(np.int32)), shape=(n, n))
    rhs = np.where(boundary, 0.0, -(h * h)).reshape(n, 1)
    return A, rhs
