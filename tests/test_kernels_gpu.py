"""Parity of every HIP kernel behind the C ABI against the CPU oracle (-m gpu).

Bars: bit-exact for SpMV / residual / Jacobi / Gauss-Seidel / SpGEMM values and all
integer outputs (same accumulation order, no FMA contraction on either side);
1e-13 relative for the reductions whose summation tree differs (norms, dot, GEMV).
"""
import math

import numpy as np
import pytest
import scipy.sparse as sp

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from learnmultigrid_amd import ops, problems as P   # noqa: E402
from oracle import kernels as K                     # noqa: E402  (checker only)

DEV = "cuda:0"


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def ragged_matrix(n, seed, long_row=None, empty_every=7, long_len=5000):
    """Random nonsymmetric CSR with empty rows, missing diagonals and (optionally) one
    row far longer than any LDS tile."""
    rng = np.random.default_rng(seed)
    A = sp.random(n, n, density=min(1.0, 6.0 / n), random_state=seed, format="lil")
    A.setdiag(rng.uniform(2, 4, n))
    for r in range(3, n, empty_every):
        A[r, :] = 0
    if long_row is not None:
        cols = rng.choice(n, size=min(n, long_len), replace=False)
        A[long_row, cols] = rng.standard_normal(cols.size)
    A = sp.csr_matrix(A)
    A.eliminate_zeros()
    return K.as_csr(A)


import functools


@functools.lru_cache(maxsize=None)
def case(name):
    """Matrices are built lazily so that collecting this module on a CPU-only run is free."""
    if name == "poisson1d_1025":
        return K.as_csr(P.poisson_1d_fd(1024)[0])
    if name == "poisson2d_65":
        return K.as_csr(P.poisson_2d_structured(64)[0])
    if name == "poisson2d_513":
        return K.as_csr(P.poisson_2d_structured(512)[0])
    if name == "ragged_1000":
        return ragged_matrix(1000, 1)
    if name == "ragged_longrow_6007":
        return ragged_matrix(6007, 2, long_row=4001)
    if name == "tiny_3":
        return K.as_csr(sp.csr_matrix(np.array([[30., 1, 15], [28, 60, 3], [100, 19, 150]])))
    if name == "prolong_65":
        return K.as_csr(P.tensor_interpolator_2d(65))
    if name == "restrict_65":
        return K.as_csr(P.tensor_interpolator_2d(65).T.tocsr())
    if name == "l2like_restrict":
        return K.as_csr(sp.kron(P.pseudo_l2_interpolator_1d(65),
                                P.pseudo_l2_interpolator_1d(65)).T.tocsr())
    raise KeyError(name)


SQUARE = ["poisson1d_1025", "poisson2d_65", "poisson2d_513", "ragged_1000", "ragged_longrow_6007",
          "tiny_3"]
ALL = SQUARE + ["prolong_65", "restrict_65", "l2like_restrict"]


@pytest.fixture(params=list(range(7)), ids=lambda r: "variant%d" % r)
def rpt(request):
    ops.tune_set("sweep_variant", request.param)
    yield request.param
    ops.tune_set("sweep_variant", 0)


@pytest.mark.parametrize("name", ALL)
def test_spmv_bit_exact(name, rpt):
    A = case(name)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(A.shape[1])
    y0 = rng.standard_normal(A.shape[0])
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    for alpha, beta in ((1.0, 0.0), (1.0, 1.0), (-0.5, 2.0)):
        y = dev(y0.copy())
        ops.csr_spmv(dA, dev(x), y, alpha, beta)
        want = K.spmv(A, x, y0, alpha, beta)
        assert np.array_equal(y.cpu().numpy(), want), (name, alpha, beta)
    y = dev(np.zeros(A.shape[0]))
    ops.csr_spmv(dA, dev(x), y, 1.0, 0.0)
    assert np.array_equal(y.cpu().numpy(), A @ x)              # SciPy itself


@pytest.mark.parametrize("name", SQUARE)
def test_residual_and_jacobi_bit_exact(name, rpt):
    A = case(name)
    rng = np.random.default_rng(6)
    n = A.shape[0]
    x, b = rng.standard_normal(n), rng.standard_normal(n)
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dx, db = dev(x), dev(b)
    r = torch.empty(n, dtype=torch.float64, device=DEV)
    part = torch.empty(ops.partials_count(n), dtype=torch.float64, device=DEV)
    n2 = torch.zeros(1, dtype=torch.float64, device=DEV)
    ops.csr_residual_norm2(dA, dx, db, r, part, n2)
    want_r, want_n2 = K.residual(A, x, b)
    assert np.array_equal(r.cpu().numpy(), want_r)
    assert abs(n2.item() - want_n2) <= 1e-13 * want_n2
    # norm only (no residual vector written) and residual only (no norm)
    n2b = torch.zeros(1, dtype=torch.float64, device=DEV)
    ops.csr_residual_norm2(dA, dx, db, None, part, n2b)
    assert n2b.item() == n2.item()                               # deterministic reduction
    r2 = torch.empty_like(r)
    ops.csr_residual_norm2(dA, dx, db, r2, None, None)
    assert torch.equal(r, r2)
    for omega in (1.0, 0.8):
        out = torch.empty_like(dx)
        ops.csr_jacobi(dA, dx, db, omega, out)
        assert np.array_equal(out.cpu().numpy(), K.jacobi(A, x, b, omega)), (name, omega)


@pytest.mark.parametrize("name", [c for c in SQUARE if c != "poisson2d_513"] + ["poisson2d_513"])
def test_gauss_seidel_lexicographic_bit_exact(name):
    A = case(name)
    rng = np.random.default_rng(7)
    n = A.shape[0]
    x, b = rng.standard_normal(n), rng.standard_normal(n)
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    sched = ops.build_gs_schedule(A, "lexicographic", DEV)
    dx = dev(x.copy())
    ops.csr_gs_schedule(dA, dx, dev(b), sched, sweeps=2)
    want = x.copy()
    K.gs_forward(A, want, b, 2)
    assert np.array_equal(dx.cpu().numpy(), want), name
    # every set is independent in A + A^T
    S = (abs(A) + abs(A.T)).tocsr()
    S.setdiag(0)
    S.eliminate_zeros()
    lab = np.empty(n, dtype=np.int64)
    rows = sched.d_rows.cpu().numpy()
    for s in range(sched.nsets):
        lab[rows[sched.h_ptr[s]:sched.h_ptr[s + 1]]] = s
    coo = S.tocoo()
    assert np.all(lab[coo.row] != lab[coo.col])


def _seven_point(m):
    """Constant-coefficient 7-point operator {-W-1, -W, -1, 0, 1, W, W+1} (P1 stiffness of a uniform mesh of
    sheared triangles) with identity boundary rows, columns untouched (like the reference's Dirichlet rows)."""
    W = m + 1
    n = W * W
    idx = np.arange(n)
    yy, xx = idx // W, idx % W
    inter = (xx > 0) & (xx < m) & (yy > 0) & (yy < m)
    rows, cols, vals = [idx[~inter]], [idx[~inter]], [np.ones((~inter).sum())]
    for off, v in ((-W - 1, -0.5), (-W, -1.0), (-1, -1.0), (0, 6.25), (1, -1.0), (W, -1.0), (W + 1, -0.5)):
        rows.append(idx[inter])
        cols.append(idx[inter] + off)
        vals.append(np.full(inter.sum(), v))
    return K.as_csr(sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr())


@pytest.mark.parametrize("name", ["poisson2d_129", "poisson2d_300", "galerkin_9pt", "galerkin_9pt_257",
                                  "with_empty_and_diagless_rows", "seven_point_150"])
def test_wavefront_gauss_seidel_bit_exact(name):
    """Exact forward Gauss-Seidel as a pipelined wavefront (gs_wave.hip: a wave per 64 grid lines, bands
    handed over through memory) against pyamg's sweep restated in oracle/lmg_oracle.c, bitwise, over several
    sweeps: one band and many, 5- / 7- / 9-point and 1-D, rows without a diagonal, empty rows."""
    if name == "poisson2d_300":
        A = K.as_csr(P.poisson_2d_structured(299)[0])
    elif name == "galerkin_9pt_257":
        Pm = P.tensor_interpolator_2d(513)
        A = K.as_csr(sp.csr_matrix(Pm.T @ P.poisson_2d_structured(512)[0] @ Pm))
    elif name == "seven_point_150":
        A = _seven_point(149)
    else:
        A = rpat_case(name)
    n = A.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dA.pack()
    assert dA.stencil is not None and dA.stencil.gs_ok and ops.stencil_gs_available(dA), name
    if name == "seven_point_150":
        assert dA.stencil.umask == 0x1BB
    rng = np.random.default_rng(41)
    x0, b = rng.standard_normal(n), rng.standard_normal(n)
    want = x0.copy()
    x = dev(x0.copy())
    db = dev(b)
    for sweeps in (1, 2, 3):
        K.lib().orc_csr_gs_forward(n, A.indptr, A.indices, A.data, want, b, sweeps)
        ops.stencil_gs(dA, x, db, sweeps)
        got = x.cpu().numpy()
        assert np.array_equal(got, want), (name, sweeps, np.flatnonzero(got != want)[:8])
    ops.stencil_gs_check(dA)
    # ... and the level-scheduled executors give the same bits (they stay the path of every other matrix)
    try:
        ops.set_wavefront_gs_enabled(False)
        assert not ops.stencil_gs_available(dA)
        pat = sp.csr_matrix((np.ones(A.nnz, dtype=np.int8), A.indices, A.indptr), shape=A.shape)
        sched = ops.build_gs_schedule(pat, "lexicographic", DEV)
        x2 = dev(x0.copy())
        ops.csr_gs_schedule(dA, x2, db, sched, 6)
        assert torch.equal(x2, x)
    finally:
        ops.set_wavefront_gs_enabled(True)


@pytest.mark.parametrize("name", ["poisson2d_300", "seven_point_150", "poisson2d_129", "poisson2d_1100", "ragged_lines_77x203",
                                  "galerkin9_151", "galerkin9_333", "galerkin9_ragged_70x151"])
def test_gauss_seidel_bands_staged_through_lds_bit_exact(name):
    """gs_band_lds_kernel (the band's lines in 16-column chunks through LDS-DMA, results written back into the tile,
    coalesced chunk stores; what single sweeps of 5- / 7-point operators run beyond the Infinity Cache) forced on small
    grids: one band and many, partial last band, line strides that are no multiple of the chunk, a last line shorter
    than the others; 9-point Galerkin operators (two columns of skew per lane, 32-line bands: one band, several, a partial
    last band, a short last line) -- against pyamg's sweep restated in oracle/lmg_oracle.c, bitwise, and against the register
    kernel; 1 - 3 sweeps pipelined in one launch."""
    if name.startswith("galerkin9_"):
        side = {"galerkin9_151": 151, "galerkin9_333": 333, "galerkin9_ragged_70x151": 151}[name]
        Af = P.poisson_2d_structured(2 * (side - 1))[0]
        Pf = P.tensor_interpolator_2d(2 * (side - 1) + 1)
        A = sp.csr_matrix(Pf.T @ Af @ Pf)
        A.sort_indices()
        if name == "galerkin9_ragged_70x151":
            A = sp.csr_matrix(A[: 70 * 151 - 3][:, : 70 * 151 - 3])
        A = K.as_csr(A)
    elif name == "poisson2d_300":
        A = K.as_csr(P.poisson_2d_structured(299)[0])
    elif name == "poisson2d_1100":
        A = K.as_csr(P.poisson_2d_structured(1099)[0])
    elif name == "seven_point_150":
        A = _seven_point(149)
    elif name == "ragged_lines_77x203":
        # 203 columns, 77 lines minus 5 rows: the last line is short (n no multiple of the line stride)
        A = K.as_csr(P.poisson_2d_structured(202)[0])[: 77 * 203 - 5][:, : 77 * 203 - 5]
        A = K.as_csr(sp.csr_matrix(A))
    else:
        A = rpat_case(name)
    n = A.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dA.pack()
    assert dA.stencil is not None and ops.stencil_gs_available(dA) and dA.stencil.W >= 64, name
    assert bool(dA.stencil.umask & 4) == name.startswith("galerkin9_"), (name, hex(dA.stencil.umask))
    rng = np.random.default_rng(43)
    x0, b = rng.standard_normal(n), rng.standard_normal(n)
    want = x0.copy()
    x = dev(x0.copy())
    db = dev(b)
    try:
        ops.tune_set("gsw_lds", 1)
        for sweeps in (1, 2, 3):
            K.lib().orc_csr_gs_forward(n, A.indptr, A.indices, A.data, want, b, sweeps)
            ops.stencil_gs(dA, x, db, sweeps)
            got = x.cpu().numpy()
            assert np.array_equal(got, want), (name, sweeps, np.flatnonzero(got != want)[:8], n, dA.stencil.W)
        ops.stencil_gs_check(dA)
        ops.tune_set("gsw_lds", 0)
        x2 = dev(x0.copy())
        ops.stencil_gs(dA, x2, db, 6)
        assert torch.equal(x2, x)
    finally:
        ops.tune_set("gsw_lds", -1)


@pytest.mark.parametrize("m,kind", [(1024, "5pt"), (640, "9pt")])
def test_wavefront_gauss_seidel_sweeps_pipelined_in_one_launch(m, kind):
    """Several sweeps in ONE launch (band b of sweep s trails band b + 1 of sweep s - 1) against the oracle and against
    one launch per sweep, bitwise: 7 sweeps (chunks of 4 + 3) on 17 / 11 bands."""
    if kind == "5pt":
        A = K.as_csr(P.poisson_2d_structured(m)[0])
    else:
        Pm = P.tensor_interpolator_2d(2 * m + 1)
        A = K.as_csr(sp.csr_matrix(Pm.T @ P.poisson_2d_structured(2 * m)[0] @ Pm))
    n = A.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dA.pack()
    assert ops.stencil_gs_available(dA) and ops.tune_get("gsw_max_sweeps") == 4 and n <= ops.tune_get("gsw_multi_max_rows")
    rng = np.random.default_rng(8)
    x0, b = rng.standard_normal(n), rng.standard_normal(n)
    want = x0.copy()
    K.lib().orc_csr_gs_forward(n, A.indptr, A.indices, A.data, want, b, 7)
    x = dev(x0.copy())
    ops.stencil_gs(dA, x, dev(b), 7)
    ops.stencil_gs_check(dA)
    assert np.array_equal(x.cpu().numpy(), want)
    try:
        ops.tune_set("gsw_max_sweeps", 1)
        x1 = dev(x0.copy())
        ops.stencil_gs(dA, x1, dev(b), 7)
        assert torch.equal(x1, x)
    finally:
        ops.tune_set("gsw_max_sweeps", 4)


def test_wavefront_gauss_seidel_leaves_chains_to_the_chain_executor():
    dA = ops.DeviceCSR.from_scipy(rpat_case("poisson1d"), DEV)
    dA.pack()
    assert dA.stencil is not None and not dA.stencil.gs_ok and not ops.stencil_gs_available(dA)


def test_wavefront_gauss_seidel_refuses_coupling_across_line_ends():
    # a 5-point-like operator whose rows in column 0 also reach i - 1 (the end of the previous line): the wavefront
    # would need a value that is not computed yet, so the builder must send it to the level schedule
    W, n = 12, 144
    idx = np.arange(n)
    rows, cols, vals = [], [], []
    for off, v in ((-W, -1.0), (-1, -1.0), (0, 4.5), (1, -1.0), (W, -1.0)):
        ok = (idx + off >= 0) & (idx + off < n)
        rows.append(idx[ok]); cols.append(idx[ok] + off); vals.append(np.full(ok.sum(), v))
    A = K.as_csr(sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr())
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dA.pack()
    assert dA.stencil is not None and not dA.stencil.gs_ok and not ops.stencil_gs_available(dA)
    with pytest.raises(ops.LmgError):
        ops.stencil_gs(dA, dev(np.zeros(n)), dev(np.ones(n)), 1)


def test_wavefront_gauss_seidel_full_size_4097():
    """One exact forward sweep on the 16.8 M-row fine level of cfg#4 (65 bands in flight) against the oracle."""
    A = K.as_csr(P.poisson_2d_structured(4096)[0])
    n = A.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dA.pack()
    assert ops.stencil_gs_available(dA)
    rng = np.random.default_rng(4)
    x0, b = rng.standard_normal(n), rng.standard_normal(n)
    want = x0.copy()
    K.lib().orc_csr_gs_forward(n, A.indptr, A.indices, A.data, want, b, 1)
    x = dev(x0)
    ops.stencil_gs(dA, x, dev(b), 1)
    assert np.array_equal(x.cpu().numpy(), want)
    ops.stencil_gs_check(dA)


def test_gauss_seidel_ell_executor_and_its_fallbacks():
    """Medium schedules run on the pattern copy in schedule order (lmg_csr_gs_schedule_ell):
    same bits as the oracle, also after the VALUES changed (the copy only holds the pattern), and
    a schedule used with another matrix rebuilds its copy."""
    rng = np.random.default_rng(12)
    for name, want_k in (("poisson2d_513", 5), ("galerkin_9pt", 9)):
        A = case(name) if name == "poisson2d_513" else packed_case(name)
        n = A.shape[0]
        dA = ops.DeviceCSR.from_scipy(A, DEV)
        sched = ops.build_gs_schedule(A, "lexicographic", DEV)
        for trial in range(2):
            x, b = rng.standard_normal(n), rng.standard_normal(n)
            dx = dev(x.copy())
            ops.csr_gs_schedule(dA, dx, dev(b), sched, sweeps=3)
            assert sched.ell is not None and sched.ell[1] == want_k, (name, sched.ell[:2])
            want = x.copy()
            K.gs_forward(A, want, b, 3)
            assert np.array_equal(dx.cpu().numpy(), want), (name, trial)
            A = A.copy()
            A.data = A.data * (1.0 + 0.1 * rng.random(A.nnz))          # new coefficients, same pattern
            dA.vals.copy_(dev(A.data))
        # the same schedule object with a second matrix of the same pattern: copy is rebuilt for it
        dB = ops.DeviceCSR.from_scipy(A, DEV)
        key_before = sched.ell[0]
        dx = dev(x.copy())
        ops.csr_gs_schedule(dB, dx, dev(b), sched, sweeps=1)
        assert sched.ell[0] != key_before
        want = x.copy()
        K.gs_forward(A, want, b, 1)
        assert np.array_equal(dx.cpu().numpy(), want)
    # rows longer than 16 entries: no pattern copy, the CSR executors take over
    A = case("ragged_1000")
    sched = ops.build_gs_schedule(A, "lexicographic", DEV)
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    x, b = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
    dx = dev(x.copy())
    ops.csr_gs_schedule(dA, dx, dev(b), sched, sweeps=1)
    assert sched.ell is None or sched.ell[1] is None
    want = x.copy()
    K.gs_forward(A, want, b, 1)
    assert np.array_equal(dx.cpu().numpy(), want)


def test_gauss_seidel_wide_sets_use_per_set_launches():
    # 513^2 has 1025 anti-diagonal sets of up to 513 rows -> one-workgroup path;
    # a 2100^2 grid has sets of up to 2100 rows (> 2048) -> per-set launch path.
    A, _ = P.poisson_2d_structured(2100 - 1)
    A = K.as_csr(A)
    n = A.shape[0]
    rng = np.random.default_rng(8)
    x, b = rng.standard_normal(n), rng.standard_normal(n)
    sched = ops.build_gs_schedule(A, "lexicographic", DEV)
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dx = dev(x.copy())
    ops.csr_gs_schedule(dA, dx, dev(b), sched, sweeps=1)
    want = x.copy()
    K.gs_forward(A, want, b, 1)
    assert sched.max_set > ops.GS_ELL_MAX_SET and sched.ell is None
    assert np.array_equal(dx.cpu().numpy(), want)


@pytest.mark.parametrize("name", ["poisson2d_513", "ragged_1000", "poisson1d_1025"])
def test_gauss_seidel_multicolor_matches_cpu_twin(name):
    A = case(name)
    rng = np.random.default_rng(9)
    n = A.shape[0]
    x, b = rng.standard_normal(n), rng.standard_normal(n)
    sched = ops.build_gs_schedule(A, "multicolor", DEV)
    if name == "poisson2d_513":
        assert sched.nsets <= 4                                   # ~red-black (+ boundary colour)
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dx = dev(x.copy())
    ops.csr_gs_schedule(dA, dx, dev(b), sched, sweeps=2)
    want = x.copy()
    order = sched.d_rows.cpu().numpy()
    for _ in range(2):
        K.gs_rows(A, want, b, order)
    assert np.array_equal(dx.cpu().numpy(), want)
    # single independent set through the plain entry point
    dx2 = dev(x.copy())
    rows0 = sched.d_rows[: int(sched.h_ptr[1])]
    ops.csr_gs_rows(dA, dx2, dev(b), rows0)
    w2 = x.copy()
    K.gs_rows(A, w2, b, order[: int(sched.h_ptr[1])])
    assert np.array_equal(dx2.cpu().numpy(), w2)


@pytest.mark.parametrize("name", ["prolong_65", "restrict_65", "l2like_restrict", "ragged_1000", "poisson2d_513",
                                  "dense_rows"])
def test_device_transpose_equals_scipy(name):
    """R = P^T by counting sort on the device (csrc/transpose.hip; the library-sort path for rows longer than
    the insertion sort takes): indices and values identical to SciPy's A.T.tocsr()."""
    if name == "dense_rows":                       # dense 1-D transfer: rows of A^T with ~1000 entries -> library sort
        A = K.as_csr(sp.csr_matrix(np.random.default_rng(2).standard_normal((1030, 515))))
    else:
        A = case(name)
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    T = dA.transpose()
    want = sp.csr_matrix(A.T)
    want.sort_indices()
    assert T.shape == want.shape and T.nnz == want.nnz
    assert np.array_equal(T.rowptr.cpu().numpy(), want.indptr)
    assert np.array_equal(T.colidx.cpu().numpy(), want.indices)
    assert np.array_equal(T.vals.cpu().numpy(), want.data)
    T2 = dA.transpose()                            # deterministic although the fill uses atomics
    assert torch.equal(T2.colidx, T.colidx) and torch.equal(T2.vals, T.vals)


@pytest.mark.parametrize("n,nmat", [(1, 3), (7, 5), (64, 40), (71, 192), (128, 9)])
def test_batched_inverse(n, nmat):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((nmat, n, n))
    A[0] = np.eye(n)[::-1]                         # a permutation: every pivot needs a row swap
    if nmat > 1 and n > 2:
        A[1, 0, 0] = 0.0                           # zero leading entry
    M = ops.batched_inverse(dev(A))
    assert M is not None
    got = M.cpu().numpy()
    for k in range(nmat):
        assert np.abs(got[k] @ A[k] - np.eye(n)).max() < 1e-9 * max(1.0, np.linalg.cond(A[k]))
    np.testing.assert_allclose(got, np.linalg.inv(A), rtol=1e-7, atol=1e-9)
    S = A.copy()
    S[nmat // 2] = 0.0
    assert ops.batched_inverse(dev(S)) is None     # exactly singular: reported, not returned


def test_vector_ops():
    rng = np.random.default_rng(10)
    for n in (1, 2, 3, 1000, 1 << 20 | 1):
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        dy = dev(y.copy())
        ops.axpby(0.75, dev(x), -1.5, dy)
        assert np.array_equal(dy.cpu().numpy(), 0.75 * x + -1.5 * y)
        dy = dev(y.copy())
        ops.axpby(2.0, dev(x), 0.0, dy)
        assert np.array_equal(dy.cpu().numpy(), 2.0 * x)
        part = torch.empty(ops.partials_count(n), dtype=torch.float64, device=DEV)
        out = torch.empty(1, dtype=torch.float64, device=DEV)
        ops.dot(dev(x), dev(y), part, out)
        assert abs(out.item() - float(x @ y)) <= 1e-13 * float(np.abs(x) @ np.abs(y))
        dz = torch.empty(n, dtype=torch.float64, device=DEV)
        ops.copy(dev(x), dz)
        assert np.array_equal(dz.cpu().numpy(), x)
        ops.zero(dz)
        assert not dz.any()
    idx = rng.permutation(5000)[:1234].astype(np.int32)
    x = rng.standard_normal(5000)
    buf = torch.empty(1234, dtype=torch.float64, device=DEV)
    ops.gather(dev(idx), dev(x), buf)
    assert np.array_equal(buf.cpu().numpy(), x[idx])
    tgt = dev(np.zeros(5000))
    ops.scatter(dev(idx), buf, tgt)
    w = np.zeros(5000)
    w[idx] = x[idx]
    assert np.array_equal(tgt.cpu().numpy(), w)


@pytest.mark.parametrize("n,m", [(1, 1), (7, 5), (300, 300), (513, 513), (1000, 1024), (2967, 2967), (3, 4097), (9000, 1030)])
def test_dense_gemv(n, m):
    rng = np.random.default_rng(11)
    M, x = rng.standard_normal((n, m)), rng.standard_normal(m)
    y = torch.empty(n, dtype=torch.float64, device=DEV)
    ops.dense_gemv(dev(M), dev(x), y)
    want = K.dense_gemv(M, x)
    scale = np.abs(M) @ np.abs(x)
    assert np.all(np.abs(y.cpu().numpy() - want) <= 1e-14 * scale)


@pytest.mark.parametrize("n", [0, 1, 5, 4096, 4097, 100000, 16785409 // 4])
def test_exclusive_scan_bit_exact(n):
    rng = np.random.default_rng(12)
    a = rng.integers(0, 30, size=n).astype(np.int32)
    out = torch.empty(n + 1, dtype=torch.int32, device=DEV)
    ops.exclusive_scan_i32(dev(a) if n else torch.empty(0, dtype=torch.int32, device=DEV), out)
    want = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(a, out=want[1:])
    assert np.array_equal(out.cpu().numpy().astype(np.int64), want)


SPGEMM = ["RA_2d", "RA_P_2d", "QtA_l2like", "QtAQ_l2like", "QtA_level1_l2like", "RA_1d", "ragged_sq",
          "ragged_medium", "ragged_large", "empty_rows"]


@functools.lru_cache(maxsize=None)
def spgemm_case(name):
    if name in ("RA_2d", "RA_P_2d", "QtA_l2like", "QtAQ_l2like"):
        A2, _ = P.poisson_2d_structured(64)
        if name.startswith("RA"):
            Pm = P.tensor_interpolator_2d(65)
        else:
            Pm = P.learned_like(sp.kron(P.pseudo_l2_interpolator_1d(65),
                                        P.pseudo_l2_interpolator_1d(65)).tocsr(), 43)
        R = Pm.T.tocsr()
        return (R, A2) if name in ("RA_2d", "QtA_l2like") else (sp.csr_matrix(R @ A2), Pm)
    if name == "QtA_level1_l2like":              # 25-entry R rows x 25-entry Galerkin rows: ~600 products per row
        A2, _ = P.poisson_2d_structured(64)
        Q0 = P.learned_like(sp.kron(P.pseudo_l2_interpolator_1d(65), P.pseudo_l2_interpolator_1d(65)).tocsr(), 43)
        A1 = sp.csr_matrix(Q0.T @ A2 @ Q0)
        Q1 = P.learned_like(sp.kron(P.pseudo_l2_interpolator_1d(33), P.pseudo_l2_interpolator_1d(33)).tocsr(), 44)
        return Q1.T.tocsr(), A1
    if name == "RA_1d":
        return P.geometric_interpolator_1d(1025).T.tocsr(), P.poisson_1d_fd(1024)[0]
    if name == "ragged_sq":                      # one row far beyond the capacity limit
        rag = ragged_matrix(3000, 3, long_row=17)
        return rag, rag
    if name == "ragged_medium":                  # ~2000 products in one row: 256-thread class
        med = ragged_matrix(3000, 4, long_row=29, long_len=300)
        return med, med
    if name == "ragged_large":                   # ~4000 products in one row: the 8192-slot class (sort and replay)
        big = ragged_matrix(3000, 5, long_row=31, long_len=700)
        return big, big
    if name == "empty_rows":
        return sp.csr_matrix((5, 4)), sp.csr_matrix((4, 6))
    raise KeyError(name)


@pytest.mark.parametrize("name", SPGEMM)
def test_spgemm_matches_scipy_bit_exact(name):
    A, B = (K.as_csr(m) for m in spgemm_case(name))
    dA, dB = ops.DeviceCSR.from_scipy(A, DEV), ops.DeviceCSR.from_scipy(B, DEV)
    plan = ops.SpGEMMPlan(dA, dB)
    if name == "ragged_sq":
        # one row needs ~18000 products > LMG_SPGEMM_MAX_ROW_PRODUCTS: dense-accumulator path
        assert plan.max_products > 8192 and plan.long_rows is not None and plan.long_rows.numel() >= 1
    if name == "ragged_medium":
        assert 1024 < plan.max_products <= 8192               # exercises the 256-thread class
    if name == "ragged_large":
        assert 2048 < plan.max_products <= 8192, plan.max_products   # replay: the 8192-slot class
    C = plan.numeric(dA, dB).to_scipy()
    want = sp.csr_matrix(A @ B)
    want.sort_indices()
    assert C.shape == want.shape
    Cs = C.copy()
    Cs.sort_indices()
    assert np.array_equal(Cs.indices, C.indices)               # rows come out sorted
    Cs.sum_duplicates()
    assert Cs.nnz == C.nnz                                     # ... and duplicate-free
    # structural pattern of C contains SciPy's (SciPy drops entries that cancel to 0.0)
    assert abs(C - want).max() == 0.0, name
    Cz = C.copy()
    Cz.eliminate_zeros()
    wz = want.copy()
    wz.eliminate_zeros()
    assert np.array_equal(Cz.indptr, wz.indptr) and np.array_equal(Cz.indices, wz.indices)
    assert np.array_equal(Cz.data, wz.data)
    # numeric-only rebuild with new values of A, same pattern
    A2 = A.copy()
    A2.data = A2.data * 1.5 + 0.25
    C2 = plan.numeric(ops.DeviceCSR.from_scipy(A2, DEV, canonical=False), dB).to_scipy()
    w2 = sp.csr_matrix(A2 @ B)
    assert abs(C2 - w2).max() == 0.0
    # that second run recorded the product map (default "lazy"); later runs replay it without
    # sorting and must give the same bits for new values of A and of B, with and without `out`
    if A.nnz and B.nnz:
        assert plan.recorded_bytes() > 0
    rng = np.random.default_rng(77)
    out = None
    for trial in range(3):
        A3, B3 = A.copy(), B.copy()
        A3.data = rng.standard_normal(A.nnz)
        B3.data = rng.standard_normal(B.nnz)
        dA3, dB3 = ops.DeviceCSR.from_scipy(A3, DEV, canonical=False), ops.DeviceCSR.from_scipy(B3, DEV, canonical=False)
        out = plan.numeric(dA3, dB3, out=out if trial else None)
        C3 = out.to_scipy()
        fresh = ops.SpGEMMPlan(dA3, dB3, record=False).numeric(dA3, dB3).to_scipy()
        assert np.array_equal(C3.indptr, fresh.indptr) and np.array_equal(C3.indices, fresh.indices)
        assert np.array_equal(C3.data, fresh.data), (name, trial)
        assert abs(C3 - sp.csr_matrix(A3 @ B3)).max() == 0.0
    # record=True records in the very first run; record=False never does
    eager = ops.SpGEMMPlan(dA, dB, record=True)
    C4 = eager.numeric(dA, dB).to_scipy()
    assert np.array_equal(C4.data, C.data) and np.array_equal(C4.indices, C.indices)
    C5 = eager.numeric(dA, dB).to_scipy()
    assert np.array_equal(C5.data, C.data) and np.array_equal(C5.indices, C.indices)
    never = ops.SpGEMMPlan(dA, dB, record=False)
    never.numeric(dA, dB), never.numeric(dA, dB), never.numeric(dA, dB)
    assert never.recorded_bytes() == 0


def test_spgemm_dense_operands_take_the_long_row_path():
    """Dense-ish transfer operators (the reference scripts pass dense ndarrays as Q): every row
    of R*A and (R*A)*P needs far more than 8192 products."""
    rng = np.random.default_rng(31)
    A = sp.csr_matrix(rng.standard_normal((40, 300)))
    B = sp.csr_matrix(rng.standard_normal((300, 150)))
    dA, dB = ops.DeviceCSR.from_scipy(A, DEV), ops.DeviceCSR.from_scipy(B, DEV)
    plan = ops.SpGEMMPlan(dA, dB)
    assert plan.long_rows.numel() == 40 and plan.c_nnz == 40 * 150
    C = plan.numeric(dA, dB).to_scipy()
    want = sp.csr_matrix(A @ B)
    want.sort_indices()
    assert np.array_equal(C.indices, want.indices) and np.array_equal(C.indptr, want.indptr)
    assert np.array_equal(C.data, want.data)                 # same products, same order


def test_graph_capture_replays_a_sweep_sequence():
    A, b = P.poisson_2d_structured(128)
    A = K.as_csr(A)
    n = A.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    db = dev(b.ravel())
    x0 = torch.zeros(n, dtype=torch.float64, device=DEV)
    x1 = torch.empty_like(x0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = ops.CapturedGraph()
        with g:
            ops.csr_jacobi(dA, x0, db, 0.8, x1)
            ops.csr_jacobi(dA, x1, db, 0.8, x0)
        for _ in range(3):
            g.launch()
    st.synchronize()
    want = np.zeros(n)
    for _ in range(6):
        want = K.jacobi(A, want, b, 0.8)
    assert np.array_equal(x0.cpu().numpy(), want)


def test_torch_custom_ops_registered():
    ops.register_torch_ops()
    A, b = P.poisson_2d_structured(32)
    A = K.as_csr(A)
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    x = dev(np.linspace(0, 1, A.shape[0]))
    out = torch.ops.lmg.csr_jacobi(dA.rowptr, dA.colidx, dA.vals, x, dev(b.ravel()), 1.0)
    assert np.array_equal(out.cpu().numpy(), K.jacobi(A, x.cpu().numpy(), b, 1.0))
    r, n2 = torch.ops.lmg.csr_residual(dA.rowptr, dA.colidx, dA.vals, x, dev(b.ravel()))
    wr, wn2 = K.residual(A, x.cpu().numpy(), b)
    assert np.array_equal(r.cpu().numpy(), wr)
    y = torch.ops.lmg.csr_spmv(dA.rowptr, dA.colidx, dA.vals, A.shape[1], x)
    assert np.array_equal(y.cpu().numpy(), A @ x.cpu().numpy())
    # Galerkin product, transpose and coarse GEMV as custom ops
    Pm = K.as_csr(P.tensor_interpolator_2d(33))
    dP = ops.DeviceCSR.from_scipy(Pm, DEV)
    trp, tci, tva = torch.ops.lmg.csr_transpose(dP.rowptr, dP.colidx, dP.vals, Pm.shape[1])
    Rm = sp.csr_matrix(Pm.T)
    Rm.sort_indices()
    assert np.array_equal(tci.cpu().numpy(), Rm.indices) and np.array_equal(tva.cpu().numpy(), Rm.data)
    crp, cci, cva = torch.ops.lmg.spgemm(trp, tci, tva, Pm.shape[0], dA.rowptr, dA.colidx, dA.vals, A.shape[1])
    C = sp.csr_matrix((cva.cpu().numpy(), cci.cpu().numpy(), crp.cpu().numpy()), shape=(Pm.shape[1], A.shape[1]))
    assert C.has_sorted_indices and abs(C - Rm @ A).max() == 0        # (SciPy drops products that cancel to 0.0)
    Md = np.random.default_rng(5).standard_normal((40, A.shape[0]))
    np.testing.assert_allclose(torch.ops.lmg.dense_gemv(dev(Md), x).cpu().numpy(), Md @ x.cpu().numpy(), rtol=1e-13)
    # operators with their lossless twins behind a handle: sweeps, fused sweeps, exact Gauss-Seidel
    h = torch.ops.lmg.operator_create(dA.rowptr, dA.colidx, dA.vals, A.shape[1])
    assert torch.ops.lmg.operator_format(h) == "stencil"
    xn, bn = x.cpu().numpy(), b.ravel()
    assert np.array_equal(torch.ops.lmg.operator_spmv(h, x).cpu().numpy(), K.spmv(A, xn, np.zeros_like(xn), 1.0, 0.0))
    r2, _ = torch.ops.lmg.operator_residual(h, x, dev(bn))
    assert np.array_equal(r2.cpu().numpy(), wr)
    want = xn
    for _ in range(5):
        want = K.jacobi(A, want, bn, 0.8)
    min_rows = ops.FUSED_MIN_ROWS
    try:
        for ops.FUSED_MIN_ROWS in (min_rows, 0):                     # one launch per sweep / fused passes
            got = torch.ops.lmg.operator_jacobi(h, x, dev(bn), 0.8, 5)
            assert np.array_equal(got.cpu().numpy(), want)
    finally:
        ops.FUSED_MIN_ROWS = min_rows
    xg = x.clone()
    torch.ops.lmg.operator_gauss_seidel_(h, xg, dev(bn), 2)
    wg = xn.copy()
    K.gs_forward(A, wg, bn, 2)
    assert np.array_equal(xg.cpu().numpy(), wg)
    torch.ops.lmg.operator_free(h)
    with pytest.raises(Exception):
        torch.ops.lmg.operator_spmv(h, x)


# ---- packed CSR (lmg_pcsr_sweep): same results, bit for bit, in every encoding -------------------
@functools.lru_cache(maxsize=None)
def packed_case(name):
    if name == "val8_col16_poisson2d_513":
        return case("poisson2d_513")
    if name == "val8_col16_prolong":
        return case("prolong_65")
    if name == "val8_col16_restrict":
        return case("restrict_65")
    if name == "val8_col16_poisson1d":
        return case("poisson1d_1025")
    if name == "val16_col16_ragged":
        return case("ragged_1000")
    if name == "val64_col16_jittered":
        return K.as_csr(P.jittered_poisson_2d(300, seed=7)[0])
    if name in ("val64_col32_random", "val16_col32_random"):
        # (sp.random would build a permutation of n*n indices: sample coordinates directly)
        n = 150000 if name.startswith("val64") else 120000
        rng = np.random.default_rng(11)
        k = 6 * n
        r, c = rng.integers(0, n, k), rng.integers(0, n, k)
        v = rng.standard_normal(k) if name.startswith("val64") else rng.integers(1, 3000, k).astype(np.float64)
        A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr() + sp.identity(n) * 3.0
        return K.as_csr(A)
    if name in ("val64_longrows_l2_galerkin", "val8_longrows_l2_galerkin"):
        A2, _ = P.poisson_2d_structured(256)
        l2 = P.pseudo_l2_interpolator_1d(257)
        Q = sp.kron(l2, l2).tocsr()
        if name.startswith("val64"):
            Q = P.learned_like(Q, 43)
        else:
            # exactly representable weights -> few distinct Galerkin values (dictionary encoding)
            Q.data = np.round(Q.data * 64) / 64
        return K.as_csr(sp.csr_matrix(Q.T @ A2 @ Q))
    if name == "galerkin_9pt":
        A2, _ = P.poisson_2d_structured(256)
        Pm = P.tensor_interpolator_2d(257)
        return K.as_csr(sp.csr_matrix(Pm.T @ A2 @ Pm))
    raise KeyError(name)


PACKED = ["val64_longrows_l2_galerkin", "val8_longrows_l2_galerkin", "val8_col16_poisson2d_513", "val8_col16_prolong", "val8_col16_restrict", "val8_col16_poisson1d",
          "val16_col16_ragged", "val64_col16_jittered", "val64_col32_random", "val16_col32_random",
          "galerkin_9pt"]


@pytest.mark.parametrize("name", PACKED)
def test_packed_sweeps_bit_exact(name):
    A = packed_case(name)
    n, m = A.shape
    rng = np.random.default_rng(21)
    x, b, y0 = rng.standard_normal(m), rng.standard_normal(n), rng.standard_normal(n)
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    ops.set_sell_enabled(False)                  # this test is about the packed CSR kernels
    try:
        Pk = dA.pack(patterns=False)
    finally:
        ops.set_sell_enabled(True)
    assert Pk is not None and dA.patterns is None and dA.sell is None
    want_modes = {"val8": 0, "val16": 1, "val64": 2}
    for key, vm in want_modes.items():
        if name.startswith(key):
            assert Pk.valmode == vm, (name, Pk.valmode, Pk.ndict)
    if "_col16_" in name:
        assert Pk.colmode == 0
    if "_col32_" in name:
        assert Pk.colmode == 1
    assert Pk.bytes() < dA.bytes()
    if "longrows" in name:
        assert Pk.tile_rows in (64, 128) and np.diff(A.indptr).max() >= 20
    elif name.startswith("val8"):
        assert Pk.tile_rows == 512
    else:
        assert Pk.tile_rows in (512, 128, 64)
    try:
        ops.set_packed_enabled(True)
        for alpha, beta in ((1.0, 0.0), (1.0, 1.0), (-0.5, 2.0)):
            y = dev(y0.copy())
            ops.csr_spmv(dA, dev(x), y, alpha, beta)
            assert np.array_equal(y.cpu().numpy(), K.spmv(A, x, y0, alpha, beta)), (name, alpha, beta)
        if n == m:
            r = torch.empty(n, dtype=torch.float64, device=DEV)
            part = torch.empty(ops.partials_count(n), dtype=torch.float64, device=DEV)
            n2 = torch.zeros(1, dtype=torch.float64, device=DEV)
            ops.csr_residual_norm2(dA, dev(x), dev(b), r, part, n2)
            wr, wn2 = K.residual(A, x, b)
            assert np.array_equal(r.cpu().numpy(), wr)
            assert abs(n2.item() - wn2) <= 1e-13 * wn2
            n2b = torch.zeros(1, dtype=torch.float64, device=DEV)
            ops.csr_residual_norm2(dA, dev(x), dev(b), None, part, n2b)
            assert n2b.item() == n2.item()
            for omega in (1.0, 0.8):
                out = torch.empty(n, dtype=torch.float64, device=DEV)
                ops.csr_jacobi(dA, dev(x), dev(b), omega, out)
                assert np.array_equal(out.cpu().numpy(), K.jacobi(A, x, b, omega)), (name, omega)
            # and the packed path really is a different kernel from the plain one
            ops.set_packed_enabled(False)
            out2 = torch.empty(n, dtype=torch.float64, device=DEV)
            ops.csr_jacobi(dA, dev(x), dev(b), 0.8, out2)
            assert torch.equal(out, out2)
    finally:
        ops.set_packed_enabled(True)


def test_packing_is_refused_for_rows_longer_than_255_and_preserves_signed_zero():
    A = case("ragged_longrow_6007")
    assert ops.DeviceCSR.from_scipy(A, DEV).pack() is None
    # -0.0 and +0.0 are different dictionary entries (bitwise dictionary)
    M = sp.csr_matrix((np.array([0.0, -0.0, 1.0, 2.0]), np.array([0, 1, 0, 1]), np.array([0, 2, 4])), shape=(2, 2))
    dM = ops.DeviceCSR(dev(M.indptr.astype(np.int32)), dev(M.indices.astype(np.int32)), dev(M.data), (2, 2))
    Pk = dM.pack(patterns=False)
    assert Pk.ndict == 4
    x = np.array([-1.0, 1.0])
    y = torch.empty(2, dtype=torch.float64, device=DEV)
    ops.csr_spmv(dM, dev(x), y, 1.0, 0.0)
    got = y.cpu().numpy()
    want = K.lib()  # noqa: F841
    ref = np.empty(2)
    K.lib().orc_csr_matvec(2, M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data, x, ref)
    assert np.array_equal(np.signbit(got), np.signbit(ref)) and np.array_equal(got, ref)


def _numpy_pack(A, T):
    """The packed format restated with numpy (what lmg.h documents), to check the device build."""
    n = A.shape[0]
    rp, ci, va = A.indptr, A.indices, A.data
    ntile = (n + T - 1) // T
    base = np.array([rp[min(t * T, n)] for t in range(ntile + 1)])
    cmin = np.zeros(ntile, dtype=np.int64)
    cmax = np.zeros(ntile, dtype=np.int64)
    for t in range(ntile):
        seg = ci[base[t]:base[t + 1]]
        if seg.size:
            cmin[t], cmax[t] = seg.min(), seg.max()
    rel = ci - np.repeat(cmin, np.diff(base))
    uniq = np.unique(va.view(np.int64))
    return base, cmin, cmax, rel, uniq, np.searchsorted(uniq, va.view(np.int64))


@pytest.mark.parametrize("name", ["val8_col16_poisson2d_513", "val16_col16_ragged", "val16_col32_random",
                                  "val8_longrows_l2_galerkin", "galerkin_9pt", "val8_col16_restrict"])
def test_packed_format_matches_its_numpy_restatement(name):
    A = packed_case(name)
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    ops.set_sell_enabled(False)
    try:
        Pk = dA.pack(patterns=False)
    finally:
        ops.set_sell_enabled(True)
    base, cmin, cmax, rel, uniq, idx = _numpy_pack(A, Pk.tile_rows)
    nnz = A.nnz
    assert np.array_equal(Pk.tile_base.cpu().numpy(), base)
    assert np.array_equal(Pk.tile_colbase.cpu().numpy(), cmin)
    assert np.array_equal(Pk.rowlen.cpu().numpy(), np.diff(A.indptr))
    assert Pk.tile_cap == np.diff(base).max()
    if Pk.colmode == 0:
        assert (cmax - cmin).max() < 65536
        assert np.array_equal(Pk.col[:2 * nnz].view(torch.int16).cpu().numpy().view(np.uint16), rel)
    else:
        assert (cmax - cmin).max() >= 65536
        assert np.array_equal(Pk.col[:4 * nnz].view(torch.int32).cpu().numpy(), A.indices)
    if Pk.valmode == 2:
        assert np.array_equal(Pk.val[:8 * nnz].view(torch.float64).cpu().numpy(), A.data)
    else:
        assert Pk.ndict == uniq.size
        assert np.array_equal(Pk.dict.view(torch.int64).cpu().numpy(), uniq)
        if Pk.valmode == 0:
            assert np.array_equal(Pk.val[:nnz].cpu().numpy(), idx)
        else:
            assert np.array_equal(Pk.val[:2 * nnz].view(torch.int16).cpu().numpy().view(np.uint16), idx)
    # padding: 16-byte aligned streams with >= 16 readable bytes behind the data
    for t, used in ((Pk.col, nnz * (2, 4)[Pk.colmode]), (Pk.val, nnz * (1, 2, 8)[Pk.valmode])):
        assert t.data_ptr() % 16 == 0 and t.numel() >= used + 16


def test_distinct_value_set_limits_and_special_bit_patterns():
    rng = np.random.default_rng(5)
    PC = ops.PackedCSR
    # exactly at the limit, one above it, and far above it (early exit)
    for count, want in ((65536, 65536), (65537, None), (3_000_000, None)):
        v = np.arange(1, count + 1, dtype=np.float64)
        v = np.concatenate([v, v[rng.integers(0, count, 100000)]])
        rng.shuffle(v)
        u = PC._distinct_values(dev(v))
        if want is None:
            assert u is None
        else:
            assert np.array_equal(u.cpu().numpy(), np.unique(v.view(np.int64)))
    # NaN payloads (including the all-ones pattern the table uses as its empty marker),
    # infinities, signed zeros, denormals: compared as bit patterns
    special = np.array([0x7FF8000000000000, 0xFFFFFFFFFFFFFFFF, 0x7FF8000000000001, 0x7FF0000000000000,
                        0xFFF0000000000000, 0x0000000000000000, 0x8000000000000000, 0x0000000000000001],
                       dtype=np.uint64).view(np.float64)
    v = special[rng.integers(0, special.size, 50000)]
    v[:special.size] = special
    u = PC._distinct_values(dev(v))
    assert np.array_equal(u.cpu().numpy(), np.unique(special.view(np.int64)))
    out = torch.zeros(v.size + 32, dtype=torch.uint8, device=DEV)
    PC._encode_values(dev(v), u, 1, out)
    assert np.array_equal(out[:v.size].cpu().numpy(), np.searchsorted(np.unique(special.view(np.int64)), v.view(np.int64)))
    # a dictionary that misses a value is reported, not silently mis-encoded
    with pytest.raises(ops.LmgError):
        PC._encode_values(dev(v), u[:-1].contiguous(), 1, out)
    # many distinct values, dictionary too large for LDS (global binary search)
    v = rng.integers(0, 40000, 500000).astype(np.float64) * 0.25 - 1000.0
    u = PC._distinct_values(dev(v))
    un = np.unique(v.view(np.int64))
    assert np.array_equal(u.cpu().numpy(), un)
    out = torch.zeros(2 * v.size + 32, dtype=torch.uint8, device=DEV)
    PC._encode_values(dev(v), u, 2, out)
    assert np.array_equal(out[:2 * v.size].view(torch.int16).cpu().numpy().view(np.uint16), np.searchsorted(un, v.view(np.int64)))


def test_update_values_reencodes_or_asks_for_a_repack():
    A = packed_case("val8_col16_poisson2d_513")
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    Pk = dA.pack(patterns=False)
    assert Pk.valmode == 0
    x = np.random.default_rng(3).standard_normal(A.shape[0])
    B = A.copy()
    B.data = np.where(B.data > 0, B.data * 3.0, B.data * 0.5)
    dA.vals.copy_(dev(B.data))
    dA.repack_values()
    assert dA.packed is Pk and Pk.valmode == 0
    y = torch.empty(A.shape[0], dtype=torch.float64, device=DEV)
    ops.csr_spmv(dA, dev(x), y)
    assert np.array_equal(y.cpu().numpy(), K.spmv(B, x, np.zeros(A.shape[0]), 1.0, 0.0))
    # values that no longer fit a uint8 dictionary: full repack with another encoding
    B.data = np.random.default_rng(4).standard_normal(B.nnz)
    dA.vals.copy_(dev(B.data))
    dA.repack_values()
    assert dA.packed is not Pk and dA.packed.valmode == 2
    ops.csr_spmv(dA, dev(x), y)
    assert np.array_equal(y.cpu().numpy(), K.spmv(B, x, np.zeros(A.shape[0]), 1.0, 0.0))


def test_inverse_diagonal_kernel():
    rng = np.random.default_rng(9)
    n = 5000
    r, c = rng.integers(0, n, 50000), rng.integers(0, n, 50000)
    off = sp.coo_matrix((rng.standard_normal(r.size)[r != c], (r[r != c], c[r != c])), shape=(n, n))
    A = K.as_csr((off + sp.diags(rng.standard_normal(n))).tocsr())
    # rows without a diagonal entry, with an explicit zero, and with duplicate diagonal entries
    rp, ci, va = A.indptr.copy(), A.indices.copy(), A.data.copy()
    for i in (3, 77):
        seg = slice(rp[i], rp[i + 1])
        va[seg] = np.where(ci[seg] == i, 0.0, va[seg])
    extra_r, extra_c, extra_v = np.array([10, 10, 4999]), np.array([10, 10, 4999]), np.array([0.5, -0.25, 2.0])
    rows = np.repeat(np.arange(n), np.diff(rp))
    rows, cols, vals = np.concatenate([rows, extra_r]), np.concatenate([ci, extra_c]), np.concatenate([va, extra_v])
    keep = ~((rows == 200) & (cols == 200))
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    order = np.argsort(rows, kind="stable")              # duplicates stay unmerged, storage order kept
    rows, cols, vals = rows[order], cols[order], vals[order]
    rp2 = np.zeros(n + 1, dtype=np.int32)
    np.add.at(rp2, rows + 1, 1)
    rp2 = np.cumsum(rp2).astype(np.int32)
    dA = ops.DeviceCSR(dev(rp2), dev(cols.astype(np.int32)), dev(vals), (n, n))
    got = ops.csr_inverse_diagonal(dA).cpu().numpy()
    want = np.zeros(n)
    for i in range(n):
        d = 0.0
        for e in range(rp2[i], rp2[i + 1]):
            if cols[e] == i:
                d += vals[e]
        want[i] = 1.0 / d if d != 0.0 else 0.0
    assert np.array_equal(got, want)
    assert want[3] == 0.0 and want[77] == 0.0 and want[200] == 0.0


# ---- row-pattern twin (rpat.hip) ---------------------------------------------------------------
def _numpy_row_patterns(A):
    """Distinct rows of A written as (length; column - row, value bits ...), with numpy."""
    keys = {}
    ids = np.empty(A.shape[0], dtype=np.int64)
    for i in range(A.shape[0]):
        s, e = A.indptr[i], A.indptr[i + 1]
        k = (tuple((A.indices[s:e] - i).tolist()), A.data[s:e].tobytes())
        ids[i] = keys.setdefault(k, len(keys))
    return keys, ids


@functools.lru_cache(maxsize=None)
def rpat_case(name):
    if name == "poisson2d_129":
        return K.as_csr(P.poisson_2d_structured(128)[0])
    if name == "poisson1d":
        return K.as_csr(P.poisson_1d_fd(3000)[0])
    if name == "galerkin_9pt":
        return packed_case("galerkin_9pt")
    if name == "galerkin_l2_rounded":                 # 25-entry rows, few patterns
        return packed_case("val8_longrows_l2_galerkin")
    if name == "with_empty_and_diagless_rows":
        A = P.poisson_2d_structured(40)[0].tolil()
        A[5, :] = 0.0                                 # empty row
        A[77, 77] = 0.0                               # row without a diagonal entry
        A = sp.csr_matrix(A)
        A.eliminate_zeros()
        return K.as_csr(A)
    if name == "duplicate_diagonal":                  # non-canonical: two diagonal entries per row
        n = 500
        rp = np.arange(0, 3 * n + 1, 3, dtype=np.int32)
        ci = np.stack([np.arange(n), np.arange(n), (np.arange(n) + 1) % n], 1).astype(np.int32).ravel()
        va = np.tile(np.array([2.0, 0.5, -1.0]), n)
        return sp.csr_matrix((va, ci, rp), shape=(n, n))
    raise KeyError(name)


@pytest.mark.parametrize("name", ["poisson2d_129", "poisson1d", "galerkin_9pt", "galerkin_l2_rounded",
                                  "with_empty_and_diagless_rows", "duplicate_diagonal"])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, "stencil"])
@pytest.mark.parametrize("nt", [False, True], ids=["cached", "nontemporal"])
def test_row_pattern_sweeps_bit_exact(name, variant, nt):
    """nt=True forces the NT = true instantiations (nontemporal id / b loads and out stores) that the
    launchers otherwise only pick from 8 M rows on -- the ones the headline bench times.
    variant "stencil": the grid-stencil kernel (stencil.hip) on the same twin, for the matrices whose
    patterns are 3x3 stencils; variants 0..4: the tile geometries of the row-pattern kernel (rpat.hip)."""
    stencil = variant == "stencil"
    eligible = name in ("poisson2d_129", "poisson1d", "galerkin_9pt", "with_empty_and_diagless_rows")
    if stencil and not eligible:
        # 25-entry rows / unsorted duplicate entries are no 3x3 stencils: the builder must refuse them
        A = rpat_case(name)
        dA = (ops.DeviceCSR(dev(A.indptr.astype(np.int32)), dev(A.indices.astype(np.int32)), dev(A.data), A.shape)
              if name == "duplicate_diagonal" else ops.DeviceCSR.from_scipy(A, DEV))
        assert isinstance(dA.pack(), ops.RowPatterns) and dA.stencil is None
        return
    A = rpat_case(name)
    n = A.shape[0]
    keys, ids = _numpy_row_patterns(A)
    if name == "duplicate_diagonal":
        dA = ops.DeviceCSR(dev(A.indptr.astype(np.int32)), dev(A.indices.astype(np.int32)), dev(A.data), A.shape)
    else:
        dA = ops.DeviceCSR.from_scipy(A, DEV)
    R = dA.pack()
    assert isinstance(R, ops.RowPatterns) and dA.packed is None
    assert (dA.stencil is not None) == eligible
    if eligible:
        S = dA.stencil
        expect_W = {"poisson2d_129": 129, "galerkin_9pt": None, "with_empty_and_diagless_rows": 41}.get(name)
        if expect_W:
            assert S.W == expect_W
        assert S.pid is R.pid and S.npat == R.npat
        assert bool(S.umask & 0x145) == (name == "galerkin_9pt")          # corner slots: 9-point only
        assert bool(S.umask & 0x1C7) == (name != "poisson1d")
    assert R.npat == len(keys) and R.nent == sum(len(k[0]) for k in keys) and R.max_len == max(len(k[0]) for k in keys)
    # same partition of the rows into patterns, and every pattern reproduces its rows
    pid = R.pid.cpu().numpy()
    assert len(set(zip(pid.tolist(), ids.tolist()))) == len(keys)
    ptr, off, val = R.pat_ptr.cpu().numpy(), R.pat_off.cpu().numpy(), R.pat_val.cpu().numpy()
    for i in (0, 1, n // 3, n // 2, n - 2, n - 1):
        s, e = A.indptr[i], A.indptr[i + 1]
        p = pid[i]
        assert np.array_equal(off[ptr[p]:ptr[p + 1]], A.indices[s:e] - i)
        assert np.array_equal(val[ptr[p]:ptr[p + 1]].view(np.int64), A.data[s:e].view(np.int64))
    assert R.bytes() < 0.2 * dA.bytes()
    rng = np.random.default_rng(23)
    x, b, y0 = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    # oracle kernels on the RAW arrays (K.spmv & co. would merge the duplicate entries first)
    rp_, ci_, va_ = A.indptr.astype(np.int32), A.indices.astype(np.int32), np.ascontiguousarray(A.data)

    class Raw:
        @staticmethod
        def spmv(_, x_, y_, al, be):
            o = y_.copy()
            K.lib().orc_csr_spmv(n, rp_, ci_, va_, x_, o, al, be)
            return o

        @staticmethod
        def residual(_, x_, b_):
            o = np.empty(n)
            return o, K.lib().orc_csr_residual(n, rp_, ci_, va_, x_, b_, o)

        @staticmethod
        def jacobi(_, x_, b_, om):
            o = np.empty(n)
            K.lib().orc_csr_jacobi(n, rp_, ci_, va_, x_, b_, om, o)
            return o
    Ac = A
    nt_default = ops.tune_get("rpat_nt_rows")
    assert nt_default == 1 << 23 and ops.tune_get("stencil_nt_rows") == 1 << 23
    try:
        ops.set_stencil_enabled(stencil)
        if not stencil:
            ops.tune_set("rpat_variant", variant)
        if nt:
            ops.tune_set("rpat_nt_rows", 1)
            ops.tune_set("stencil_nt_rows", 1)
            assert ops.tune_get("rpat_nt_rows") == 1 and ops.tune_get("stencil_nt_rows") == 1
        for alpha, beta in ((1.0, 0.0), (1.0, 1.0), (-0.5, 2.0)):
            y = dev(y0.copy())
            ops.csr_spmv(dA, dev(x), y, alpha, beta)
            assert np.array_equal(y.cpu().numpy(), Raw.spmv(Ac, x, y0, alpha, beta)), (name, alpha, beta)
        r = torch.empty(n, dtype=torch.float64, device=DEV)
        part = torch.empty(ops.partials_count(n), dtype=torch.float64, device=DEV)
        n2 = torch.zeros(1, dtype=torch.float64, device=DEV)
        ops.csr_residual_norm2(dA, dev(x), dev(b), r, part, n2)
        wr, wn2 = Raw.residual(Ac, x, b)
        assert np.array_equal(r.cpu().numpy(), wr)
        assert abs(n2.item() - wn2) <= 1e-13 * wn2
        n2b = torch.zeros(1, dtype=torch.float64, device=DEV)
        ops.csr_residual_norm2(dA, dev(x), dev(b), None, part, n2b)
        assert n2b.item() == n2.item()
        for omega in (1.0, 0.8):
            out = torch.empty(n, dtype=torch.float64, device=DEV)
            ops.csr_jacobi(dA, dev(x), dev(b), omega, out)
            assert np.array_equal(out.cpu().numpy(), Raw.jacobi(Ac, x, b, omega)), (name, omega)
        # ... and it really is another kernel than the CSR one
        ops.set_packed_enabled(False)
        out2 = torch.empty(n, dtype=torch.float64, device=DEV)
        ops.csr_jacobi(dA, dev(x), dev(b), 0.8, out2)
        assert torch.equal(out, out2)
    finally:
        ops.set_packed_enabled(True)
        ops.set_stencil_enabled(True)
        ops.tune_set("rpat_variant", 0)
        ops.tune_set("rpat_nt_rows", nt_default)
        ops.tune_set("stencil_nt_rows", nt_default)


@pytest.mark.parametrize("name", ["poisson2d_129", "poisson1d", "galerkin_9pt", "with_empty_and_diagless_rows",
                                  "poisson2d_300", "poisson2d_601", "poisson2d_602", "galerkin_9pt_515"])
@pytest.mark.parametrize("seg_lines,pf", [(0, 0), (4, 2), (7, 2), (1000, 2), ("tile", 16), ("tile", 32)])
def test_fused_stencil_smoothing_equals_separate_sweeps(name, seg_lines, pf):
    """lmg_stencil_smooth (S sweeps [+ residual] in one pass, iterates in registers) and lmg_stencil_smooth_tiled
    (the same with the iterates in LDS, what small levels run) against the oracle's separate Jacobi sweeps and
    residual, bitwise, for S = 1..3, zero / non-zero initial iterate, with and without the residual, few and many
    line segments per strip / 16- and 32-line tiles."""
    # (line strides from 512 on run the four-elements-per-lane march: 601 = 1, 602 = 2, 515 = 3 mod 4)
    if name == "poisson2d_300":
        A = K.as_csr(P.poisson_2d_structured(299)[0])
    elif name in ("poisson2d_601", "poisson2d_602"):
        A = K.as_csr(P.poisson_2d_structured(int(name[-3:]) - 1)[0])
    elif name == "galerkin_9pt_515":
        Pf = P.tensor_interpolator_2d(1029)
        A = K.as_csr(sp.csr_matrix(Pf.T @ P.poisson_2d_structured(1028)[0] @ Pf))
    else:
        A = rpat_case(name)
    n = A.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dA.pack()
    assert dA.stencil is not None
    tiled = seg_lines == "tile"
    if tiled and n > 200000:
        pytest.skip("the tiled pass is covered by the smaller cases")
    if tiled and not ops._lib.lib().lmg_stencil_smooth_tiled_supported(dA.stencil.umask):
        pytest.skip("1-D chains run the register kernel")
    rng = np.random.default_rng(77)
    x0, b = rng.standard_normal(n), rng.standard_normal(n)
    min_tiled = ops.TILED_MIN_ROWS
    tile_rows0 = ops.tune_get("tile_rows")
    try:
        if tiled:
            ops.TILED_MIN_ROWS = 0
            ops.tune_set("tile_rows", pf)
            assert ops._fused_kind(dA) == "tile"
            seg_lines = pf = 0
        else:
            ops.set_tiled_enabled(False)                   # small level: the product would pick the tiled pass
            assert ops._fused_kind(dA) is None
        ops.tune_set("fused_seg_lines", seg_lines)
        ops.tune_set("fused_pf", pf)
        for omega in (0.8, 1.0):
            for zero in (False, True):
                want = np.zeros(n) if zero else x0.copy()
                for S in (1, 2, 3):
                    want = K.jacobi(A, want, b, omega)
                    wr, _ = K.residual(A, want, b)
                    for resid in (False, True):
                        out = torch.full((n,), np.nan, dtype=torch.float64, device=DEV)
                        r = torch.full((n,), np.nan, dtype=torch.float64, device=DEV) if resid else None
                        ops.stencil_smooth(dA, None if zero else dev(x0), dev(b), omega, S, out, r)
                        got = out.cpu().numpy()
                        assert not np.isnan(got).any(), (name, S, zero, resid)
                        # (a zero iterate and b_i = -0.0 would differ in the sign of zero only; b is random)
                        assert np.array_equal(got, want), (name, omega, S, zero, resid, np.flatnonzero(got != want)[:8])
                        if resid:
                            assert np.array_equal(r.cpu().numpy(), wr), (name, omega, S, zero)
    finally:
        ops.tune_set("fused_seg_lines", 0)
        ops.tune_set("fused_pf", 0)
        ops.tune_set("tile_rows", tile_rows0)
        ops.set_tiled_enabled(True)
        ops.TILED_MIN_ROWS = min_tiled


@pytest.mark.parametrize("m,kind", [(32, "5pt"), (64, "5pt"), (150, "5pt"), (64, "9pt"), (129, "9pt"), (300, "5pt"), (301, "5pt"),
                                    (257, "9pt")])
@pytest.mark.parametrize("seg_lines", [0, 5, 1000, "tile"])
def test_fused_post_smoothing_with_the_correction_folded_in(m, kind, seg_lines):
    """lmg_stencil_smooth_prolong / lmg_stencil_smooth_tiled_prolong: x_out = J^S(x + P e) in one pass, against the
    oracle's prolongation (K.spmv alpha = beta = 1) followed by S separate Jacobi sweeps, bitwise; 5-point fine
    operators and 9-point Galerkin operators, tensor-product interpolation between (2m+1)^2 and (m+1)^2 nodes,
    strips / segments that start on odd and even lines and columns; the LDS-tiled pass ("tile": what these sizes
    run in the product) and the register-blocked one."""
    side = 2 * m + 1
    if kind == "5pt":
        A = K.as_csr(P.poisson_2d_structured(side - 1)[0])
    else:
        Af = P.poisson_2d_structured(2 * (side - 1))[0]
        Pf = P.tensor_interpolator_2d(2 * (side - 1) + 1)
        A = K.as_csr(sp.csr_matrix(Pf.T @ Af @ Pf))
    Pm = K.as_csr(sp.csr_matrix(P.tensor_interpolator_2d(side)))
    n, nc = A.shape[0], Pm.shape[1]
    assert Pm.shape[0] == n and nc == (m + 1) ** 2
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dA.pack()
    dP = ops.DeviceCSR.from_scipy(Pm, DEV)
    dP.pack()
    T = dP.prolong
    assert dA.stencil is not None and T is not None and (T.W, T.Wc, T.nc) == (side, m + 1, nc)
    assert T._hot_pairs[0] >= 0 and T._hot_pairs[1] >= 0
    rng = np.random.default_rng(5)
    x0, b, e = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(nc)
    tiled = seg_lines == "tile"
    try:
        if tiled:
            seg_lines = 0
            assert ops._fused_kind(dA) == "tile" and ops.stencil_smooth_prolong_available(dA, dP)
        else:
            ops.set_tiled_enabled(False)
            assert ops._fused_kind(dA) is None
        ops.tune_set("fused_seg_lines", seg_lines)
        for omega in (0.8, 1.0):
            want = K.spmv(Pm, e, x0.copy(), 1.0, 1.0)
            for S in (1, 2, 3):
                want = K.jacobi(A, want, b, omega)
                out = torch.full((n,), np.nan, dtype=torch.float64, device=DEV)
                ops.stencil_smooth(dA, dev(x0), dev(b), omega, S, out, None, prolong=(dP, dev(e)))
                got = out.cpu().numpy()
                assert not np.isnan(got).any(), (m, kind, S)
                assert np.array_equal(got, want), (m, kind, omega, S, np.flatnonzero(got != want)[:8])
        # without the frequent-pair shortcut (every line through the pattern table): same bits
        hot = (T._hot_pairs[0], T._hot_pairs[1])
        T._hot_pairs[0] = T._hot_pairs[1] = -1
        out = torch.full((n,), np.nan, dtype=torch.float64, device=DEV)
        ops.stencil_smooth(dA, dev(x0), dev(b), 1.0, 3, out, None, prolong=(dP, dev(e)))
        T._hot_pairs[0], T._hot_pairs[1] = hot
        assert np.array_equal(out.cpu().numpy(), want)
    finally:
        ops.tune_set("fused_seg_lines", 0)
        ops.set_tiled_enabled(True)


@pytest.mark.parametrize("m,kind", [(64, "5pt"), (150, "5pt"), (64, "9pt"), (129, "9pt"), (300, "5pt"), (301, "5pt"), (257, "9pt")])
@pytest.mark.parametrize("seg_lines", [0, 5, 6, 1000, "tile"])
def test_fused_pre_smoothing_with_the_restriction_folded_in(m, kind, seg_lines):
    """lmg_stencil_smooth_restrict: x_out = J^S(x), b_c = R (b - A x_out) in one pass without storing the residual,
    against the oracle's S Jacobi sweeps, residual and restriction (K.spmv with R = P^T), bitwise; zero and
    non-zero initial iterates; segments that start on odd and even lines."""
    side = 2 * m + 1
    if kind == "5pt":
        A = K.as_csr(P.poisson_2d_structured(side - 1)[0])
    else:
        Af = P.poisson_2d_structured(2 * (side - 1))[0]
        Pf = P.tensor_interpolator_2d(2 * (side - 1) + 1)
        A = K.as_csr(sp.csr_matrix(Pf.T @ Af @ Pf))
    Rm = K.as_csr(sp.csr_matrix(sp.csr_matrix(P.tensor_interpolator_2d(side)).T))
    n, nc = A.shape[0], Rm.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dA.pack()
    dR = ops.DeviceCSR.from_scipy(Rm, DEV)
    dR.pack()
    T = dR.restrict
    assert dA.stencil is not None and T is not None and (T.W, T.Wc, T.nc, T.n) == (side, m + 1, nc, n) and T.hot >= 0
    rng = np.random.default_rng(6)
    x0, b = rng.standard_normal(n), rng.standard_normal(n)
    try:
        if seg_lines == "tile":                            # the LDS-tiled pass: what these sizes run in the product
            seg_lines = 0
            assert ops._fused_kind(dA) == "tile" and ops.stencil_smooth_restrict_available(dA, dR)
        else:
            ops.set_tiled_enabled(False)
            assert ops._fused_kind(dA) is None
        ops.tune_set("fused_seg_lines", seg_lines)
        for zero in (False, True):
            want = np.zeros(n) if zero else x0.copy()
            for S in (1, 2, 3):
                want = K.jacobi(A, want, b, 0.8)
                wr, _ = K.residual(A, want, b)
                wbc = K.spmv(Rm, wr)
                out = torch.full((n,), np.nan, dtype=torch.float64, device=DEV)
                bc = torch.full((nc,), np.nan, dtype=torch.float64, device=DEV)
                ops.stencil_smooth(dA, None if zero else dev(x0), dev(b), 0.8, S, out, None, restrict=(dR, bc))
                assert np.array_equal(out.cpu().numpy(), want), (m, kind, S, zero)
                got = bc.cpu().numpy()
                assert not np.isnan(got).any(), (m, kind, S, zero, np.flatnonzero(np.isnan(got))[:8])
                assert np.array_equal(got, wbc), (m, kind, S, zero, np.flatnonzero(got != wbc)[:8])
        # every coarse row through the pattern table instead of the frequent-pattern shortcut: same bits
        hot, T.hot = T.hot, -1
        bc = torch.full((nc,), np.nan, dtype=torch.float64, device=DEV)
        ops.stencil_smooth(dA, None, dev(b), 0.8, 3, out, None, restrict=(dR, bc))
        T.hot = hot
        assert np.array_equal(bc.cpu().numpy(), wbc)
    finally:
        ops.tune_set("fused_seg_lines", 0)
        ops.set_tiled_enabled(True)


def test_tiled_post_smoothing_eight_wave_variant_against_the_oracle():
    """Grids of >= 768 lines run the correcting tiled pass on 8-wave workgroups of four lines per wave
    (stencil_tile_kernel<..., PROL, RBV = 4>: level 1 of cfg#4).  That instantiation against the oracle's prolongation +
    S Jacobi sweeps directly, bitwise -- 5-point 769^2 and the 9-point Galerkin operator of a 1537^2 grid -- and the
    4-line-per-wave variant forced on a small grid as well."""
    for side, kind in ((769, "5pt"), (769, "9pt"), (257, "5pt")):
        if kind == "5pt":
            A = K.as_csr(P.poisson_2d_structured(side - 1)[0])
        else:
            Af = P.poisson_2d_structured(2 * (side - 1))[0]
            Pf = P.tensor_interpolator_2d(2 * (side - 1) + 1)
            A = K.as_csr(sp.csr_matrix(Pf.T @ Af @ Pf))
        Pm = K.as_csr(sp.csr_matrix(P.tensor_interpolator_2d(side)))
        n, nc = A.shape[0], Pm.shape[1]
        dA = ops.DeviceCSR.from_scipy(A, DEV)
        dA.pack()
        dP = ops.DeviceCSR.from_scipy(Pm, DEV)
        dP.pack()
        assert ops._fused_kind(dA) == "tile" and ops.stencil_smooth_prolong_available(dA, dP)
        rng = np.random.default_rng(side)
        x0, b, e = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(nc)
        old = ops.tune_get("tile_prol_wide_lines")
        try:
            ops.tune_set("tile_prol_wide_lines", 0 if side < 768 else old)       # small grid: force the wide variant
            assert side >= ops.tune_get("tile_prol_wide_lines")
            want = K.spmv(Pm, e, x0.copy(), 1.0, 1.0)
            for S in (1, 2, 3):
                want = K.jacobi(A, want, b, 0.8)
                out = torch.full((n,), np.nan, dtype=torch.float64, device=DEV)
                ops.stencil_smooth(dA, dev(x0), dev(b), 0.8, S, out, None, prolong=(dP, dev(e)))
                assert np.array_equal(out.cpu().numpy(), want), (side, kind, S)
        finally:
            ops.tune_set("tile_prol_wide_lines", old)


def test_prolong_twin_only_for_two_by_two_window_transfers():
    """ProlongTwin.from_patterns accepts the tensor-product interpolation and nothing wider: an L2-type
    transfer (3-point rows per axis) keeps its own launch."""
    Pm = sp.csr_matrix(P.tensor_interpolator_2d(33))
    dP = ops.DeviceCSR.from_scipy(Pm, DEV)
    dP.pack()
    assert dP.patterns is not None and dP.prolong is not None
    l2 = P.pseudo_l2_interpolator_1d(33)
    dQ = ops.DeviceCSR.from_scipy(sp.kron(l2, l2).tocsr(), DEV)
    dQ.pack()
    assert dQ.prolong is None
    dR = dP.transpose()
    dR.pack()
    assert dR.restrict is not None and dP.restrict is None and dR.prolong is None
    dQt = dQ.transpose()
    dQt.pack()
    assert dQt.restrict is None
    A = K.as_csr(P.poisson_2d_structured(32)[0])
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dA.pack()
    assert dA.prolong is None
    with pytest.raises(ops.LmgError):
        ops.stencil_smooth(dA, dev(np.zeros(33 * 33)), dev(np.zeros(33 * 33)), 0.8, 3,
                           torch.empty(33 * 33, dtype=torch.float64, device=DEV), None, prolong=(dQ, dev(np.zeros(17 * 17))))


def test_row_pattern_sweeps_full_size_4097_bit_exact_vs_oracle():
    """The instantiations the headline bench times -- stencil_sweep_kernel<JACOBI|RESIDUAL, NT = true> on
    the 16.8 M-row fine level of cfg#4 (and rpat_sweep_kernel<.., 5, 2, NT = true>, its fallback) --
    against oracle/lmg_oracle.c, bitwise, at full size."""
    A = K.as_csr(P.poisson_2d_structured(4096)[0])
    n = A.shape[0]
    assert n == 4097 * 4097 and n >= ops.tune_get("rpat_nt_rows") and n >= ops.tune_get("stencil_nt_rows")
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    R = dA.pack()
    assert isinstance(R, ops.RowPatterns) and R.npat == 2 and R.max_len == 5
    assert dA.stencil is not None and dA.stencil.W == 4097 and dA.stencil.umask == 0b010111010
    rng = np.random.default_rng(4097)
    x, b = rng.standard_normal(n), rng.standard_normal(n)
    dx, db = dev(x), dev(b)
    out = torch.empty(n, dtype=torch.float64, device=DEV)
    want_j = {omega: K.jacobi(A, x, b, omega) for omega in (0.8, 1.0)}
    wr, wn2 = K.residual(A, x, b)
    y0 = rng.standard_normal(n)
    want_y = K.spmv(A, x, y0, 1.0, 1.0)
    part = torch.empty(ops.partials_count(n), dtype=torch.float64, device=DEV)
    n2 = torch.zeros(1, dtype=torch.float64, device=DEV)
    try:
        for stencil in (True, False):
            ops.set_stencil_enabled(stencil)
            for omega in (0.8, 1.0):
                out.zero_()
                ops.csr_jacobi(dA, dx, db, omega, out)
                assert np.array_equal(out.cpu().numpy(), want_j[omega]), (stencil, omega)
            ops.csr_residual_norm2(dA, dx, db, out, part, n2)
            assert np.array_equal(out.cpu().numpy(), wr), stencil
            assert abs(n2.item() - wn2) <= 1e-13 * wn2
            y = dev(y0.copy())
            ops.csr_spmv(dA, dx, y, 1.0, 1.0)
            assert np.array_equal(y.cpu().numpy(), want_y), stencil
    finally:
        ops.set_stencil_enabled(True)
    # the packed twin and the plain CSR kernel at the same size (what --no-patterns / --no-packed time)
    dB = ops.DeviceCSR.from_scipy(A, DEV)
    assert isinstance(dB.pack(patterns=False), ops.PackedCSR)
    want = K.jacobi(A, x, b, 0.8)
    ops.csr_jacobi(dB, dx, db, 0.8, out)
    assert np.array_equal(out.cpu().numpy(), want)
    try:
        ops.set_packed_enabled(False)
        ops.csr_jacobi(dB, dx, db, 0.8, out)
        assert np.array_equal(out.cpu().numpy(), want)
    finally:
        ops.set_packed_enabled(True)


@pytest.mark.parametrize("kind", ["tensor2d_odd", "tensor2d_even", "1d_odd", "1d_even", "l2_tensor2d"])
@pytest.mark.parametrize("nt", [False, True], ids=["cached", "nontemporal"])
def test_grid_transfer_row_patterns_bit_exact(kind, nt):
    """Prolongation P and restriction R = P^T of nested grids as row patterns relative to a column-base
    map (lmg_rpat_sweep_grid): u += P e (Multigrid.py:115) and r_c = R r (:93) bitwise against the
    oracle's CSR loops, for the reference's 1-D interpolator (odd and even n: the even-n quirk of
    Multigrid.py:139-142 is just more patterns), its tensor product, and the L2-type transfer."""
    if kind.startswith("tensor2d"):
        s = 65 if kind.endswith("odd") else 66
        P1 = P.geometric_interpolator_1d(s)
        Pm = sp.kron(P1, P1).tocsr()
    elif kind.startswith("1d"):
        Pm = P.geometric_interpolator_1d(3001 if kind.endswith("odd") else 3000).tocsr()
    else:
        l2 = P.pseudo_l2_interpolator_1d(65)
        Pm = sp.kron(l2, l2).tocsr()
    Pm = K.as_csr(Pm)
    Rm = K.as_csr(sp.csr_matrix(Pm.T))
    rng = np.random.default_rng(31)
    nt_default = ops.tune_get("rpat_nt_rows")
    try:
        if nt:
            ops.tune_set("rpat_nt_rows", 1)
        for M in (Pm, Rm):
            dM = ops.DeviceCSR.from_scipy(M, DEV)
            R = dM.pack()
            assert isinstance(R, ops.RowPatterns) and R.grid_map is not None and dM.packed is None, (kind, M.shape)
            assert dM.stencil is None and R.bytes() < 0.2 * dM.bytes()
            if kind.startswith("tensor2d") or kind == "l2_tensor2d":
                assert R.grid_map[0] == math.isqrt(M.shape[0])            # the 2-D map, not the 1-D one
            # the map, restated: every entry's column = base(row) + pattern offset
            pid, ptr, off = R.pid.cpu().numpy(), R.pat_ptr.cpu().numpy(), R.pat_off.cpu().numpy()
            for i in (0, 1, M.shape[0] // 3, M.shape[0] // 2, M.shape[0] - 2, M.shape[0] - 1):
                s_, e_ = M.indptr[i], M.indptr[i + 1]
                base = int(ops.RowPatterns.grid_base(R.grid_map, np.array([i], dtype=np.int64))[0])
                assert np.array_equal(off[ptr[pid[i]]:ptr[pid[i] + 1]] + base, M.indices[s_:e_])
            x, y0 = rng.standard_normal(M.shape[1]), rng.standard_normal(M.shape[0])
            for variant in (0, 1, 2, 3, 4):
                ops.tune_set("rpat_variant", variant)
                for alpha, beta in ((1.0, 0.0), (1.0, 1.0), (-0.5, 2.0)):
                    y = dev(y0.copy())
                    ops.csr_spmv(dM, dev(x), y, alpha, beta)
                    assert np.array_equal(y.cpu().numpy(), K.spmv(M, x, y0, alpha, beta)), (kind, variant, alpha, beta)
            # new values, same pattern (Galerkin rebuild path): the twin keeps its map
            dM.vals.mul_(2.0)
            dM.repack_values()
            assert dM.patterns is not None and dM.patterns.grid_map == R.grid_map
            y = dev(y0.copy())
            ops.csr_spmv(dM, dev(x), y, 1.0, 0.0)
            assert np.array_equal(y.cpu().numpy(), K.spmv(M * 2.0, x, y0, 1.0, 0.0))
    finally:
        ops.tune_set("rpat_variant", 0)
        ops.tune_set("rpat_nt_rows", nt_default)


def test_grid_maps_never_pass_unverified():
    # learned-like transfers (all-distinct values) have no repeating rows under any map
    base = sp.kron(P.pseudo_l2_interpolator_1d(65), P.pseudo_l2_interpolator_1d(65)).tocsr()
    Q = K.as_csr(P.learned_like(base, 43))
    for M in (Q, K.as_csr(sp.csr_matrix(Q.T))):
        dM = ops.DeviceCSR.from_scipy(M, DEV)
        tw = dM.pack()
        assert dM.patterns is None and tw is not None
        x = np.random.default_rng(3).standard_normal(M.shape[1])
        y = torch.empty(M.shape[0], dtype=torch.float64, device=DEV)
        ops.csr_spmv(dM, dev(x), y)
        assert np.array_equal(y.cpu().numpy(), K.spmv(M, x, np.zeros(M.shape[0]), 1.0, 0.0))
    # a tensor transfer with ONE entry perturbed still verifies (it is just one more pattern) ...
    Pm = K.as_csr(sp.kron(P.geometric_interpolator_1d(33), P.geometric_interpolator_1d(33)).tocsr())
    Pm.data[777] *= 1.5
    dM = ops.DeviceCSR.from_scipy(Pm, DEV)
    assert isinstance(dM.pack(), ops.RowPatterns)
    x = np.random.default_rng(4).standard_normal(Pm.shape[1])
    y = torch.empty(Pm.shape[0], dtype=torch.float64, device=DEV)
    ops.csr_spmv(dM, dev(x), y)
    assert np.array_equal(y.cpu().numpy(), K.spmv(Pm, x, np.zeros(Pm.shape[0]), 1.0, 0.0))
    # ... and with the maps switched off the packed CSR runs as before
    try:
        ops.set_grid_maps_enabled(False)
        dM = ops.DeviceCSR.from_scipy(Pm, DEV)
        assert isinstance(dM.pack(), ops.PackedCSR) and dM.patterns is None
    finally:
        ops.set_grid_maps_enabled(True)


def test_row_patterns_are_refused_when_rows_do_not_repeat():
    for name in ("val64_col16_jittered", "val16_col16_ragged"):
        dA = ops.DeviceCSR.from_scipy(packed_case(name), DEV)
        Pk = dA.pack()
        assert dA.patterns is None and isinstance(Pk, ops.PackedCSR), name
    # too many pattern entries in total (more than 1024), although few rows
    rng = np.random.default_rng(2)
    A = sp.csr_matrix(rng.standard_normal((40, 40)))
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dA.pack()
    assert dA.patterns is None and dA.packed is not None
    # values changed in place: the pattern twin is rebuilt, or dropped when rows stop repeating
    A = rpat_case("poisson2d_129")
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    assert isinstance(dA.pack(), ops.RowPatterns)
    x = rng.standard_normal(A.shape[0])
    y = torch.empty(A.shape[0], dtype=torch.float64, device=DEV)
    B = A.copy()
    B.data = B.data * 3.0
    dA.vals.copy_(dev(B.data))
    dA.repack_values()
    assert dA.patterns is not None
    ops.csr_spmv(dA, dev(x), y)
    assert np.array_equal(y.cpu().numpy(), K.spmv(B, x, np.zeros_like(x), 1.0, 0.0))
    B.data = rng.standard_normal(B.nnz)
    dA.vals.copy_(dev(B.data))
    dA.repack_values()
    assert dA.patterns is None and dA.packed is not None
    ops.csr_spmv(dA, dev(x), y)
    assert np.array_equal(y.cpu().numpy(), K.spmv(B, x, np.zeros_like(x), 1.0, 0.0))


# ---- sliced-ELL twin (sell.hip) ----------------------------------------------------------------
@functools.lru_cache(maxsize=None)
def sell_case(name):
    if name == "l2_galerkin_25":
        return packed_case("val64_longrows_l2_galerkin")
    if name == "l2_galerkin_48":                      # second Galerkin level of learned-like transfers
        A2, _ = P.variable_coeff_poisson_2d_structured(128, seed=44)
        M = A2
        for li, sz in enumerate((129, 65)):
            l2 = P.pseudo_l2_interpolator_1d(sz)
            Q = P.learned_like(sp.kron(l2, l2).tocsr(), 43 + li)
            M = sp.csr_matrix(Q.T @ M @ Q)
        return K.as_csr(M)
    if name == "dense_rows_300":                      # dense transfer operators give rows beyond the packed limit of 255
        rng = np.random.default_rng(19)
        return K.as_csr(sp.csr_matrix(rng.standard_normal((300, 300)) + 300.0 * np.eye(300)))
    if name == "random_wide_20":                      # columns all over the place: int32 columns
        rng = np.random.default_rng(17)
        n = 70001                                     # not a multiple of 64: ragged last slice
        r = np.repeat(np.arange(n), 19)               # 19 random columns in EVERY row: near-uniform lengths
        c = rng.integers(0, n, r.size)
        return K.as_csr(sp.coo_matrix((rng.standard_normal(r.size), (r, c)), shape=(n, n)).tocsr() + sp.identity(n) * 5.0)
    raise KeyError(name)


@pytest.mark.parametrize("name", ["l2_galerkin_25", "l2_galerkin_48", "random_wide_20", "dense_rows_300"])
def test_sliced_ell_sweeps_bit_exact(name):
    A = sell_case(name)
    n = A.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    S = dA.pack()
    assert isinstance(S, ops.SellCSR) and dA.packed is None and dA.patterns is None, name
    assert S.colmode == (1 if name == "random_wide_20" else 0)
    # the format, restated: slices of 64 rows padded to their longest row, column-major inside a slice
    lens = np.diff(A.indptr)
    nsl = (n + 63) // 64
    padded_lens = np.array([lens[s * 64:(s + 1) * 64].max() for s in range(nsl)])
    assert np.array_equal(S.slice_len.cpu().numpy(), padded_lens)
    base = np.concatenate([[0], np.cumsum(padded_lens * 64)[:-1]])
    assert np.array_equal(S.slice_base.cpu().numpy(), base) and S.padded == int(padded_lens.sum()) * 64
    val, col = S.val.cpu().numpy(), S.col.cpu().numpy()
    cmin = S.slice_cmin.cpu().numpy()
    for r in (0, 1, 63, 64, n // 2, n - 1):
        s, e = A.indptr[r], A.indptr[r + 1]
        pos = base[r // 64] + 64 * np.arange(e - s) + (r % 64)
        assert np.array_equal(val[pos], A.data[s:e])
        got_cols = col[pos].astype(np.int64)
        if S.colmode == 0:
            got_cols = (got_cols & 0xFFFF) + cmin[r // 64]
        assert np.array_equal(got_cols, A.indices[s:e])
    rng = np.random.default_rng(29)
    x, b, y0 = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    try:
        for alpha, beta in ((1.0, 0.0), (1.0, 1.0), (-0.5, 2.0)):
            y = dev(y0.copy())
            ops.csr_spmv(dA, dev(x), y, alpha, beta)
            assert np.array_equal(y.cpu().numpy(), K.spmv(A, x, y0, alpha, beta)), (name, alpha, beta)
        r = torch.empty(n, dtype=torch.float64, device=DEV)
        part = torch.empty(ops.partials_count(n), dtype=torch.float64, device=DEV)
        n2 = torch.zeros(1, dtype=torch.float64, device=DEV)
        ops.csr_residual_norm2(dA, dev(x), dev(b), r, part, n2)
        wr, wn2 = K.residual(A, x, b)
        assert np.array_equal(r.cpu().numpy(), wr)
        assert abs(n2.item() - wn2) <= 1e-13 * wn2
        n2b = torch.zeros(1, dtype=torch.float64, device=DEV)
        ops.csr_residual_norm2(dA, dev(x), dev(b), None, part, n2b)
        assert n2b.item() == n2.item()
        for omega in (1.0, 0.8):
            out = torch.empty(n, dtype=torch.float64, device=DEV)
            ops.csr_jacobi(dA, dev(x), dev(b), omega, out)
            assert np.array_equal(out.cpu().numpy(), K.jacobi(A, x, b, omega)), (name, omega)
        ops.set_packed_enabled(False)
        out2 = torch.empty(n, dtype=torch.float64, device=DEV)
        ops.csr_jacobi(dA, dev(x), dev(b), 0.8, out2)
        assert torch.equal(out, out2)
        ops.set_packed_enabled(True)
        # new values on the same pattern (Galerkin rebuild): only the value stream is rewritten
        B = A.copy()
        B.data = rng.standard_normal(B.nnz)
        dA.vals.copy_(dev(B.data))
        dA.repack_values()
        assert dA.sell is S
        ops.csr_jacobi(dA, dev(x), dev(b), 0.8, out)
        assert np.array_equal(out.cpu().numpy(), K.jacobi(B, x, b, 0.8))
    finally:
        ops.set_packed_enabled(True)


def test_sliced_ell_is_refused_for_ragged_rows():
    # one very long row per slice would pad every slice to its length
    A = case("ragged_1000")
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dA.pack()
    assert dA.sell is None
