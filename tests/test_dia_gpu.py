"""Fused smoothing passes on variable-coefficient / jittered-mesh grid operators (csrc/dia_tile.hip, -m gpu): the
operators the reference's learned transfers are built for (Multigrid.py:306-370, :741-765) against the CPU oracle's
separate Jacobi sweeps and residual, bitwise."""
import numpy as np
import pytest
import scipy.sparse as sp

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from learnmultigrid_amd import ops, problems as P   # noqa: E402
from learnmultigrid_amd.hierarchy import Hierarchy  # noqa: E402
from oracle import kernels as K                     # noqa: E402  (checker only)
from oracle import vcycle_ref as V                  # noqa: E402

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def matrix(name):
    if name == "varcoeff5_151":
        return K.as_csr(P.variable_coeff_poisson_2d_structured(150, seed=44)[0])
    if name == "jittered7_151":
        return K.as_csr(P.jittered_poisson_2d(150, seed=42)[0])
    if name == "jittered7_300x":
        return K.as_csr(P.jittered_poisson_2d(299, seed=7)[0])
    if name == "perturbed9_129":
        A = P.poisson_2d_structured(256)[0]
        Pm = P.tensor_interpolator_2d(257)
        G = sp.csr_matrix(Pm.T @ A @ Pm)
        G.sort_indices()
        rng = np.random.default_rng(3)
        G.data = G.data * (1.0 + 0.1 * rng.random(G.nnz))            # 9-point, every value distinct
        return K.as_csr(G)
    raise KeyError(name)


@pytest.mark.parametrize("name,umask", [("varcoeff5_151", 0x0BA), ("jittered7_151", None), ("jittered7_300x", None),
                                        ("perturbed9_129", 0x1FF)])
@pytest.mark.parametrize("rows", [32, 64])
def test_dia_fused_smoothing_equals_separate_sweeps(name, umask, rows):
    A = matrix(name)
    n = A.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dA.pack()
    assert dA.patterns is None and dA.stencil is None and dA.dia is not None and dA.packed is not None
    D = dA.dia
    assert D.W * D.W == n and (umask is None or D.umask == umask) and D.umask in (0x0BA, 0x1BB, 0x0FE, 0x1FF)
    assert ops._fused_kind(dA) == "dia" and ops.stencil_smooth_available(dA)
    # the twin is the matrix: slot arrays against the CSR entries
    coo = A.tocoo()
    slots = [s for s in range(9) if (D.umask >> s) & 1]
    dia = D.dia.cpu().numpy().reshape(len(slots), n)
    off = coo.col.astype(np.int64) - coo.row
    for q, s in enumerate(slots):
        want = np.zeros(n)
        m = off == (s // 3 - 1) * D.W + (s % 3 - 1)
        want[coo.row[m]] = coo.data[m]
        assert np.array_equal(dia[q], want), (name, s)
    rng = np.random.default_rng(77)
    x0, b = rng.standard_normal(n), rng.standard_normal(n)
    try:
        ops.tune_set("dia_rows", rows)
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
                        assert np.array_equal(got, want), (name, omega, S, zero, resid, np.flatnonzero(got != want)[:8])
                        if resid:
                            assert np.array_equal(r.cpu().numpy(), wr), (name, omega, S, zero)
    finally:
        ops.tune_set("dia_rows", 0)


def test_dia_twin_only_for_three_by_three_grid_operators():
    for A in (K.as_csr(sp.random(5000, 5000, density=0.001, random_state=1) + sp.identity(5000)),
              K.as_csr(P.poisson_1d_fd(8192)[0])):
        dA = ops.DeviceCSR.from_scipy(A, DEV)
        dA.pack()
        assert dA.dia is None
    # 25-point Galerkin operator of an L2-type transfer: radius 2, no twin
    A = P.jittered_poisson_2d(160, seed=42)[0]
    l2 = P.pseudo_l2_interpolator_1d(161)
    Q = P.learned_like(sp.kron(l2, l2).tocsr(), 43)
    G = sp.csr_matrix(Q.T @ A @ Q)
    dG = ops.DeviceCSR.from_scipy(G, DEV)
    dG.pack()
    assert dG.dia is None


def test_cycle_with_dia_passes_equals_cycle_with_separate_sweeps_and_the_oracle():
    """cfg#3 in small: jittered 7-point fine operator, learned-like L2-type transfers, 3 levels -- the fused fine-level
    passes change no bit of the cycle (fused == one launch per sweep) and the history matches the oracle at 1e-10;
    a numeric Galerkin rebuild refreshes the twin."""
    m = 128
    A, rhs = P.jittered_poisson_2d(m, seed=42)
    hier = []
    for li, sz in enumerate(P.level_sizes(m + 1, 3)[:-1]):
        l2 = P.pseudo_l2_interpolator_1d(sz)
        hier.append(P.learned_like(sp.kron(l2, l2).tocsr(), 43 + li))

    def run(dia):
        ops.set_dia_enabled(dia)
        try:
            H = Hierarchy(A, hier, DEV)
            assert (H.levels[0].A.dia is not None) == dia
            with torch.cuda.stream(H.stream):
                H.levels[0].b.copy_(dev(rhs.ravel().copy()))
                ops.zero(H.levels[0].x)
                hist = [H.residual_norm()]
                for _ in range(4):
                    H.cycle("Jacobi", 3, 0.8)
                    hist.append(H.residual_norm())
                x = H.levels[0].x.cpu().numpy().copy()
            return H, hist, x
        finally:
            ops.set_dia_enabled(True)
    H1, h1, x1 = run(True)
    _H0, h0, x0 = run(False)
    assert np.array_equal(x1, x0) and h1 == h0
    ref = V.HoistedVCycle(K.as_csr(A), [sp.csr_matrix(q) for q in hier])
    x = np.zeros(A.shape[0])
    want = [np.linalg.norm(rhs.ravel() - A @ x)]
    for _ in range(4):
        x = ref.cycle(x, rhs.ravel(), "Jacobi", 3, 0.8)
        want.append(np.linalg.norm(rhs.ravel() - A @ x))
    np.testing.assert_allclose(h1, want, rtol=1e-10, atol=1e-14 * max(want))
    # new coefficients on the same pattern: the twin follows
    newv = H1.levels[0].A.vals * 1.5
    H1.rebuild_numeric(newv)
    D = H1.levels[0].A.dia
    assert D is not None
    A2 = K.as_csr(A * 1.5)
    n = A2.shape[0]
    rng = np.random.default_rng(1)
    xx, bb = rng.standard_normal(n), rng.standard_normal(n)
    out = torch.empty(n, dtype=torch.float64, device=DEV)
    ops.stencil_smooth(H1.levels[0].A, dev(xx), dev(bb), 0.8, 1, out, None)
    assert np.array_equal(out.cpu().numpy(), K.jacobi(K.as_csr(sp.csr_matrix((newv.cpu().numpy(), A2.indices, A2.indptr), shape=A2.shape)), xx, bb, 0.8))


def test_dia_pass_full_size_4097_varcoeff_bit_exact_vs_oracle():
    """cfg#5's fine-level operator class at 16.8 M rows: 3 sweeps + residual in one pass against the oracle's four
    separate passes, bitwise."""
    A = K.as_csr(P.variable_coeff_poisson_2d_structured(4096, seed=44)[0])
    n = A.shape[0]
    dA = ops.DeviceCSR.from_scipy(A, DEV)
    dA.pack()
    assert dA.dia is not None and dA.dia.W == 4097 and dA.dia.umask == 0x0BA
    rng = np.random.default_rng(4097)
    x0, b = rng.standard_normal(n), rng.standard_normal(n)
    want = x0
    for _ in range(3):
        want = K.jacobi(A, want, b, 0.8)
    wr, _ = K.residual(A, want, b)
    out = torch.empty(n, dtype=torch.float64, device=DEV)
    r = torch.empty(n, dtype=torch.float64, device=DEV)
    ops.stencil_smooth(dA, dev(x0), dev(b), 0.8, 3, out, r)
    assert np.array_equal(out.cpu().numpy(), want)
    assert np.array_equal(r.cpu().numpy(), wr)
