#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference in the build container.

Runs only where /root/reference exists (the build container); the GPU box and the
test-suite never execute this file, they only read the committed .npz fixtures
(inputs + expected outputs, no reference source).

learn_multigrid/solvers/Multigrid.py imports two packages that are not installed
here and are not vendored in the reference (no version pinned anywhere):
  * pyamg.relaxation.relaxation.gauss_seidel  (Multigrid.py:7, called :88,:121)
  * cachetools.cached / TTLCache              (Multigrid.py:19, :29, :126)
They are replaced by the two minimal stand-ins below before the import:
a sequential forward Gauss-Seidel sweep (the restatement in oracle/lmg_oracle.c,
so the V-cycle goldens pin the reference's CONTROL FLOW and SciPy arithmetic but
NOT pyamg's kernel: "parity unpinned" at that boundary) and a dict memoiser.
The stand-in sweep is cross-checked here against the reference's own importable
GaussSeidel class (GaussSeidel.py:22-37) and the check result is stored.

Usage:  python tools/make_golden.py            (writes tests/golden/)
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")

from oracle import kernels as K  # noqa: E402  (the sweep restatement)


def _install_stubs():
    def gauss_seidel(A, x, b, iterations=1, sweep="forward"):
        assert sweep == "forward"
        K.gs_forward(A, x, b, iterations)

    pyamg = types.ModuleType("pyamg")
    rel = types.ModuleType("pyamg.relaxation")
    rr = types.ModuleType("pyamg.relaxation.relaxation")
    rr.gauss_seidel = gauss_seidel
    rel.relaxation = rr
    pyamg.relaxation = rel
    sys.modules.update({"pyamg": pyamg, "pyamg.relaxation": rel,
                        "pyamg.relaxation.relaxation": rr})

    ct = types.ModuleType("cachetools")

    class TTLCache(dict):
        def __init__(self, maxsize=0, ttl=0):
            super().__init__()

    def cached(cache):
        def deco(f):
            def wrapped(*a):
                if a not in cache:
                    cache[a] = f(*a)
                return cache[a]
            return wrapped
        return deco

    ct.TTLCache, ct.cached = TTLCache, cached
    sys.modules["cachetools"] = ct


_install_stubs()
with contextlib.redirect_stdout(io.StringIO()):
    from learn_multigrid.solvers.Solver import DirectSolver            # noqa: E402
    from learn_multigrid.solvers.Jacobi import Jacobi                  # noqa: E402
    from learn_multigrid.solvers.GaussSeidel import GaussSeidel        # noqa: E402
    from learn_multigrid.solvers.Multigrid import (GeometricMG, SemiGeometricMG,  # noqa: E402
                                                   NeuralMG, Multigrid)
    from learn_multigrid.utilities.laplacian import laplacian_1d_fd_bc  # noqa: E402
    from learn_multigrid.mesh.Mesh1D import Mesh1D                     # noqa: E402
    from learn_multigrid.mesh.Mesh2D import Mesh2D                     # noqa: E402
    from learn_multigrid.L2_projection.L2Projection import L2Projection  # noqa: E402
    from learn_multigrid.assembly.StiffnessMatrix import StiffnessMatrix  # noqa: E402
    from learn_multigrid.assembly.MassMatrix import MassMatrix         # noqa: E402
    from learn_multigrid.assembly.LoadVector import LoadVector         # noqa: E402
    from learn_multigrid.assembly.LoadFunction import LoadFunction     # noqa: E402
    from learn_multigrid.assembly.Quadrature import Quadrature, Quadrature2D  # noqa: E402
    from learn_multigrid.assembly.ShapeFunction import (Function, Gradient,  # noqa: E402
                                                        FunctionTriangle, GradientTriangle)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def coo(M, prefix):
    C = sp.coo_matrix(M)
    return {prefix + "_row": C.row.astype(np.int32), prefix + "_col": C.col.astype(np.int32),
            prefix + "_data": C.data.astype(np.float64),
            prefix + "_shape": np.array(C.shape, dtype=np.int64)}


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-34s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def ones(x):
    return np.ones(shape=x.shape)


# ----------------------------------------------------------------------------- G1
def g1_small_solvers():
    """test/test_solver.py:30 matrix with the rhs of test/testCG.py:18."""
    A = sp.csc_matrix(np.array([[30, 1, 15], [28, 60, 3], [100, 19, 150]], dtype=float))
    rhs = np.array([[1.0], [2.0], [3.0]])
    d = quiet(DirectSolver, A, rhs)
    quiet(d.solve)
    j = quiet(Jacobi, A, rhs)
    quiet(j.solve)
    g = quiet(GaussSeidel, A, rhs)
    quiet(g.solve)
    save("g1_small_solvers", A=A.toarray(), rhs=rhs,
         direct_solution=d.get_solution(), direct_residual=d.get_residual(),
         jacobi_track=j.get_track_res(), jacobi_iterations=j.get_iterations(),
         jacobi_solution=j.get_solution(), jacobi_residual=j.get_residual(),
         gs_track=g.get_track_res(), gs_iterations=g.get_iterations(),
         gs_solution=g.get_solution(), gs_residual=g.get_residual())


# ----------------------------------------------------------------------------- G2
def poisson_1d(ne):
    mesh = Mesh1D(regular=True, ne=ne)
    mesh.construct()
    L, X, rhs = laplacian_1d_fd_bc(mesh, ones)
    return mesh, sp.csr_matrix(L), rhs


def g2_interpolators():
    mg = quiet(Multigrid, sp.identity(3, format="csc"), np.zeros((3, 1)))
    arrs = {}
    for n in (2, 3, 9, 10, 17, 64, 1025):
        arrs["interp_%d" % n] = quiet(mg.interpolator, n, False)
    sizes = [1025]
    for _ in range(6):
        sizes.append(int(np.floor((sizes[-1] - 1) / 2)) + 1)       # Multigrid.py:130
    arrs["level_sizes_from_1025"] = np.array(sizes, dtype=np.int64)
    save("g2_interpolators", **arrs)


def g2_poisson_1d():
    for ne in (16, 64, 1024):
        mesh, A, rhs = quiet(poisson_1d, ne)
        coarse = Mesh1D(regular=True, ne=ne // 2)
        quiet(coarse.construct)
        arrs = dict(ne=ne, rhs=rhs, **coo(A, "A"))
        for kind in ("pseudo", "quasi"):
            Q, _t = quiet(L2Projection(kind, mesh, coarse).compute_transfer_1d)
            arrs.update(coo(sp.csr_matrix(Q), "Q_" + kind))
            for levels, steps in ((2, 1), (2, 3), (3, 1)):
                s = quiet(SemiGeometricMG, A, rhs, Q)
                quiet(s.solve, smoother="GaussSeidel", smooth_steps=steps, levels=levels,
                      max_iterations=100, error=1e-11)
                key = "semi_%s_L%d_s%d" % (kind, levels, steps)
                arrs[key + "_track"] = s.get_track_res()
                arrs[key + "_iterations"] = s.get_iterations()
                arrs[key + "_solution"] = s.get_solution()
        for levels, steps in ((2, 1), (3, 3), (4, 2)):
            s = quiet(GeometricMG, A, rhs)
            quiet(s.solve, smoother="GaussSeidel", smooth_steps=steps, levels=levels,
                  max_iterations=100, error=1e-11)
            key = "geo_L%d_s%d" % (levels, steps)
            arrs[key + "_track"] = s.get_track_res()
            arrs[key + "_iterations"] = s.get_iterations()
            arrs[key + "_solution"] = s.get_solution()
        # default arguments of Multigrid.solve (levels=2, "Jacobi" ignored, 1 step, 1e-8)
        s = quiet(GeometricMG, A, rhs)
        quiet(s.solve)
        arrs["geo_default_track"] = s.get_track_res()
        arrs["geo_default_iterations"] = s.get_iterations()
        # caller-supplied initial guess is smoothed IN PLACE (Multigrid.py:43,:88)
        x0 = np.linspace(0.0, 1.0, ne + 1).reshape(-1, 1) ** 2
        arrs["x0"] = x0.copy()
        s = quiet(GeometricMG, A, rhs)
        quiet(s.solve, levels=2, smooth_steps=2, max_iterations=3, error=1e-30, initial_guess=x0)
        arrs["geo_x0_track"] = s.get_track_res()
        arrs["geo_x0_solution"] = s.get_solution()
        arrs["geo_x0_mutated_guess"] = x0
        # one bare v_cycle call (Multigrid.py:77 signature)
        s = quiet(GeometricMG, A, rhs)
        u0 = np.zeros((ne + 1, 1))
        u = quiet(s.v_cycle, s.get_matrix(), u0, rhs, "GaussSeidel", 2, 1e-8, 2)
        arrs["vcycle_u"] = u
        arrs["vcycle_u0_after"] = u0
        # stand-alone smoothers on the same matrix
        j = quiet(Jacobi, A, rhs)
        quiet(j.solve, max_iterations=25)
        arrs["jacobi25_track"] = j.get_track_res()
        arrs["jacobi25_solution"] = j.get_solution()
        if ne <= 64:
            g = quiet(GaussSeidel, A, rhs)
            quiet(g.solve, max_iterations=25)
            arrs["gs25_track"] = g.get_track_res()
            arrs["gs25_solution"] = g.get_solution()
        save("g2_poisson1d_ne%d" % ne, **arrs)


def g2_gs_crosscheck():
    """Stand-in sweep vs the reference's own GaussSeidel class (n = 1025, 3 sweeps)."""
    _mesh, A, rhs = quiet(poisson_1d, 64)
    g = quiet(GaussSeidel, A, rhs)
    quiet(g.solve, max_iterations=3, error=0.0)
    x = np.zeros((65, 1))
    K.gs_forward(A, x, rhs, 3)
    rel = np.linalg.norm(x - g.get_solution()) / np.linalg.norm(g.get_solution())
    print("GS sweep restatement vs reference GaussSeidel class: rel diff %.3e" % rel)
    assert rel < 1e-14
    return rel


# ----------------------------------------------------------------------------- G3 / G6
class FakeModel:
    """Deterministic stand-in for the absent Keras models (data/models is empty):
    returns, for every 7-feature patch, 9 outputs shaped like a perturbed pseudo-L2
    coupling stencil.  Stored in the fixture so the scatter of Multigrid.py:345-368
    is pinned independently of any network."""

    def __init__(self, scale, seed):
        self.scale, self.seed = scale, seed
        self.last = None

    def predict(self, data):
        rng = np.random.RandomState(self.seed)
        base = np.array([0, 0, 1 / 12, 0, 1 / 2, 5 / 6, 1 / 2, 0, 1 / 12]) * self.scale
        out = base[None, :] * (1.0 + 0.05 * rng.standard_normal((data.shape[0], 9)))
        self.last = out
        return out


def g3_fem_1d_learnedlike():
    """test/test_B_patch.py:54-194 with a seeded irregular mesh and a fake model."""
    def fm1(x):
        return -1
    for ne in (32, 256):
        np.random.seed(1234 + ne)
        mesh = Mesh1D(regular=False, ne=ne)
        quiet(mesh.construct)
        q, phi, dphi = Quadrature(3), Function(2), Gradient(2)
        A = quiet(StiffnessMatrix(mesh).compute_stiffness_1d, dphi, q)
        M = quiet(MassMatrix(mesh).compute_mass_1d, phi, q)
        rhs = quiet(LoadVector(mesh).compute_rhs_1d, fm1)
        rhs[0] = 0
        rhs[-1] = 0
        A[1, 0] = 0
        A[-2, -1] = 0
        A[0, :] = 0
        A[-1, :] = 0
        A[0, 0] = 1
        A[-1, -1] = 1
        std = np.ones(7) * 1e-3
        mean = np.ones(7) * 1e-4
        model = FakeModel(scale=1.0 / ne, seed=77)
        nmg = quiet(NeuralMG, A, rhs, model, M, std, mean)
        Q = quiet(nmg.transfer_op, M)
        s = quiet(SemiGeometricMG, A, rhs, Q)
        quiet(s.solve, levels=2, smoother="GaussSeidel", smooth_steps=3, error=1e-10,
              max_iterations=15)
        coarse = Mesh1D(regular=True, ne=ne // 2)
        quiet(coarse.construct)
        coarse.x = mesh.get_mesh()[0::2]
        quiet(coarse.connection_matrix)
        Qq, _ = quiet(L2Projection("quasi", mesh, coarse).compute_transfer_1d)
        s2 = quiet(SemiGeometricMG, A, rhs, Qq)
        quiet(s2.solve, levels=2, smoother="GaussSeidel", smooth_steps=3, error=1e-10,
              max_iterations=15)
        save("g3_fem1d_ne%d" % ne, ne=ne, x=mesh.get_mesh(), rhs=rhs, fake_pred=model.last,
             M=np.asarray(M), **coo(sp.csr_matrix(A), "A"),
             **coo(sp.csr_matrix(Q), "Q_learned"), **coo(sp.csr_matrix(Qq), "Q_quasi"),
             learned_track=s.get_track_res(), learned_iterations=s.get_iterations(),
             learned_solution=s.get_solution(),
             quasi_track=s2.get_track_res(), quasi_iterations=s2.get_iterations())


# ----------------------------------------------------------------------------- G4
def g4_structured_2d():
    """test/thesis_structured_2d.py:380-414 without refine(): P1 assembly on the
    structured triangulation + Dirichlet rows as identity (columns untouched)."""
    def fm1(x):
        return -1
    for k in (4, 16):
        mesh = quiet(Mesh2D, k * k)
        q, dphi, phi = Quadrature2D(3), GradientTriangle(1), FunctionTriangle(1)
        A = quiet(StiffnessMatrix(mesh).compute_stiffness_2d, dphi, q)
        M = quiet(MassMatrix(mesh).compute_mass_2d, phi, q)
        rhs = quiet(LoadVector(mesh).compute_rhs_2d, LoadFunction(fm1), phi, q)
        p = mesh.p
        border = np.logical_or(np.logical_or(p[:, 0] == 0, p[:, 0] == 1),
                               np.logical_or(p[:, 1] == 0, p[:, 1] == 1))
        nodes = np.where(border)[0]
        eye = np.eye(len(p))
        A_free = sp.csr_matrix(A)
        A[nodes, :] = eye[nodes, :]
        rhs[nodes] = 0
        save("g4_structured2d_k%d" % k, k=k, p=p, conn=mesh.conn.astype(np.int32), rhs=rhs,
             **coo(A_free, "A_free"), **coo(sp.csr_matrix(A), "A"), **coo(sp.csr_matrix(M), "M"))


# ----------------------------------------------------------------------------- G5 / G6
def g5_saved_files():
    """Files written by the reference's own save() methods (StiffnessMatrix.py:38-45,
    LoadVector.py:36-38): fixtures for learnmultigrid_amd/io.py."""
    def fm1(x):
        return -1
    mesh = quiet(Mesh2D, 16)
    q, dphi, phi = Quadrature2D(3), GradientTriangle(1), FunctionTriangle(1)
    st = StiffnessMatrix(mesh)
    quiet(st.compute_stiffness_2d, dphi, q)
    st.save(os.path.join(OUT, "g5_saved_A"))
    lv = LoadVector(mesh)
    quiet(lv.compute_rhs_2d, LoadFunction(fm1), phi, q)
    lv.save(os.path.join(OUT, "g5_saved_rhs"))
    print("wrote g5_saved_A.npz / g5_saved_rhs.npy with the reference's save()")


class ScaleAwareFakeModel:
    """Like FakeModel, but scales its stencil with the mass-matrix entries it is shown, so that it
    behaves sensibly on every level of a multi-level NeuralMG run (std = 1, mean = 0)."""

    def __init__(self, seed):
        self.seed = seed
        self.calls = []

    def predict(self, data):
        rng = np.random.RandomState(self.seed + data.shape[0])
        base = np.array([0, 0, 1 / 12, 0, 1 / 2, 5 / 6, 1 / 2, 0, 1 / 12])
        out = base[None, :] * (1.5 * data[:, 1:2]) * (1.0 + 0.05 * rng.standard_normal((data.shape[0], 9)))
        self.calls.append((data.copy(), out.copy()))
        return out


def g7_neural_mg_multilevel():
    """NeuralMG(...).solve(levels=3) (Multigrid.py:200-370, call shape of
    test/test_more_levels_NN.py:62-63) with a deterministic stand-in model: Q is rebuilt from
    the coarsened mass matrix on every level."""
    def fm1(x):
        return -1
    ne = 64
    np.random.seed(4321)
    mesh = Mesh1D(regular=False, ne=ne)
    quiet(mesh.construct)
    q, phi, dphi = Quadrature(3), Function(2), Gradient(2)
    A = quiet(StiffnessMatrix(mesh).compute_stiffness_1d, dphi, q)
    M = quiet(MassMatrix(mesh).compute_mass_1d, phi, q)
    rhs = quiet(LoadVector(mesh).compute_rhs_1d, fm1)
    rhs[0] = 0
    rhs[-1] = 0
    A[1, 0] = 0
    A[-2, -1] = 0
    A[0, :] = 0
    A[-1, :] = 0
    A[0, 0] = 1
    A[-1, -1] = 1
    model = ScaleAwareFakeModel(seed=99)
    nmg = quiet(NeuralMG, A, rhs, model, M, np.ones(7), np.zeros(7))
    quiet(nmg.solve, levels=3, smoother="GaussSeidel", smooth_steps=3, error=1e-10, max_iterations=12)
    # the first two predict() calls belong to the first cycle: level 0 and level 1
    (d0, p0), (d1, p1) = model.calls[0], model.calls[1]
    save("g7_neuralmg_ne64", rhs=rhs, M=np.asarray(M), **coo(sp.csr_matrix(A), "A"),
         track=nmg.get_track_res(), iterations=nmg.get_iterations(), solution=nmg.get_solution(),
         features_l0=d0, pred_l0=p0, features_l1=d1, pred_l1=p1, n_predict_calls=len(model.calls))


def g6_cg():
    """CG.py needs np.asscalar (removed in NumPy 1.23): shimmed for this run only."""
    from learn_multigrid.solvers.CG import CG
    if not hasattr(np, "asscalar"):
        np.asscalar = lambda a: a.item()
    _mesh, A, rhs = quiet(poisson_1d, 64)
    c = quiet(CG, A, rhs)
    quiet(c.solve, max_iterations=200, error=1e-10)
    save("g6_cg_ne64", track=c.get_track_res(), iterations=c.get_iterations(), solution=c.get_solution(),
         rhs=rhs, **coo(A, "A"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    g5_saved_files()
    g6_cg()
    g7_neural_mg_multilevel()
    rel = g2_gs_crosscheck()
    g1_small_solvers()
    g2_interpolators()
    g2_poisson_1d()
    g3_fem_1d_learnedlike()
    g4_structured_2d()
    save("g0_meta", gs_crosscheck_rel=rel,
         numpy=np.array(np.__version__), scipy=np.array(__import__("scipy").__version__))
