"""Distributed V-cycle with the real HIP kernels and RCCL, as far as ONE GPU allows:
a world_size-1 "nccl" group drives the partitioned code path (local rectangular matrices,
ghost layout, all_gather of the replicated level, all-reduced norm).  The multi-rank
logic itself is covered by tests/test_dist_cpu.py (gloo, world 2 and 3)."""
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world1_nccl_path_matches_single_gpu_bitwise():
    import torch.distributed as dist
    from learnmultigrid_amd import problems as P
    from learnmultigrid_amd.dist import DistributedVCycle
    from learnmultigrid_amd.hierarchy import Hierarchy
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        m, levels = 256, 5
        A, rhs = P.poisson_2d_structured(m)
        hier = P.geometric_hierarchy_2d(m + 1, levels)
        D = DistributedVCycle.from_problem(A, hier, "cuda:0", grid_side=m + 1, replicate_below=10000)
        assert D.n_dist == 2
        D.set_rhs(rhs)
        with torch.cuda.stream(D.stream):
            norms = [D.residual_norm()]
            for _ in range(4):
                D.cycle("Jacobi", 3, 0.8)
                norms.append(D.residual_norm())
            x = D.dl[0].x[D.dl[0].own].cpu().numpy()
        H = Hierarchy(A, hier, "cuda:0")
        H.levels[0].b.copy_(torch.from_numpy(rhs.ravel().copy()).to("cuda:0"))
        with torch.cuda.stream(H.stream):
            ref = [H.residual_norm()]
            for _ in range(4):
                H.cycle("Jacobi", 3, 0.8)
                ref.append(H.residual_norm())
            xr = H.levels[0].x.cpu().numpy()
        assert np.array_equal(x, xr)
        assert np.allclose(norms, ref, rtol=1e-13, atol=0)
        assert norms[-1] < 1e-3 * norms[0]
    finally:
        dist.destroy_process_group()


def _gpu_worker(rank, world, port, out_dir, transfer="geometric", fuse_all=False):
    import math
    import os
    import sys
    import torch.distributed as dist
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    if fuse_all:
        # the product runs the register-blocked passes only from 12 M rows on (LDS-tiled ones below); force them on these small blocks
        from learnmultigrid_amd import ops as _ops
        _ops.FUSED_MIN_ROWS = 0
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from learnmultigrid_amd import problems as P
        from learnmultigrid_amd.dist import DistributedVCycle
        from learnmultigrid_amd.hierarchy import Hierarchy
        torch.cuda.set_device(0)
        m, levels = 384, 5
        A, rhs = P.poisson_2d_structured(m)
        if transfer == "geometric":
            hier = P.geometric_hierarchy_2d(m + 1, levels)
        else:
            import scipy.sparse as sp
            hier = []
            for li, sz in enumerate(P.level_sizes(m + 1, levels)[:-1]):
                l2 = P.pseudo_l2_interpolator_1d(sz)
                hier.append(P.learned_like(sp.kron(l2, l2).tocsr(), 43 + li))
        D = DistributedVCycle.from_problem(A, hier, "cuda:0", grid_side=m + 1, replicate_below=20000)
        assert D.host_staged and D.n_dist == 2
        if transfer == "geometric":
            # local operators run on their lossless twins: row patterns for the square grid operators
            # (ghost rows are one more pattern) and for the transfers, relative to the column-base map of the
            # local blocks' line lengths (verified entry by entry)
            assert all(d.A.patterns is not None for d in D.dl)
            for l, d in enumerate(D.dl):
                wf, wc = math.isqrt(D.full.levels[l].n), math.isqrt(D.full.levels[l + 1].n)
                assert d.P.patterns is not None and d.P.patterns.grid_map == (wf, wc, 1, 1, 0), (l, d.P.patterns)
                assert d.R.patterns is not None and d.R.patterns.grid_map == (wc, 2 * wf, 0, 0, 1), (l, d.R.patterns)
        else:
            # 25-entry Galerkin rows and restrictions with all-distinct values: sliced ELL
            assert D.dl[1].A.sell is not None and D.dl[0].R.sell is not None
        D.set_rhs(rhs)
        with torch.cuda.stream(D.stream):
            norms = [D.residual_norm()]
            for _ in range(3):
                D.cycle("Jacobi", 3, 0.8)
                norms.append(D.residual_norm())
            if transfer == "learned":
                # config #5: new coefficients on the same pattern, numeric Galerkin rebuild, go on
                rng = np.random.default_rng(77)
                new_vals = torch.from_numpy(np.ascontiguousarray(A.tocsr().data * (1.0 + 0.2 * rng.random(A.nnz)))).to("cuda:0")
                D.rebuild_numeric(new_vals)
                for _ in range(2):
                    D.cycle("Jacobi", 3, 0.8)
                    norms.append(D.residual_norm())
        x = D.gather_solution()
        H = Hierarchy(A, hier, "cuda:0")
        H.levels[0].b.copy_(torch.from_numpy(rhs.ravel().copy()).to("cuda:0"))
        with torch.cuda.stream(H.stream):
            ref = [H.residual_norm()]
            for _ in range(3):
                H.cycle("Jacobi", 3, 0.8)
                ref.append(H.residual_norm())
            if transfer == "learned":
                H.rebuild_numeric(new_vals)
                for _ in range(2):
                    H.cycle("Jacobi", 3, 0.8)
                    ref.append(H.residual_norm())
            xr = H.levels[0].x.cpu().numpy()
        ok = bool(np.array_equal(x, xr)) and bool(np.allclose(norms, ref, rtol=1e-13, atol=0))
        np.save(os.path.join(out_dir, "ok_%d.npy" % rank),
                np.array([ok, norms[3] < (1e-3 if transfer == "geometric" else 0.9) * norms[0]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_several_ranks_on_one_gpu_match_single_gpu_bitwise(tmp_path, world):
    """Real HIP kernels on the partitioned layouts (ghost rows, packed local operators, hipGraph
    tail) with 2 and 3 ranks sharing cuda:0; messages go through gloo + host staging because RCCL
    refuses two ranks on one device.  RCCL itself is exercised by the world-1 test above."""
    import os
    import torch.multiprocessing as mp
    mp.spawn(_gpu_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok = np.load(os.path.join(str(tmp_path), "ok_%d.npy" % r))
        assert ok.all(), (r, ok)


def test_two_ranks_with_fused_smoothing_passes_match_single_gpu_bitwise(tmp_path):
    """The ranks' sweeps (and the residual) as fused passes over the local blocks with ghost layers
    (what a rank runs from 4 M local rows on: N = 2, 4 at cfg#4): same bits as the single-GPU cycle."""
    import os
    import torch.multiprocessing as mp
    mp.spawn(_gpu_worker, args=(2, _free_port(), str(tmp_path), "geometric", True), nprocs=2, join=True)
    for r in range(2):
        ok = np.load(os.path.join(str(tmp_path), "ok_%d.npy" % r))
        assert ok.all(), (r, ok)


def test_two_ranks_with_learned_transfers_match_single_gpu_bitwise(tmp_path):
    """Same check with wide learned-like transfers: sliced-ELL / packed local operators with real
    ghost rows, halo requirements measured from the matrices."""
    import os
    import torch.multiprocessing as mp
    mp.spawn(_gpu_worker, args=(2, _free_port(), str(tmp_path), "learned"), nprocs=2, join=True)
    for r in range(2):
        ok = np.load(os.path.join(str(tmp_path), "ok_%d.npy" % r))
        assert ok.all(), (r, ok)


def _gs_gpu_worker(rank, world, port, out_dir, halo_depth):
    import os
    import sys
    import torch.distributed as dist
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from learnmultigrid_amd import ops, problems as P
        from learnmultigrid_amd.dist import DistributedVCycle
        from oracle import vcycle_ref as V
        torch.cuda.set_device(0)
        m, levels = 256, 4
        A, rhs = P.poisson_2d_structured(m)
        hier = P.geometric_hierarchy_2d(m + 1, levels)
        D = DistributedVCycle.from_problem(A, hier, "cuda:0", grid_side=m + 1, replicate_below=20000, halo_depth=halo_depth)
        twin = V.HybridGSVCycle(A, hier, [D.bounds[l] for l in range(D.n_dist)])
        b = rhs.ravel().copy()
        rng = np.random.default_rng(3)
        x0 = rng.standard_normal(A.shape[0])
        D.set_rhs(rhs)
        D.set_x(x0)
        with torch.cuda.stream(D.stream):
            D._smooth_gs(D.dl[0], 3, False)
            torch.cuda.synchronize()
        used_wave = halo_depth == 1 and ops.stencil_gs_available(D.dl[0].A)
        got = D.gather_solution()
        want = twin.smooth(0, x0.copy(), b, "GaussSeidel", 3, 1.0)
        smooth_equal = bool(np.array_equal(got, want))
        D.set_x(np.zeros(A.shape[0]))
        with torch.cuda.stream(D.stream):
            norms = [D.residual_norm()]
            x = np.zeros(A.shape[0])
            ref = [float(np.linalg.norm(b - A @ x))]
            for _ in range(3):
                D.cycle("GaussSeidel", 2)
                norms.append(D.residual_norm())
                x = twin.cycle(x, b, "GaussSeidel", 2, 1.0)
                ref.append(float(np.linalg.norm(b - A @ x)))
        for d in D.dl:
            ops.stencil_gs_check(d.A)
        ok_hist = bool(np.allclose(norms, ref, rtol=1e-10, atol=1e-14 * ref[0]))
        # (rank 0's block starts with owned lines: its operator is a plain grid operator and takes the wavefront kernel)
        np.save(os.path.join(out_dir, "gs_%d.npy" % rank), np.array([smooth_equal, ok_hist, norms[3] < 0.05 * norms[0],
                                                                     rank > 0 or used_wave == (halo_depth == 1)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,halo_depth", [(2, 1), (3, 6)])
def test_processor_block_gauss_seidel_on_the_device_matches_its_cpu_twin(tmp_path, world, halo_depth):
    """The reference's shipped smoother across ranks (processor-block Gauss-Seidel) with the real kernels -- the wavefront
    kernel on local blocks with empty ghost rows (classic halo), the level-scheduled executor on the owned rows under a deep
    halo: one smoothing step bit-identical to oracle.vcycle_ref.HybridGSVCycle, V-cycle histories at 1e-10."""
    import os
    import torch.multiprocessing as mp
    mp.spawn(_gs_gpu_worker, args=(world, _free_port(), str(tmp_path), halo_depth), nprocs=world, join=True)
    for r in range(world):
        ok = np.load(os.path.join(str(tmp_path), "gs_%d.npy" % r))
        assert ok.all(), (r, ok)
