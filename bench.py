#!/usr/bin/env python3
"""Headline benchmark: fine-level DoF.sweeps/s of the 2-D Poisson V-cycle on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json cfg#4, the size the north-star target is quoted on; it fits one
GPU): 2-D structured P1 Poisson, 4096x4096 elements = 4097^2 DoF, CSR fp64/int32, 6-level
V-cycle with the tensor-product geometric transfer and Galerkin coarse operators built by
the device SpGEMM at setup.  A "step" is ONE full V(3,3) cycle (weighted Jacobi, omega =
0.8, "as_named" semantics) over all levels, replayed from a hipGraph: 6 Jacobi sweeps +
1 residual SpMV on the fine level, the same on every coarser level, restrictions,
prolongations and the dense coarsest solve.  value = fine-level DoF x fine-level sweeps
per cycle (2*nu + 1) / time: the fine-level sweep rate the whole cycle sustains.

For N > 1 the fine grid is row-block partitioned over the ranks (strong scaling: the
problem size is fixed), halos travel over RCCL, see learnmultigrid_amd/dist.py.

Extra objects on the JSON line: `roofline` (the DOMINANT KERNEL OF THE TIMED STEP -- the fused
fine-level pre-smoothing pass: 3 sweeps + residual + restriction in one launch --, average
launch duration measured with HIP events on the launch stream, bytes = what the launch has to
move; the single fine-level Jacobi sweep, the kernel the north-star target is quoted on, is the
sub-object `single_sweep`), `cpu_baseline` (the CPU oracle running the same cycle on the same
hierarchy, 1 core), `time_to_solution`, and `other_configs`: the cycles of BASELINE configs #3
(jittered 7-point, learned-like Q) and #5 at 4097^2 (variable coefficients, learned-like Q) on
the same GPU in the same run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s copy-achievable

# HBM bytes per launch of the fine-level Jacobi sweep from rocprofv3 PMC passes
# (2 x FETCH_SIZE [gfx950 counts wide reads at half size] + WRITE_SIZE, KB -> B), keyed by
# (--size, fine-level format).  NOT measured in this run (counters need their own rocprofv3 pass):
# each entry names the committed profile it was read from, and the JSON line repeats that.
PMC_TRAFFIC = {
    (4096, "csr"): (1485261824, "profiles/r01_csr_sweepmode_pmc_fetch_write.txt"),
    (4096, "pcsr"): (672639665, "profiles/r01_packed_sweep_pmc_fetch_write.txt"),
    (4096, "rpat"): (451701453, "profiles/r01_rpat_sweep_pmc_fetch_write.txt"),
    (4096, "stencil"): (420151166, "profiles/r02_stencil_sweep_pmc_fetch_write.txt"),
    # the fused passes re-read halo lines / columns of neighbouring strips (28-line segments; 48 with the restriction):
    # 3 sweeps 2 x 202 042 KB + 133 088 KB; + residual 2 x 206 810 KB + 266 517 KB; + restricted residual (r not written)
    # 2 x 200 979 KB + 166 875 KB; 3 sweeps with the correction folded in 2 x 246 720 KB + 133 016 KB
    # (compulsory: 419.6 / 553.9 / 457.4 / 470.0 MB)
    (4096, "fused"): (550063968, "profiles/r02_fused_pass_pmc_fetch_write.txt"),
    (4096, "fused_resid"): (696460718, "profiles/r02_fused_pass_pmc_fetch_write.txt"),
    (4096, "fused_restrict"): (582484229, "profiles/r02_fused_pass_pmc_fetch_write.txt"),
    (4096, "fused_prolong"): (641489453, "profiles/r02_fused_pass_pmc_fetch_write.txt"),
}
# round 3 (scalarised passes, balanced decomposition, 28-line segments everywhere -- shorter than round 2's, hence MORE
# halo re-reads per launch although the launches are faster): 3 sweeps 2 x 217 634 KB + 132 790 KB; + residual
# 2 x 220 188 + 266 411; + restricted residual 2 x 252 008 + 169 248; correction + 3 sweeps 2 x 259 151 + 132 847
PMC_TRAFFIC.update({
    (4096, "fused"): (581691802, "profiles/r03_fused_pass_pmc_fetch_write.txt"),
    (4096, "fused_resid"): (723749069, "profiles/r03_fused_pass_pmc_fetch_write.txt"),
    (4096, "fused_restrict"): (689422541, "profiles/r03_fused_pass_pmc_fetch_write.txt"),
    (4096, "fused_prolong"): (666776781, "profiles/r03_fused_pass_pmc_fetch_write.txt"),
})
# end of round 3: the passes with a transfer folded in on 53-line segments (the restricting one at 2 waves / SIMD):
# 2 x 206 175.1 KB + 165 646.8 KB; 2 x 217 409.6 KB + 133 377.5 KB
PMC_TRAFFIC.update({
    (4096, "fused_restrict"): (591868818, "profiles/r03b_fused_pass_pmc_fetch_write.txt"),
    (4096, "fused_prolong"): (581833420, "profiles/r03b_fused_pass_pmc_fetch_write.txt"),
})


def sweep_bytes(n, nnz):
    """Algorithmic HBM bytes of one sweep (Jacobi or residual) -- SURVEY.md 8(d):
    CSR matrix (12 B/nnz + 4 B/row) + x once + b + output."""
    return 12 * nnz + 4 * (n + 1) + 24 * n


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=4096, help="elements per side (nodes = size+1)")
    ap.add_argument("--levels", type=int, default=6)
    ap.add_argument("--nu", type=int, default=3, help="pre/post smoothing steps")
    ap.add_argument("--omega", type=float, default=0.8)
    ap.add_argument("--mode", default="vcycle", choices=["vcycle", "sweep"])
    ap.add_argument("--no-graph", action="store_true", help="launch kernels from Python instead of replaying a hipGraph")
    ap.add_argument("--variant", type=int, default=-1, help="sweep kernel tile variant (-1 = library default)")
    ap.add_argument("--cpu-cycles", type=int, default=10, help="V-cycles timed by the CPU baseline leg (about 10 s of one core)")
    ap.add_argument("--problem", default="poisson", choices=["poisson", "varcoeff", "jittered"],
                    help="poisson = cfg#2/#4 (default), varcoeff = cfg#5, jittered = cfg#3 (7-point)")
    ap.add_argument("--transfer", default="geometric", choices=["geometric", "learned"],
                    help="learned = row-stochastic perturbed L2-type Q per level (cfg#3/#5)")
    ap.add_argument("--rebuild", type=int, default=0, help="time this many numeric Galerkin rebuilds (cfg#5)")
    ap.add_argument("--coarse", default="auto", choices=["auto", "dense", "banded", "bcr"], help="coarsest-level solver")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-tts", action="store_true", help="skip the time-to-solution leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the legs of the other BASELINE configs (#3, #5-style)")
    ap.add_argument("--no-packed", action="store_true", help="plain CSR kernels (no lossless twins at all)")
    ap.add_argument("--no-patterns", action="store_true", help="packed CSR twin only (no row-pattern twin)")
    return ap.parse_args()


def cpu_info():
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    pools = []
    try:
        from threadpoolctl import threadpool_info
        pools = [{k: d.get(k) for k in ("user_api", "internal_api", "num_threads")} for d in threadpool_info()]
    except Exception:
        pass
    return {"model": model, "logical_cpus": os.cpu_count() or 0, "threadpools": pools,
            "note": "the oracle's C loops and SciPy's sparse kernels are single-threaded: 1 core does the work"}


def solve_to_tolerance(H, rhs, nu, omega, abs_tol=1e-10, max_cycles=60):
    """x = 0, then V(nu,nu) cycles (hipGraph replay) until ||b - A x||_2 <= abs_tol -- the scripts' stopping
    test `error=1e-10` (test/test_B_patch.py:193, Multigrid.py:69): cycles, seconds, final relative residual."""
    import torch
    from learnmultigrid_amd import ops
    fine = H.levels[0]
    with torch.cuda.stream(H.stream):
        fine.b.copy_(torch.from_numpy(rhs.ravel().copy()).to(H.device))
        g = H.captured_cycle("Jacobi", nu, omega, "lexicographic")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.zero(fine.x)
        r0 = H.residual_norm(want_vector=False)
        r, cycles = r0, 0
        while r > abs_tol and cycles < max_cycles:
            g.launch()
            cycles += 1
            r = H.residual_norm(want_vector=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return cycles, dt, (r / r0 if r0 > 0 else 0.0)


def cpu_baseline(A, hier, rhs, args):
    """CPU oracle (port of the SciPy/pyamg loops the reference calls, setup hoisted like
    on the GPU) on the SAME matrix and hierarchy, single thread."""
    from oracle import vcycle_ref as V
    n = A.shape[0]
    t0 = time.perf_counter()
    H = V.HoistedVCycle(A, hier)
    t_setup = time.perf_counter() - t0
    x = np.zeros(n)
    b = rhs.ravel()
    if args.mode == "sweep":
        from oracle import kernels as K
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 8.0:
            x = K.jacobi(H.A[0], x, b, args.omega)
            K.residual(H.A[0], x, b)
            reps += 1
        dt = time.perf_counter() - t0
        sweeps = 2 * reps
        what = "%d x (Jacobi sweep + residual) on the fine level" % reps
    else:
        t0 = time.perf_counter()
        for _ in range(args.cpu_cycles):
            x = H.cycle(x, b, "Jacobi", args.nu, args.omega)
        dt = time.perf_counter() - t0
        sweeps = (2 * args.nu + 1) * args.cpu_cycles
        what = "%d full V(%d,%d) cycles, %d levels" % (args.cpu_cycles, args.nu, args.nu, len(hier) + 1)
    out = {"value": n * sweeps / dt, "unit": "DoF*sweeps/s", "cores": 1, "kind": "port",
           "sample": "%s of the same %d-DoF problem in %.1f s (setup %.1f s not counted), "
                     "oracle/ single thread; host has %d logical CPUs"
                     % (what, n, dt, t_setup, os.cpu_count() or 0),
           "cpu": cpu_info(), "setup_s": t_setup,
           "s_per_cycle": dt / args.cpu_cycles if args.mode == "vcycle" else None}
    if args.mode == "vcycle":
        # for honesty: ONE cycle the way the reference really runs it -- forward
        # Gauss-Seidel whatever the smoother argument says, transfer lookup, R A P and the SuperLU
        # factorisation all inside the cycle (Multigrid.py:77-124) -- at cfg#2 size
        from learnmultigrid_amd import problems as P
        m2 = 512
        A2, rhs2 = P.poisson_2d_structured(m2)
        ref = V.RefMultigrid(A2, rhs2.reshape(-1, 1), hierarchy=P.geometric_hierarchy_2d(m2 + 1, 3))
        u0 = np.zeros((A2.shape[0], 1))
        t0 = time.perf_counter()
        ref.v_cycle(ref.matrix, u0, ref.rhs, "Jacobi", args.nu, 1e-10, 3, first_call=True)
        t_ref = time.perf_counter() - t0
        out["reference_style_cycle_cfg2"] = {
            "ms_per_cycle": t_ref * 1e3, "DoF_sweeps_per_s": A2.shape[0] * (2 * args.nu + 1) / t_ref,
            "what": "one un-hoisted as-shipped V(%d,%d) cycle, 513^2 DoF, 3 levels: forward GS smoothing, "
                    "Galerkin products and SuperLU spsolve inside the cycle (oracle/vcycle_ref.py RefMultigrid)"
                    % (args.nu, args.nu)}
    return out


def other_config_leg(label, problem, size, levels, nu, omega, dev, steps=8, warmup=2):
    """One more BASELINE config on the same GPU in the same run: hierarchy with learned-like (row-stochastic perturbed
    L2-type) transfers, V(nu,nu) weighted-Jacobi cycle replayed from a hipGraph; K timed cycles between synchronisations."""
    import scipy.sparse as sp
    import torch
    from learnmultigrid_amd import ops, problems as P
    from learnmultigrid_amd.hierarchy import Hierarchy
    if problem == "jittered":
        A, rhs = P.jittered_poisson_2d(size, seed=42)
    else:
        A, rhs = P.variable_coeff_poisson_2d_structured(size, seed=44)
    hier = []
    for li, sz in enumerate(P.level_sizes(size + 1, levels)[:-1]):
        l2 = P.pseudo_l2_interpolator_1d(sz)
        hier.append(P.learned_like(sp.kron(l2, l2).tocsr(), 43 + li))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    H = Hierarchy(A, hier, dev)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    fine = H.levels[0]
    with torch.cuda.stream(H.stream):
        fine.b.copy_(torch.from_numpy(rhs.ravel().copy()).to(dev))
        g = H.captured_cycle("Jacobi", nu, omega, "lexicographic")
        ops.zero(fine.x)
        r0 = H.residual_norm(want_vector=False)
        for _ in range(warmup):
            g.launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.launch()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        r1 = H.residual_norm(want_vector=False)
    n = A.shape[0]
    out = {"workload": label, "n": int(n), "nnz": int(A.nnz), "level_sizes": [int(v) for v in H.sizes],
           "ms_per_cycle": dt * 1e3, "DoF_sweeps_per_s": n * (2 * nu + 1) / dt, "cycles_timed": steps, "setup_s": setup_s,
           "fine_level_format": ops._fused_kind(fine.A) or ("stencil" if fine.A.stencil is not None else
                                                            ("pcsr" if fine.A.packed is not None else "csr")),
           "coarse_level_formats": [("sell" if lev.A.sell is not None else ("pcsr" if lev.A.packed is not None else
                                                                            ("stencil" if lev.A.stencil is not None else "csr")))
                                    for lev in H.levels[1:-1]],
           "coarse_solver": H.coarse.kind, "coarse_refine": H.coarse_refine,
           "residual_reduction_over_%d_cycles" % (steps + warmup): (r1 / r0 if r0 > 0 else 0.0)}
    del H
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from learnmultigrid_amd import ops, problems as P
    from learnmultigrid_amd.hierarchy import Hierarchy

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test knobs: LMG_FORCE_DEVICE puts every rank on one GPU and LMG_DIST_BACKEND=gloo moves the
    # messages through the host (RCCL refuses two ranks per device) -- rehearsal of the N > 1 path
    # on a one-GPU box; the driver's runs use neither
    if os.environ.get("LMG_FORCE_DEVICE"):
        local = int(os.environ["LMG_FORCE_DEVICE"])
    backend = os.environ.get("LMG_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.zeros(1, device=dev)                     # HIP context of the process: not part of the hierarchy setup
    torch.cuda.synchronize()
    force_dist = os.environ.get("LMG_FORCE_DIST") == "1"     # drive the partitioned path on 1 GPU
    if world > 1 or force_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.variant >= 0:
        ops.tune_set("sweep_variant", args.variant)
    if args.no_packed:
        ops.set_packed_enabled(False)
    if args.no_patterns:
        ops.set_patterns_enabled(False)
    if os.environ.get("LMG_RPAT_NT_ROWS"):
        ops.tune_set("rpat_variant", int(os.environ["LMG_RPAT_NT_ROWS"]))     # tuning knob (>= 1000: threshold)

    m, levels, nu = args.size, args.levels, args.nu
    if args.problem == "poisson":
        A, rhs = P.poisson_2d_structured(m)
    elif args.problem == "varcoeff":
        A, rhs = P.variable_coeff_poisson_2d_structured(m, seed=44)
    else:
        A, rhs = P.jittered_poisson_2d(m, seed=42)
    if args.transfer == "geometric":
        hier = P.geometric_hierarchy_2d(m + 1, levels)
    else:
        import scipy.sparse as sp
        hier = []
        for li, sz in enumerate(P.level_sizes(m + 1, levels)[:-1]):
            l2 = P.pseudo_l2_interpolator_1d(sz)
            hier.append(P.learned_like(sp.kron(l2, l2).tocsr(), 43 + li))
    n, nnz = A.shape[0], A.nnz

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1 or force_dist:
        from learnmultigrid_amd.dist import DistributedVCycle
        t0 = time.perf_counter()
        D = DistributedVCycle.from_problem(A, hier, dev, grid_side=m + 1,
                                           halo_depth=int(os.environ.get("LMG_HALO_DEPTH", 2 * nu + 2)))
        setup_s = time.perf_counter() - t0
        D.set_rhs(rhs)
        D.use_tail_graph = not args.no_graph
        stream = D.stream
        with torch.cuda.stream(stream):
            step = D.make_step("Jacobi", nu, args.omega, graph=not args.no_graph)
        H = D.local_hierarchy
        fine_A = D.fine_local_matrix
        n_loc_fine = D.fine_local_rows
    else:
        t0 = time.perf_counter()
        H = Hierarchy(A, hier, dev, coarse_solver=args.coarse)
        torch.cuda.synchronize()
        setup_s = time.perf_counter() - t0
        fine = H.levels[0]
        fine.b.copy_(torch.from_numpy(rhs.ravel().copy()).to(dev))
        stream = H.stream
        stream.wait_stream(torch.cuda.current_stream())
        fine_A = fine.A
        n_loc_fine = n
        with torch.cuda.stream(stream):
            if args.mode == "sweep":
                r = torch.empty_like(fine.x)

                def step():
                    ops.csr_jacobi(fine.A, fine.x, fine.b, args.omega, fine.tmp)
                    fine.x, fine.tmp = fine.tmp, fine.x
                    ops.csr_residual_norm2(fine.A, fine.x, fine.b, r, H.partials, H.norm2)
            elif args.no_graph:
                def step():
                    H.cycle("Jacobi", nu, args.omega)
            else:
                g = H.captured_cycle("Jacobi", nu, args.omega, "lexicographic")
                step = g.launch

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        sweeps_per_step = 2 if args.mode == "sweep" else 2 * nu + 1
        value = n * sweeps_per_step * args.steps / dt

        # ---- roofline of the dominant kernel (fine-level Jacobi sweep), HIP events on this stream
        xa = torch.zeros(fine_A.shape[1], dtype=torch.float64, device=dev)
        ya = torch.zeros(fine_A.shape[0], dtype=torch.float64, device=dev)
        ba = torch.ones(fine_A.shape[0], dtype=torch.float64, device=dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 40
        for _ in range(3):
            ops.csr_jacobi(fine_A, xa, ba, args.omega, ya)
        torch.cuda.synchronize()
        ev0.record(stream)
        for _ in range(reps):
            ops.csr_jacobi(fine_A, xa, ba, args.omega, ya)
        ev1.record(stream)
        torch.cuda.synchronize()
        t_jac = ev0.elapsed_time(ev1) * 1e-3 / reps
        # the same sweep through the plain CSR kernel (what a CSR stream can reach), and the
        # device copy ceiling of this GPU, both from this run
        t_csr = None
        if ops._PACKED_ENABLED and (fine_A.patterns is not None or fine_A.packed is not None):
            ops.set_packed_enabled(False)
            for _ in range(2):
                ops.csr_jacobi(fine_A, xa, ba, args.omega, ya)
            ev0.record(stream)
            for _ in range(10):
                ops.csr_jacobi(fine_A, xa, ba, args.omega, ya)
            ev1.record(stream)
            torch.cuda.synchronize()
            t_csr = ev0.elapsed_time(ev1) * 1e-3 / 10
            ops.set_packed_enabled(True)
        cp_src = torch.zeros(1 << 27, dtype=torch.float64, device=dev)       # 1 GiB
        cp_dst = torch.empty_like(cp_src)
        for _ in range(2):
            ops.copy(cp_src, cp_dst)
        ev0.record(stream)
        for _ in range(10):
            ops.copy(cp_src, cp_dst)
        ev1.record(stream)
        torch.cuda.synchronize()
        copy_gbps = 2 * cp_src.numel() * 8 / (ev0.elapsed_time(ev1) * 1e-3 / 10) / 1e9
        for _ in range(2):
            cp_dst.copy_(cp_src)
        ev0.record(stream)
        for _ in range(10):
            cp_dst.copy_(cp_src)                    # the runtime's own device-to-device copy
        ev1.record(stream)
        torch.cuda.synchronize()
        copy_gbps = max(copy_gbps, 2 * cp_src.numel() * 8 / (ev0.elapsed_time(ev1) * 1e-3 / 10) / 1e9)
        for _ in range(2):
            ops.axpby(1.0, cp_src, 0.0, cp_dst)
        ev0.record(stream)
        for _ in range(10):
            ops.axpby(1.0, cp_src, 0.0, cp_dst)     # our own 16-B-per-lane streaming kernel as a copy
        ev1.record(stream)
        torch.cuda.synchronize()
        copy_gbps = max(copy_gbps, 2 * cp_src.numel() * 8 / (ev0.elapsed_time(ev1) * 1e-3 / 10) / 1e9)
        del cp_src, cp_dst
    B = sweep_bytes(fine_A.shape[0], fine_A.nnz)
    st_ = fine_A.stencil if (ops._PACKED_ENABLED and ops._STENCIL_ENABLED) else None
    rp_ = fine_A.patterns if ops._PACKED_ENABLED else None
    pk = fine_A.packed if (ops._PACKED_ENABLED and rp_ is None) else None
    sl = fine_A.sell if (ops._PACKED_ENABLED and rp_ is None and pk is None) else None
    kind = "stencil" if st_ is not None else ("rpat" if rp_ is not None else ("pcsr" if pk is not None else
                                                                              ("sell" if sl is not None else "csr")))
    # bytes the launch really has to move: the lossless twin of the CSR arrays (DESIGN.md
    # section 3) + x once + b + output
    twin = st_ if st_ is not None else (rp_ if rp_ is not None else (pk if pk is not None else sl))
    nrow = fine_A.shape[0]
    B_moved = (twin.bytes() if twin is not None else fine_A.bytes()) + 24 * nrow
    if st_ is not None:
        kname = ("stencil_sweep_kernel<JACOBI> (grid stencil: %d patterns, line stride %d, 1 B/row of matrix)"
                 % (st_.npat, st_.W))
    elif rp_ is not None:
        kname = "rpat_sweep_kernel<JACOBI> (row patterns: %d distinct rows, %d entries, 1 B/row)" % (rp_.npat, rp_.nent)
    elif pk is not None:
        kname = ("pcsr_sweep_kernel<JACOBI> (packed CSR: colmode %d, valmode %d, %d dictionary values)"
                 % (pk.colmode, pk.valmode, pk.ndict))
    elif sl is not None:
        kname = "sell_sweep_kernel<JACOBI> (sliced ELL, colmode %d)" % sl.colmode
    else:
        kname = "csr_sweep_kernel<JACOBI>"
    pmc = PMC_TRAFFIC.get((args.size, kind)) if (world == 1 and not force_dist and args.problem == "poisson") else None
    achieved = B_moved / t_jac / 1e9
    roofline = {"bound": "hbm",
                "kernel": kname,
                "what": "ONE fine-level weighted-Jacobi sweep (the kernel the north-star target is quoted on), HIP events "
                        "around %d launches on the launch stream; achieved = bytes the kernel has to move / avg launch" % reps,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "moved_bytes_per_launch": B_moved,
                "traffic": None if pmc is None else pmc[0],
                "traffic_source": None if pmc is None else
                "%s (separate rocprofv3 --pmc passes of this kernel, NOT this run: 2 x FETCH_SIZE + WRITE_SIZE)" % pmc[1],
                "avg_launch_ms": t_jac * 1e3, "rows_per_launch": int(nrow),
                "copy_ceiling_GBps": copy_gbps, "frac_of_copy_ceiling": achieved / copy_gbps,
                # the same sweep priced in the CSR bytes of SURVEY.md 8(d): what a CSR stream would have to
                # sustain to be as fast -- a rate of WORK, not of bytes moved, hence not a roofline fraction
                "csr_equivalent_bytes_per_launch": B, "csr_equivalent_GBps": B / t_jac / 1e9,
                "byte_reduction_vs_csr": B / B_moved,
                "csr_kernel_same_sweep": None if t_csr is None else {
                    "kernel": "csr_sweep_kernel<JACOBI>", "avg_launch_ms": t_csr * 1e3, "GBps": B / t_csr / 1e9,
                    "frac_of_peak": B / t_csr / 1e9 / HBM_PEAK_GBS, "frac_of_copy_ceiling": B / t_csr / 1e9 / copy_gbps}}
    # the fused smoothing passes the cycle really launches on this level (3 sweeps [+ residual] per pass)
    if world == 1 and not force_dist and args.mode == "vcycle" and ops.stencil_smooth_available(fine_A):
        with torch.cuda.stream(stream):
            ra = torch.empty_like(ya)
            fused = {}
            lev0 = H.levels[0]
            with_corr = ops.stencil_smooth_prolong_available(fine_A, lev0.P)
            with_rest = ops.stencil_smooth_restrict_available(fine_A, lev0.R)
            ec = torch.zeros(lev0.P.shape[1], dtype=torch.float64, device=dev)
            bcv = torch.zeros_like(ec)
            k3 = min(nu, 3)
            for lab, k_, r_, corr, rest in (
                    ("pre_smoothing_%d_sweeps_plus_%s" % (k3, "restricted_residual" if with_rest else "residual"), k3,
                     None if with_rest else ra, None, (lev0.R, bcv) if with_rest else None),
                    ("post_smoothing_%d_sweeps%s" % (k3, "_with_correction" if with_corr else ""), k3, None,
                     (lev0.P, ec) if with_corr else None, None)):
                for _ in range(2):
                    ops.stencil_smooth(fine_A, xa, ba, args.omega, k_, ya, r_, prolong=corr, restrict=rest)
                ev0.record(stream)
                for _ in range(20):
                    ops.stencil_smooth(fine_A, xa, ba, args.omega, k_, ya, r_, prolong=corr, restrict=rest)
                ev1.record(stream)
                torch.cuda.synchronize()
                tf = ev0.elapsed_time(ev1) * 1e-3 / 20
                # ids + x + b + out (+ r) (+ the ids of P and the coarse vector once) (+ the ids of R and the coarse rhs)
                moved = (nrow * (1 + 24 + (8 if r_ is not None else 0)) + (nrow + 8 * ec.numel() if corr is not None else 0)
                         + (9 * ec.numel() if rest is not None else 0))
                has_resid = r_ is not None or rest is not None
                napply = k_ + (1 if has_resid else 0) + (1 if (corr is not None or rest is not None) else 0)
                key = "fused_restrict" if rest is not None else ("fused_resid" if r_ is not None else
                                                                 ("fused_prolong" if corr is not None else "fused"))
                pm = PMC_TRAFFIC.get((args.size, key)) if (args.problem == "poisson" and k_ == 3) else None
                fused[lab] = {"kernel": "stencil_fused_kernel" + ("<PROL>: x + P e formed on the fly" if corr is not None else
                                                                  ("<REST>: R r formed on the fly, r not written" if rest is not None else "")),
                              "avg_launch_ms": tf * 1e3,
                              "traffic": None if pm is None else pm[0],
                              "traffic_source": None if pm is None else "%s (separate rocprofv3 --pmc passes, not this run)" % pm[1],
                              "operator_applications_per_launch": napply,
                              "compulsory_bytes_per_launch": moved, "GBps": moved / tf / 1e9,
                              "frac": moved / tf / 1e9 / HBM_PEAK_GBS,
                              "ms_per_operator_application": tf * 1e3 / napply,
                              "separate_launches_would_take_ms": (k_ + (1 if has_resid else 0)) * t_jac * 1e3}
            # the top-level object describes the DOMINANT KERNEL OF THE TIMED STEP (the slower of the two fused passes);
            # the single sweep -- the kernel the north-star target is quoted on, not launched by the cycle -- goes below it
            single = roofline
            dom_key = max(fused, key=lambda k_: fused[k_]["avg_launch_ms"])
            dom = fused[dom_key]
            roofline = {"bound": "hbm", "kernel": dom["kernel"], "pass": dom_key,
                        "what": "the dominant kernel of the timed V-cycle: the fused fine-level pass (%d operator applications "
                                "of %d rows in one launch), HIP events around 20 launches on the launch stream; achieved = "
                                "bytes the launch has to move (ids + x + b + out + coarse vector / ids) / avg launch"
                                % (dom["operator_applications_per_launch"], nrow),
                        "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
                        "bytes_per_launch": dom["compulsory_bytes_per_launch"],
                        "traffic": dom["traffic"], "traffic_source": dom["traffic_source"],
                        "traffic_over_bytes": None if dom["traffic"] is None else dom["traffic"] / dom["compulsory_bytes_per_launch"],
                        "avg_launch_ms": dom["avg_launch_ms"], "rows_per_launch": int(nrow),
                        "share_of_timed_step": dom["avg_launch_ms"] / (dt / args.steps * 1e3),
                        "operator_applications_per_launch": dom["operator_applications_per_launch"],
                        "csr_equivalent_GBps": dom["operator_applications_per_launch"] * B / (dom["avg_launch_ms"] * 1e-3) / 1e9,
                        "copy_ceiling_GBps": copy_gbps, "frac_of_copy_ceiling": dom["GBps"] / copy_gbps,
                        "fused_passes_in_the_cycle": fused, "single_sweep": single}
    cyc_bytes, coarse_bytes = H.cycle_bytes(nu) if args.mode == "vcycle" and world == 1 else (None, None)
    if world > 1 or force_dist:
        out_extra = {"distributed_levels": D.n_dist, "rows_per_rank_fine": n_loc_fine,
                     "halo_values_fine": D.dl[0].n_lo + D.dl[0].n_hi, "halo_depth": D.halo_depth,
                     "halo_exchanges_per_cycle": D.n_exchanges / max(1, args.steps + args.warmup)}
    else:
        out_extra = {}

    rebuild_ms = None
    if args.rebuild:
        # replicated numeric Galerkin rebuild; with several ranks every rank rebuilds its replica and
        # re-cuts its local operators (no communication)
        dist_run = world > 1 or force_dist
        owner = D if dist_run else H
        newv = H.levels[0].A.vals.clone()
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            t0 = time.perf_counter()
            owner.rebuild_numeric(newv)  # first rebuild: sorts once more and records the product maps
            torch.cuda.synchronize()
            rebuild_first_ms = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            for _ in range(args.rebuild):
                owner.rebuild_numeric(newv)
            torch.cuda.synchronize()
            rebuild_ms = (time.perf_counter() - t0) / args.rebuild * 1e3
    out = {"metric": "fine-level DoF*sweeps/s, 2-D Poisson V-cycle", "value": value,
           "unit": "DoF*sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "%s %dx%d elements (%d DoF, %d nnz, CSR fp64/int32), %d-level V(%d,%d) cycle, "
                                  "weighted Jacobi omega=%.2f, %s, Galerkin RAP by device SpGEMM"
                                  % ({"poisson": "2-D structured P1 Poisson",
                                      "varcoeff": "2-D variable-coefficient P1 stiffness (k = exp(0.5 N(0,1)) smoothed, seed 44)",
                                      "jittered": "2-D P1 Poisson on a jittered triangulation (7-point, seed 42)"}[args.problem],
                                     m, m, n, nnz, levels, nu, nu, args.omega,
                                     {"geometric": "tensor-product geometric transfer",
                                      "learned": "learned-like transfer (row-stochastic perturbed L2-type Q, seeds 43+level)"}[args.transfer]),
                      "mode": args.mode, "step": "one full V-cycle" if args.mode == "vcycle"
                      else "1 Jacobi sweep + 1 residual on the fine level",
                      "fine_sweeps_per_step": sweeps_per_step,
                      "hipgraph": (not args.no_graph) and world == 1 and not force_dist and args.mode == "vcycle",
                      "level_sizes": [int(s) for s in P.level_sizes(m + 1, levels)],
                      "partition": "row blocks of grid lines over %d rank(s)" % world,
                      "sweep_variant": ops.tune_get("sweep_variant"),
                      "packed_csr": not args.no_packed, "fine_level_format": kind},
           "setup_s": setup_s, "roofline": roofline}
    out["config"].update(out_extra)
    if world > 1:
        out["scaling"] = "strong"                 # cfg#4 is a fixed-size problem cut into row blocks
    if rebuild_ms is not None:
        out["galerkin_rebuild_ms"] = rebuild_ms
        out["galerkin_rebuild_first_ms"] = rebuild_first_ms
        out["galerkin_product_map_bytes"] = sum(lev.plan_RA.recorded_bytes() + lev.plan_RAP.recorded_bytes()
                                                for lev in H.levels[:-1])
    out["config"]["problem"] = args.problem
    out["config"]["transfer"] = args.transfer
    if cyc_bytes is not None:
        out["cycle_algorithmic_GBps"] = cyc_bytes / (dt / args.steps) / 1e9
        out["cycle_algorithmic_bytes"] = cyc_bytes
        out["coarse_dense_bytes"] = coarse_bytes
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(A, hier, rhs, args)
    if rank == 0 and world == 1 and not force_dist and args.mode == "vcycle" and not args.no_tts:
        # time to solution (||r||_2 <= 1e-10 from x = 0, the scripts' `error=1e-10`): upload + setup + cycles, next to the CPU oracle's
        # setup + the same number of its timed cycles (same arithmetic, hence the same cycle count)
        cyc, t_solve, rel = solve_to_tolerance(H, rhs, nu, args.omega)
        tts = {"this_config": {"cycles": cyc, "final_relative_residual": rel,
                               "setup_s_cold_incl_upload": setup_s, "solve_s": t_solve,
                               "time_to_solution_s": setup_s + t_solve}}
        cb = out.get("cpu_baseline")
        if cb and cb.get("s_per_cycle"):
            tts["this_config"]["cpu_time_to_solution_s"] = cb["setup_s"] + cyc * cb["s_per_cycle"]
            tts["this_config"]["cpu_note"] = "oracle setup + %d x its measured cycle time (1 core)" % cyc
        if args.problem == "poisson" and args.transfer == "geometric" and args.size != 512:
            m2 = 512                                                             # cfg#2: 513^2, 3 levels
            A2, rhs2 = P.poisson_2d_structured(m2)
            t0 = time.perf_counter()
            H2 = Hierarchy(A2, P.geometric_hierarchy_2d(m2 + 1, 3), dev)
            torch.cuda.synchronize()
            s2 = time.perf_counter() - t0
            c2, ts2, rel2 = solve_to_tolerance(H2, rhs2, nu, args.omega)
            tts["cfg2_513x513_3_levels"] = {"cycles": c2, "final_relative_residual": rel2, "setup_s_warm_incl_upload": s2,
                                            "solve_s": ts2, "time_to_solution_s": s2 + ts2}
            del H2
        out["time_to_solution"] = tts
        # the same hierarchy once more in this process: the WARM setup time (code objects loaded, allocator primed)
        if args.problem == "poisson" and args.transfer == "geometric":
            del H
            torch.cuda.empty_cache()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            Hw = Hierarchy(A, hier, dev, coarse_solver=args.coarse)
            torch.cuda.synchronize()
            tts["this_config"]["setup_s_warm_incl_upload"] = time.perf_counter() - t0
            del Hw
            torch.cuda.empty_cache()
    if (rank == 0 and world == 1 and not force_dist and args.mode == "vcycle" and not args.no_extra
            and args.problem == "poisson" and args.transfer == "geometric" and args.size == 4096):
        H = None
        torch.cuda.empty_cache()
        # the same cfg#4 cycle with the reference's SHIPPED semantics: forward Gauss-Seidel whatever the smoother's name
        # (Multigrid.py:79-88, :121), exact lexicographic order on the device
        Hs = Hierarchy(A, hier, dev, coarse_solver=args.coarse)
        with torch.cuda.stream(Hs.stream):
            Hs.levels[0].b.copy_(torch.from_numpy(rhs.ravel().copy()).to(dev))
            gs_graph = Hs.captured_cycle("GaussSeidel", nu, 1.0, "lexicographic")
            gs_graph.launch()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                gs_graph.launch()
            torch.cuda.synchronize()
            t_gs = (time.perf_counter() - t0) / 3
            Hs.check_smoothers()
        del Hs, gs_graph
        torch.cuda.empty_cache()
        shipped = {"workload": "cfg#4 with the reference's shipped smoother semantics: V(%d,%d), exact forward (lexicographic) "
                               "Gauss-Seidel on every level, hipGraph replay" % (nu, nu),
                   "ms_per_cycle": t_gs * 1e3, "DoF_sweeps_per_s": n * (2 * nu + 1) / t_gs, "cycles_timed": 3}
        out["other_configs"] = {
            "cfg4_as_shipped_forward_gauss_seidel": shipped,
            "cfg3_jittered_7pt_1441x1441_learned_q_5_levels": other_config_leg(
                "BASELINE cfg#3: P1 Poisson on a jittered triangulation, 1441^2 = 2.08 M DoF, learned-like Q, 5 levels",
                "jittered", 1440, 5, nu, args.omega, dev),
            "cfg5_style_varcoeff_4097x4097_learned_q_6_levels": other_config_leg(
                "BASELINE cfg#5 at 4097^2 (one GPU's share of the 8193^2 problem is half of this): variable-coefficient "
                "stiffness, learned-like Q, 6 levels", "varcoeff", 4096, 6, nu, args.omega, dev, steps=5)}
    if rank == 0:
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
