#!/usr/bin/env python3
"""Headline benchmark: fine-level DoF.sweeps/s of the 2-D Poisson V-cycle path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--size 4096] [--mode sweep|vcycle]

A "step" is one pass of the hot path over the synthetic fine grid:
  mode sweep  : one weighted-Jacobi sweep + one residual SpMV (with fused ||r||^2) on the
                fine level of cfg#4 (4097^2 DoF 5-point P1 Poisson, CSR fp64/int32);
  mode vcycle : one full V(nu,nu) cycle over all levels (added once the hierarchy is built).
Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel:
the fine-level Jacobi sweep, timed with HIP events on the launch stream) and
`cpu_baseline` (the CPU oracle on the same matrix, bounded sample, 1 core).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s copy-achievable


def sweep_bytes(n, nnz):
    """Algorithmic HBM bytes of one fine-level sweep (Jacobi or residual), SURVEY.md 8(d):
    CSR matrix (12 B/nnz + 4 B/row) + x once + b + output."""
    return 12 * nnz + 4 * (n + 1) + 24 * n


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=4096, help="elements per side (nodes = size+1)")
    ap.add_argument("--mode", default="sweep", choices=["sweep"])
    ap.add_argument("--omega", type=float, default=0.8)
    ap.add_argument("--rpt", type=int, default=0, help="sweep kernel rows/thread (0 = library default)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    return ap.parse_args()


def cpu_baseline(A, rhs, omega, budget_s):
    """The CPU oracle (oracle/lmg_oracle.c = port of the SciPy/pyamg loops the reference
    calls) on the SAME matrix: alternating Jacobi sweep and residual, single thread."""
    from oracle import kernels as K
    A = K.as_csr(A)
    n = A.shape[0]
    x = np.zeros(n)
    b = rhs.ravel()
    K.jacobi(A, x, b, omega)                       # warm the caches / page in
    t0 = time.perf_counter()
    sweeps = 0
    while True:
        x = K.jacobi(A, x, b, omega)
        K.residual(A, x, b)
        sweeps += 2
        dt = time.perf_counter() - t0
        if dt >= budget_s or sweeps >= 200:
            break
    return {"value": n * sweeps / dt, "unit": "DoF*sweeps/s", "cores": 1, "kind": "port",
            "sample": "%d sweeps (Jacobi+residual alternating) of the same %d-DoF matrix, %.1f s, "
                      "oracle/lmg_oracle.c single thread; host has %d logical CPUs"
                      % (sweeps, n, dt, os.cpu_count() or 0),
            "GBps": sweep_bytes(n, A.nnz) * sweeps / dt / 1e9}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from learnmultigrid_amd import ops, problems as P

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group("nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.rpt:
        ops.tune_set("sweep_rpt", args.rpt)

    m = args.size
    A, rhs = P.poisson_2d_structured(m)
    n, nnz = A.shape[0], A.nnz
    dA = ops.DeviceCSR.from_scipy(A, dev, canonical=False)
    b = torch.from_numpy(rhs.ravel().copy()).to(dev)
    x = torch.zeros(n, dtype=torch.float64, device=dev)
    y = torch.empty_like(x)
    r = torch.empty_like(x)
    part = torch.empty(ops.partials_count(n), dtype=torch.float64, device=dev)
    n2 = torch.empty(1, dtype=torch.float64, device=dev)

    def step():
        nonlocal x, y
        ops.csr_jacobi(dA, x, b, args.omega, y)
        x, y = y, x
        ops.csr_residual_norm2(dA, x, b, r, part, n2)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    sweeps_per_step = 2
    value = world * n * sweeps_per_step * args.steps / dt

    # ---- roofline of the dominant kernel: fine-level Jacobi sweep, HIP events on the launch stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(20, args.steps)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        ops.csr_jacobi(dA, x, b, args.omega, y)
        x, y = y, x
    ev1.record()
    torch.cuda.synchronize()
    t_jac = ev0.elapsed_time(ev1) * 1e-3 / reps
    ev0.record()
    for _ in range(reps):
        ops.csr_residual_norm2(dA, x, b, r, part, n2)
    ev1.record()
    torch.cuda.synchronize()
    t_res = ev0.elapsed_time(ev1) * 1e-3 / reps
    B = sweep_bytes(n, nnz)
    achieved = B / t_jac / 1e9
    roofline = {"bound": "hbm", "kernel": "csr_sweep_kernel<JACOBI>", "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "algorithmic_bytes_per_launch": B,
                "avg_launch_ms": t_jac * 1e3,
                "residual_kernel_GBps": B / t_res / 1e9, "residual_avg_launch_ms": t_res * 1e3}

    out = {"metric": "fine-level DoF*sweeps/s (2-D Poisson V-cycle path)", "value": value,
           "unit": "DoF*sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "cfg#4 fine level: 2-D structured P1 Poisson %dx%d elements "
                                  "(%d DoF, %d nnz CSR fp64/int32); step = 1 weighted-Jacobi sweep "
                                  "(omega=%.2f) + 1 residual SpMV with fused norm" % (m, m, n, nnz, args.omega),
                      "mode": args.mode, "sweeps_per_step": sweeps_per_step,
                      "sweep_rpt": ops.tune_get("sweep_rpt")},
           "achieved_GBps": B * sweeps_per_step * args.steps / dt / 1e9 * world,
           "roofline": roofline}
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(A, rhs, args.omega, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
