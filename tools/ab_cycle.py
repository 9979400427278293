#!/usr/bin/env python3
"""Same-box, same-process A/B of tuning knobs IN THE CYCLE: builds the cfg#4 hierarchy once, captures the
V(3,3) cycle once per setting and times the replays round-robin (a stand-alone timing loop and a different
gpurun box have both pointed the wrong way before; see csrc/stencil_fused.hip).

  python tools/ab_cycle.py --set base --set fused_seg_lines=24 --set py:FUSED_MIN_ROWS=1000000
Settings: `key=value[,key=value...]` for ops.tune_set keys, `py:NAME=value` / `coarse:NAME=value` for module
constants of ops / coarse.
"""
import argparse, ast, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learnmultigrid_amd import coarse, ops, problems as P
from learnmultigrid_amd.hierarchy import Hierarchy

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--levels", type=int, default=6)
ap.add_argument("--nu", type=int, default=3)
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
settings = a.set or ["base"]

A, rhs = P.poisson_2d_structured(a.size)
hier = P.geometric_hierarchy_2d(a.size + 1, a.levels)
dev = torch.device("cuda:0")
H = Hierarchy(A, hier, dev)
fine = H.levels[0]
fine.b.copy_(torch.from_numpy(rhs.ravel().copy()).to(dev))
H.stream.wait_stream(torch.cuda.current_stream())


def apply(setting, undo=False):
    saved = []
    if setting == "base":
        return saved
    for kv in setting.split(","):
        k, v = kv.split("=")
        if k.startswith("py:") or k.startswith("coarse:"):
            mod, name = (ops, k[3:]) if k.startswith("py:") else (coarse, k[7:])
            saved.append((k, getattr(mod, name)))
            setattr(mod, name, ast.literal_eval(v))
        else:
            saved.append((k, ops.tune_get(k)))
            ops.tune_set(k, int(v))
    return saved


def restore(saved):
    for k, v in saved:
        if k.startswith("coarse:"):
            setattr(coarse, k[7:], v)
        elif k.startswith("py:"):
            setattr(ops, k[3:], v)
        else:
            ops.tune_set(k, v)


graphs = {}
with torch.cuda.stream(H.stream):
    for s in settings:
        saved = apply(s)
        H._graphs.clear()
        graphs[s] = H.captured_cycle("Jacobi", a.nu, 0.8, "lexicographic")
        restore(saved)
    times = {s: [] for s in settings}
    for rnd in range(a.rounds):
        for s in settings:
            g = graphs[s]
            for _ in range(3):
                g.launch()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                g.launch()
            torch.cuda.synchronize()
            times[s].append((time.perf_counter() - t0) / a.steps * 1e3)
for s in settings:
    t = sorted(times[s])
    print("%-50s median %.4f ms   min %.4f   max %.4f" % (s, t[len(t) // 2], t[0], t[-1]))
