#!/usr/bin/env python3
"""Run one of bench.py's extra legs alone (for rocprofv3 --kernel-trace: which kernels a cfg#3 / cfg#5-style cycle spends its
time in).    python tools/prof_other.py cfg5 | cfg3   [--steps 8]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("which", choices=["cfg3", "cfg5"])
ap.add_argument("--steps", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
if a.which == "cfg3":
    out = bench.other_config_leg("cfg3", "jittered", 1440, 5, 3, 0.8, dev, steps=a.steps)
else:
    out = bench.other_config_leg("cfg5s", "varcoeff", 4096, 6, 3, 0.8, dev, steps=a.steps)
print(json.dumps(out))
