"""Randomised bitwise comparisons as part of the -m gpu suite: a few seeded cases of tools/fuzz_fused.py (fused smoothing passes
-- register and tiled kernels, transfers folded in, 1 - 3 sweeps -- against the separate launches) and tests/fuzz_gs.py (LDS-staged
Gauss-Seidel bands, single and pipelined sweeps, 5- and 9-point operators, against the register kernel and the oracle's
sequential sweep).  Each tool runs in its own process and exits non-zero on the first mismatch."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, tool), *args], cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, (tool, p.stdout[-2000:], p.stderr[-2000:])
    return p.stdout


@pytest.mark.gpu
def test_fused_passes_on_random_grids_equal_the_separate_launches():
    out = _run("tools/fuzz_fused.py", "--cases", "8", "--seed", "17", "--max", "900")
    assert "done: 0 mismatches" in out, out[-500:]


@pytest.mark.gpu
def test_lds_gauss_seidel_bands_on_random_grids_equal_the_register_kernel_and_the_oracle():
    out = _run("tests/fuzz_gs.py", "--cases", "10", "--seed", "23", "--max", "900")
    assert "done: 0 problems" in out, out[-500:]
