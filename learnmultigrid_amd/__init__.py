"""MI355X-native multigrid V-cycle (drop-in for learn_multigrid/solvers of
claudiotomasi/LearnMultigrid): Python host code driving hand-written gfx950 HIP
kernels through the C ABI in include/lmg.h.  No CPU fallback exists: using a solver
without the built HIP library (or without a GPU) raises."""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
