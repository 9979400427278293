"""Drop-in mirror of learn_multigrid/solvers (same class names and call signatures)."""
from .Solver import Solver, DirectSolver, IterativeSolver
from .Jacobi import Jacobi
from .GaussSeidel import GaussSeidel
from .Multigrid import Multigrid, GeometricMG, SemiGeometricMG, HierarchyMG

__all__ = ["Solver", "DirectSolver", "IterativeSolver", "Jacobi", "GaussSeidel", "Multigrid",
           "GeometricMG", "SemiGeometricMG", "HierarchyMG"]
