"""Drop-in mirror of learn_multigrid/solvers (same class names and call signatures)."""
from .Solver import Solver, DirectSolver, IterativeSolver
from .Jacobi import Jacobi
from .GaussSeidel import GaussSeidel
from .CG import CG
from .Multigrid import Multigrid, GeometricMG, SemiGeometricMG, HierarchyMG, NeuralMG

__all__ = ["Solver", "DirectSolver", "IterativeSolver", "Jacobi", "GaussSeidel", "CG", "Multigrid",
           "GeometricMG", "SemiGeometricMG", "HierarchyMG", "NeuralMG"]
