"""sublinear_time_solver_amd — MI355X-native push / Neumann-series solver for diagonally dominant Ax = b.

One hot path of ruvnet/sublinear-time-solver rebuilt for gfx950: hand-written HIP kernels behind a
C ABI (include/sublinear_hip.h, csrc/), plus this host-side mirror of the reference's solver interfaces.
"""
from ._lib import SolverError, load  # noqa: F401
from .solver import (Communicator, ConjugateGradientSolver, GaussSouthwellSolver, NeumannSolver, NeumannState, PushSolver, QuerySession, SolverOptions, SolverResult,  # noqa: F401
                     SparseMatrix, SublinearSolver, estimate_entry, random_walk_solve)
from . import generators, utils  # noqa: F401

__all__ = ["SolverError", "load", "Communicator", "ConjugateGradientSolver", "GaussSouthwellSolver", "NeumannSolver", "NeumannState", "PushSolver", "QuerySession", "SolverOptions", "SolverResult", "SparseMatrix",
           "SublinearSolver", "estimate_entry", "random_walk_solve", "generators", "utils"]
