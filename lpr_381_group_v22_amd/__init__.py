"""MI355X-native simplex pivot engine behind the solver surface of LPR_381_Group_V22.

Importing this package loads ``_lib/liblpr_engine.so`` (hand-written HIP for gfx950, built by
``__graft_entry__.build()``); there is no CPU fallback.
"""
from . import _native
from .engine import Engine, RevisedState, Tableau, default_engine
from .branch_and_bound import (BranchAndBoundAdapter, BranchBoundTree, Comm,
                               solve_level_sync_native, solve_level_synchronous,
                               torch_collectives)
from .input_file_parser import Constraint, InputFileParser
from .primal_simplex_solver import PrimalSimplexSolver
from .revised_primal_simplex_solver import RevisedPrimalSimplexSolver, SolverException

__all__ = [
    "Engine", "Tableau", "default_engine", "Constraint", "InputFileParser",
    "PrimalSimplexSolver", "RevisedPrimalSimplexSolver", "RevisedState", "SolverException",
    "BranchAndBoundAdapter", "BranchBoundTree", "solve_level_synchronous", "torch_collectives",
    "Comm", "solve_level_sync_native",
    "_native",
]
