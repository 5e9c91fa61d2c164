"""Host-side mirror of ``RevisedPrimalSimplexSolver``
(Simplex/RevisedPrimalSimplexSolver.cs:10-449): same constructor arguments, public members and
exceptions (same messages); the numeric work runs on the MI355X through the C ABI."""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import _native as N
from .engine import Engine, RevisedState, default_engine
from .input_file_parser import Constraint

# the C# `throw new Exception(...)` texts (:91, :179, :183, :267)
MESSAGES = {
    N.LPR_INFEASIBLE_BASIS: "Infeasible basis (negative basic value).",
    N.LPR_UNBOUNDED: "Unbounded problem (no positive component in direction).",
    N.LPR_ENTERING_ALREADY_BASIC: "Internal error: entering variable is already basic.",
    N.LPR_PIVOT_TOO_SMALL: "Pivot too small.",
}


class SolverException(Exception):
    """Stands for the plain System.Exception the reference throws; ``status`` is the lpr_status."""

    def __init__(self, status: int):
        super().__init__(MESSAGES.get(status, f"solver status {status}"))
        self.status = status


class RevisedPrimalSimplexSolver:
    def __init__(self, objective: Sequence[float], constraints: Sequence[Constraint],
                 isMinimization: bool, *, engine: Optional[Engine] = None):
        if objective is None or len(objective) == 0:  # :43
            raise ValueError("Objective cannot be null or empty.")
        if constraints is None or len(constraints) == 0:  # :44
            raise ValueError("Constraints cannot be null or empty.")
        self.numVariables = n = len(objective)
        self.numConstraints = m = len(constraints)
        self.isMinimization = isMinimization
        A = np.zeros((m, n), dtype=np.float64)
        b = np.zeros(m, dtype=np.float64)
        for i, c in enumerate(constraints):
            if len(c.Coefficients) != n:  # :57-58
                raise ValueError(f"Constraint {i + 1} has incorrect number of coefficients.")
            A[i, :] = c.Coefficients
            b[i] = c.RHS  # Relation is never read by the reference (:55-61)
        self._engine = engine or default_engine()
        self._st = RevisedState.create(self._engine, list(objective), A, b, isMinimization)
        self.IterationSnapshots: List[str] = []
        self.FinalZ: float = 0.0
        self.SolutionVector: List[float] = []
        self.Status: Optional[int] = None

    def Solve(self, max_pivots: int = 0) -> None:  # :82-251
        res = self._st.solve(max_pivots=max_pivots)
        self.Status = res.status
        if res.status == N.LPR_OK_OPTIMAL:  # :124-146
            x, z = self._st.solution()
            self.SolutionVector = [float(v) for v in x]
            self.FinalZ = z
        elif res.status in MESSAGES:
            raise SolverException(res.status)
        # LPR_PIVOT_LIMIT has no C# counterpart (`while (true)`, :86)

    @property
    def BasicVariables(self) -> List[int]:  # :39
        return [int(v) for v in self._st.basis()]

    @property
    def PivotLog(self) -> np.ndarray:
        """(leavingRow 0-based, entering variable, leaving variable) per iteration."""
        return self._st.log()

    @property
    def state(self) -> RevisedState:
        return self._st
