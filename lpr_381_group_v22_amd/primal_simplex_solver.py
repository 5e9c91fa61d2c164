"""Host-side mirror of ``PrimalSimplexSolver`` (Simplex/PrimalSimplexSolver.cs:10-279).

Same constructor arguments, public members and error behaviour as the C# class; every numeric
step (tableau build, entering/leaving selection, pivot, solution extraction) runs on the MI355X
through the C ABI (include/lpr_engine.h).  The tableau stays in HBM for the whole solve; the host
sees it only when a caller asks for ``FinalTableau`` / snapshots.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import _native as N
from . import table_iteration_formater as fmt
from .engine import Engine, Tableau, default_engine
from .input_file_parser import Constraint


class PrimalSimplexSolver:
    #: snapshots are O(R*C) text each (three per pivot in the C#, :139,146,148); above this many
    #: tableau elements the default policy keeps only the initial and final ones.
    SNAPSHOT_ALL_LIMIT = 4096

    def __init__(self, objective: Sequence[float], constraints: Sequence[Constraint],
                 isMaximization: bool = True, *, engine: Optional[Engine] = None,
                 snapshots: str = "auto", verbose: bool = False):
        self._engine = engine or default_engine()
        self._verbose = verbose
        self.numVariables = n = len(objective)  # :29
        self.numConstraints = m = len(constraints)  # :53
        self.iteration = 0
        self.IterationSnapshots: List[str] = []
        self.FinalZ: float = 0.0  # C# auto-property default
        self.SolutionVector: Optional[List[float]] = None
        self.FinalTableau: Optional[np.ndarray] = None
        self.Status: Optional[int] = None

        # Flatten List<Constraint> for the ABI.  Relation strings map as in :36-50: ">=" negates,
        # "=" and anything else are kept as "<=".
        A = np.zeros((m, max(n, 1)), dtype=np.float64)
        ncoef = np.zeros(m, dtype=np.int32)
        rel = np.zeros(m, dtype=np.int8)
        rhs = np.zeros(m, dtype=np.float64)
        for i, c in enumerate(constraints):
            k = min(n, len(c.Coefficients))  # :68-72 `if (j < constraint.Coefficients.Count)`
            A[i, :k] = c.Coefficients[:k]
            ncoef[i] = k
            rel[i] = N.LPR_REL_GE if c.Relation == ">=" else (
                N.LPR_REL_EQ if c.Relation == "=" else N.LPR_REL_LE)
            rhs[i] = c.RHS
        self._tab = Tableau.from_lp(self._engine, list(objective), A[:, :n] if n else A[:, :0],
                                    rel, rhs, is_max=isMaximization, ncoef=ncoef)
        elems = self._tab.rows * self._tab.cols
        if snapshots == "auto":
            snapshots = "all" if elems <= self.SNAPSHOT_ALL_LIMIT else "none"
        self._snapshots = snapshots
        if snapshots != "none":
            self._capture("Initial Tableau")  # :86

    # -- helpers ---------------------------------------------------------------------------
    def _col_label(self, col: int) -> str:  # :253-254
        return f"x{col + 1}" if col < self.numVariables else f"t{col - self.numVariables + 1}"

    def _capture(self, title: str) -> None:  # :89-92
        self.IterationSnapshots.append(fmt.Format(self._tab.read(), self.numVariables, title))

    def _say(self, text: str) -> None:
        if self._verbose:
            print(text)

    def _solution_summary(self, title: str = "Optimal solution") -> str:  # :256-267
        nl = fmt.NEWLINE
        s = title + ":" + nl + f"Z = {fmt.F6(self.FinalZ)}" + nl
        if self.SolutionVector is not None:
            for i in range(self.numVariables):
                s += f"x{i + 1} = {fmt.F6(self.SolutionVector[i])}" + nl
        return s

    # -- Solve :102-150 --------------------------------------------------------------------
    def Solve(self, max_pivots: int = 0) -> None:
        if self._snapshots == "all":
            status = self._solve_stepwise(max_pivots)
        else:
            res = self._tab.solve(max_pivots=max_pivots)
            status = res.status
            self.iteration = int(res.total_pivots)
        self.Status = status
        if status == N.LPR_OK_OPTIMAL:  # :110-126
            x, z = self._tab.extract_solution(self.numVariables)
            self.FinalZ = z
            self.SolutionVector = [float(v) for v in x]
            self.FinalTableau = self._tab.read()
            self._say("Optimal Solution Found!")
            if self._snapshots != "none":
                block = fmt.Format(self.FinalTableau, self.numVariables,
                                   "Final Tableau (Optimal)") + fmt.NEWLINE
                block += self._solution_summary() + fmt.NEWLINE
                self.IterationSnapshots.append(block)
            self._say(self._solution_summary())
            self._say("-" * 100)
        elif status == N.LPR_UNBOUNDED:  # :129-135: prints, keeps FinalZ = 0, SolutionVector null
            self._say("Unbounded Solution!")
            self.FinalTableau = self._tab.read()
            if self._snapshots != "none":
                self.IterationSnapshots.append(
                    fmt.Format(self.FinalTableau, self.numVariables, "Unbounded Tableau"))
        # LPR_PIVOT_LIMIT has no C# counterpart (the reference loop is uncapped)

    def _solve_stepwise(self, max_pivots: int) -> int:
        """Snapshot policy "all": one pivot at a time so that every tableau can be formatted."""
        t = self._tab
        while True:
            e = t.select_entering()
            if e == -1:
                return N.LPR_OK_OPTIMAL
            r = t.select_leaving(e)
            if r == -1:
                return N.LPR_UNBOUNDED
            if max_pivots > 0 and self.iteration >= max_pivots:
                return N.LPR_PIVOT_LIMIT
            self.iteration += 1
            self._say(f"\nIteration {self.iteration}: pivot @ constraint {r}, column "
                      f"{self._col_label(e)}")
            if self._verbose:
                print(fmt.Format(t.read(), self.numVariables, "Before pivot"))
            t.pivot(r, e)
            if self._verbose:
                print(f"After pivot (constraint {r}, column {self._col_label(e)}):")
                print(fmt.Format(t.read(), self.numVariables, "After pivot"))
            self._capture(f"Iteration {self.iteration} - After pivot")  # :148

    # -- result accessors :18-24, 269-278 -----------------------------------------------------
    def GetFinalTableau(self) -> np.ndarray:
        return self._tab.read()

    @property
    def BasicVariables(self) -> List[int]:
        return [int(v) for v in self._tab.basis()]

    @property
    def FinalLabels(self) -> List[str]:
        return [self._col_label(i) for i in self.BasicVariables]

    @property
    def FinalTable(self) -> str:
        if self.FinalTableau is None:
            return ""
        return fmt.Format(self.FinalTableau, self.numVariables, "Final Table", self.FinalLabels)

    @property
    def PivotLog(self) -> np.ndarray:
        """(row, column) of every pivot, row 1-based as in the console line of :138."""
        return self._tab.pivot_log()

    @property
    def tableau(self) -> Tableau:
        return self._tab
