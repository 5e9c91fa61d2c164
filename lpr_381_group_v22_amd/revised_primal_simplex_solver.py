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


def VarLabel(idx: int, n: int) -> str:  # :289-292
    return f"x{idx + 1}" if idx < n else f"S{idx - n + 1}"


class RevisedPrimalSimplexSolver:
    #: one snapshot prints the m x (n + m + 1) table B^-1 A | B^-1 | RHS (:359-384); above this
    #: many entries ``snapshots="auto"`` keeps none (the C# would produce megabytes of text per pivot)
    SNAPSHOT_ALL_LIMIT = 4096

    def __init__(self, objective: Sequence[float], constraints: Sequence[Constraint],
                 isMinimization: bool, *, engine: Optional[Engine] = None,
                 snapshots: str = "auto"):
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
        if snapshots == "auto":
            snapshots = "all" if m * (n + m + 1) <= self.SNAPSHOT_ALL_LIMIT else "none"
        if snapshots not in ("all", "none"):
            raise ValueError("snapshots must be 'auto', 'all' or 'none'")
        self._snapshots = snapshots
        self.IterationSnapshots: List[str] = []
        self.FinalZ: float = 0.0
        self.SolutionVector: List[float] = []
        self.Status: Optional[int] = None
        self._iteration = 0  # the C#'s local `iteration` (:84), kept across resumed Solve() calls

    def Solve(self, max_pivots: int = 0) -> None:  # :82-251
        if self._snapshots == "all":
            return self._solve_with_snapshots(max_pivots)
        res = self._st.solve(max_pivots=max_pivots)
        self._finish(res.status)

    def _finish(self, status: int) -> None:
        self.Status = status
        if status == N.LPR_OK_OPTIMAL:  # :124-146
            x, z = self._st.solution()
            self.SolutionVector = [float(v) for v in x]
            self.FinalZ = z
        elif status in MESSAGES:
            raise SolverException(status)
        # LPR_PIVOT_LIMIT has no C# counterpart (`while (true)`, :86)

    def _solve_with_snapshots(self, max_pivots: int) -> None:
        """The same loop one iteration at a time (lpr_revised_step); after every pivot, and at the
        optimum, the text block of CaptureSnapshot (:294-387) is appended."""
        done = 0
        while True:
            if max_pivots > 0 and done >= max_pivots:
                self.Status = N.LPR_PIVOT_LIMIT
                return
            info = self._st.step()
            if info.status == N.LPR_PIVOT_LIMIT:      # a pivot: "Iteration k" (:232-247)
                done += 1
                self._iteration += 1
                self._capture(f"Iteration {self._iteration}", info)
                continue
            if info.status == N.LPR_OK_OPTIMAL:       # :124-146
                self._finish(info.status)
                self._capture("Optimal", info)
                return
            self._finish(info.status)                  # raises with the C# message
            return

    def _capture(self, title: str, info) -> None:  # CaptureSnapshot :294-387
        from .table_iteration_formater import N3, NEWLINE as nl
        n, m = self.numVariables, self.numConstraints
        y, rc, u_pre, ratios_pre, basis_pre, xB = self._st.snapshot()
        basis = self._st.basis()
        entering = info.entering
        if entering < 0:  # the "Optimal" call passes fresh zeros / infinities (:133-134)
            u_pre = np.zeros(m)
            ratios_pre = np.full(m, np.inf)
            basis_pre = basis
        rcX, rcS = rc[:n], rc[n:]
        kind = "MIN (solving by MAX of -c)" if self.isMinimization else "MAX"
        sb = [title + nl, "Current Tableau (Revised Simplex)" + nl, f"Problem type: {kind}" + nl, nl]
        sb += ["Dual prices (y = c_B^T B^{-1}):" + nl, "\t".join(N3(v) for v in y) + nl, nl]
        sb += ["Reduced costs:" + nl, "  x: ", "\t".join(N3(v) for v in rcX) + nl,
               "  s: ", "\t".join(N3(v) for v in rcS) + nl, nl]
        if entering >= 0:
            elabel = VarLabel(entering, n)
            sb.append(f"Entering variable (chosen pre-pivot): {elabel}  (reduced cost pre = "
                      f"{N3(info.entering_rc_pre)})" + nl)
            sb.append("Direction u = B^{-1} a_enter (pre-pivot):" + nl)
            sb.append("\t".join(N3(v) for v in u_pre) + nl)
            sb.append(nl)
            sb.append("Ratio test (xB_i / u_i; \u221e if u_i \u2264 0)  [labels = pre-pivot basis]:" + nl)
            for i in range(m):
                rstr = "\u221e" if ratios_pre[i] == np.inf else N3(ratios_pre[i])
                sb.append(f"{VarLabel(int(basis_pre[i]), n)}: {rstr}" + nl)
            if info.leaving_row >= 0 and info.leaving_var >= 0:
                sb.append(f"Pivot (pre\u2192post): {VarLabel(info.leaving_var, n)}  \u2192  {elabel}"
                          f"    (pivot = {N3(u_pre[info.leaving_row])})" + nl)
                sb.append(nl)
        sb.append(f"Working objective Z_working (maxified): {N3(info.z_working)}" + nl)
        sb.append(f"Original objective Z_original ({'MIN' if self.isMinimization else 'MAX'}): "
                  f"{N3(info.z_original)}" + nl)
        sb.append(nl)
        BInvA = self._st.binv_a_exact()   # MultiplyMatrices(BInverse, A) :360, the C#'s own order
        BInv = self._st.binv()
        sb.append("Table\t" + "".join(f"x{j + 1}\t" for j in range(n))
                  + "".join(f"S{j + 1}\t" for j in range(m)) + "RHS" + nl)
        sb.append("Z~\t" + "".join(N3(v) + "\t" for v in rcX) + "".join(N3(v) + "\t" for v in rcS)
                  + N3(info.z_working) + nl)
        for i in range(m):
            sb.append(VarLabel(int(basis[i]), n) + "\t"
                      + "".join(N3(v) + "\t" for v in BInvA[i])
                      + "".join(N3(v) + "\t" for v in BInv[i]) + N3(xB[i]) + nl)
        sb.append("Basic Variables: " + ", ".join(VarLabel(int(v), n) for v in basis) + nl)
        self.IterationSnapshots.append("".join(sb))

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
