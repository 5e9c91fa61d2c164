"""Host-side mirror of ``SensitivityAnalyzer``
(SensitivityAnalysis/SensitivityAnalyzer.cs:8-730).

Same public members as the C# class.  The C# methods prompt on the console; here the answers to
the prompts are arguments (raw, i.e. 1-based where the prompt says "1-based") and the text the C#
would have written with ``Console.WriteLine`` is appended to ``self.Out`` (and echoed when
``verbose``).  The tableau lives in HBM (lpr_sens_*, csrc/sens_engine.hip): every edit, the dual /
primal re-solve after it, RebuildBasicsFromTableau and the O(m*n) column sums run on the device;
the host only does the O(rows + cols) arithmetic of the range displays on the rows / columns it
fetches.  There is no CPU fallback.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np

from . import _native as N
from . import table_iteration_formater as fmt
from .engine import Engine, SensState, default_engine

EPS = 1e-9  # :20


class InvalidOperationException(Exception):
    """System.InvalidOperationException thrown by Pivot / ReOptimize / DualSimplexIfNeeded."""


_MESSAGES = {
    N.LPR_SENS_ZERO_PIVOT: "Zero pivot encountered.",                          # :101
    N.LPR_SENS_ITER_LIMIT: "Re-optimization exceeded iteration limit.",        # :126 / :183
    N.LPR_SENS_UNBOUNDED: "Unbounded during re-optimization.",                 # :151
    N.LPR_SENS_INFEASIBLE: "Infeasible after RHS change (dual simplex).",      # :197
}


def F(v: float) -> str:
    """v.ToString("0.###", CultureInfo.InvariantCulture) (:53)."""
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "Infinity" if v > 0 else "-Infinity"
    return fmt._custom_0_hashes(v)


class SensitivityAnalyzer:
    def __init__(self, finalTableau, solution: Sequence[float], zValue: float,
                 basicVariables: Optional[Sequence[int]] = None, *,
                 engine: Optional[Engine] = None, verbose: bool = False,
                 _state: Optional[SensState] = None):
        # basicVariables is accepted for signature parity; the C# overwrites it at once with
        # RebuildBasicsFromTableau (:35), and so does the device.
        self._verbose = verbose
        self.Out: List[str] = []
        self.LastOutcome: int = N.LPR_SENS_OK
        if _state is not None:
            self._s = _state
        else:
            self._s = SensState.create(engine or default_engine(),
                                       np.asarray(finalTableau, dtype=np.float64),
                                       list(solution), float(zValue))
        self._validate_binary()  # :38

    @classmethod
    def FromPrimalSolver(cls, primalSolver, *, verbose: bool = False) -> "SensitivityAnalyzer":
        """Program.cs:147-151 without the round trip through the host: the solved tableau is
        copied device to device."""
        st = SensState.from_tableau(primalSolver._tab, primalSolver.numVariables)
        return cls(None, (), 0.0, None, verbose=verbose, _state=st)

    # -- plumbing ----------------------------------------------------------------------------
    def _say(self, text: str) -> None:
        self.Out.append(text)
        if self._verbose:
            print(text)

    @property
    def numRows(self) -> int:
        return self._s.shape()[0]

    @property
    def numCols(self) -> int:
        return self._s.shape()[1]

    @property
    def basicVars(self) -> List[int]:
        return [int(b) for b in self._s.read(tableau=False)[1]]

    @property
    def CurrentTableau(self) -> np.ndarray:  # :727
        return self._s.read()[0]

    @property
    def CurrentZ(self) -> float:  # :728
        return self._s.shape()[4]

    @property
    def CurrentSolutionVector(self) -> List[float]:  # :729
        return [float(v) for v in self._s.read(tableau=False)[2]]

    def PivotLog(self):
        """(kind 0 dual / 1 primal, leaveRow, enterCol) of every pivot so far (not in the C#)."""
        return self._s.log()

    def _row(self, i: int) -> np.ndarray:
        return self._s.read_block(i, 1, 0, self.numCols)[0]

    def _col(self, j: int) -> np.ndarray:
        return self._s.read_block(0, self.numRows, j, 1)[:, 0]

    def _validate_binary(self) -> None:  # :43-51
        sol = self.CurrentSolutionVector
        for i in range(min(len(sol), 6)):
            if abs(sol[i] - fmt.dotnet_round_half_even(sol[i])) > EPS:
                self._say(f"Warning: x{i + 1} = {F(sol[i])} violates binary constraint.")

    def ColLabel(self, col: int) -> str:  # :55-60
        m = self.numRows - 1
        n = self.numCols - m - 1
        return f"x{col + 1}" if col < n else f"s{col - n + 1}"

    def SlackColForConstraint(self, i: int) -> int:  # :62-67
        m = self.numRows - 1
        n = self.numCols - m - 1
        return n + (i - 1)

    def GetBasicRow(self, col: int) -> int:  # :69-77
        return self._s.basic_row(col)

    def ShadowPrices(self) -> np.ndarray:  # :212-222
        m = self.numRows - 1
        n = self.numCols - m - 1
        return self._s.read_block(0, 1, n, m)[0] if m > 0 else np.zeros(0)

    def _finish(self, outcome: int, title: str) -> int:
        """What follows ResolveAll() in every edit: the exception or the PrintTableau."""
        self.LastOutcome = outcome
        if outcome in _MESSAGES:
            raise InvalidOperationException(_MESSAGES[outcome])
        if outcome == N.LPR_SENS_INDEX_OUT_OF_RANGE:
            raise IndexError("Index was outside the bounds of the array.")
        self.PrintTableau(title)
        return outcome

    # -- display -----------------------------------------------------------------------------
    def PrintTableau(self, title: Optional[str] = None, max_elements: int = 1 << 16) -> None:
        """:250-272.  Tableaux above ``max_elements`` entries are summarised (the C# would print
        them in full; at BASELINE sizes that is hundreds of megabytes of text)."""
        if title is not None and title.strip():
            self._say(f"\n=== {title} ===")
        rows, cols, nsol, _, z, _ = self._s.shape()
        m = rows - 1
        n = cols - m - 1
        if rows * cols <= max_elements:
            T = self.CurrentTableau
            self._say("\t".join([self.ColLabel(j) for j in range(n + m)] + ["RHS/Z"]))
            for i in range(rows):
                self._say("\t".join(F(v) for v in T[i]))
        else:
            self._say(f"[{rows} x {cols} tableau not printed]")
        self._say(f"Current Solution: Z = {F(z)}")
        sol = self.CurrentSolutionVector
        if rows * cols <= max_elements:
            for j in range(cols - 1):
                if j >= len(sol):  # solutionVector[j] throws in the C#
                    raise IndexError("Index was out of range. Must be non-negative and less than "
                                     "the size of the collection.")
                self._say(f"{self.ColLabel(j)} = {F(sol[j])}")

    def DisplayRangeNonBasic(self, indexRaw: int) -> None:  # :276-297
        index = indexRaw - 1
        if index < 0 or index >= self.numCols - 1 or index in self.basicVars:
            self._say("Invalid index or variable is basic.")
            return
        cbar = float(self._s.read_block(0, 1, index, 1)[0, 0])
        self._say(f"Reduced Cost for {self.ColLabel(index)}: {F(cbar)}")
        if cbar > EPS:
            self._say(f"Range for c{index + 1}: can DECREASE by at most {F(cbar)}, INCREASE "
                      "without bound.")
        elif abs(cbar) <= EPS:
            self._say(f"Range for c{index + 1}: at boundary (c̄=0). Any decrease makes "
                      f"{self.ColLabel(index)} enter; any increase is fine.")
        else:
            self._say("Warning: tableau not optimal (negative reduced cost found). Consider "
                      "re-optimizing.")

    def BasicRange(self, index: int):
        """The interval of DisplayRangeBasic (:340-355) for 0-based column ``index`` with basic
        row r; (deltaLower, deltaUpper)."""
        r = self.GetBasicRow(index)
        if r == -1:
            return None
        row_r = self._row(r)
        row_0 = self._row(0)
        basic = set(self.basicVars)
        lo, hi = -math.inf, math.inf
        for j in range(self.numCols - 1):
            if j == index or j in basic:
                continue
            a, cbar = float(row_r[j]), float(row_0[j])
            if a > EPS:
                lo = max(lo, -cbar / a)
            elif a < -EPS:
                hi = min(hi, -cbar / a)
        return lo, hi

    def DisplayRangeBasic(self, indexRaw: int) -> None:  # :324-359
        index = indexRaw - 1
        if index < 0 or index >= self.numCols - 1 or index not in self.basicVars:
            self._say("Invalid index or variable is non-basic.")
            return
        rng = self.BasicRange(index)
        if rng is None:
            self._say("Error: Basic variable not found in tableau.")
            return
        lo, hi = rng
        self._say(f"Allowable change Δ for {self.ColLabel(index)}’s objective coeff that keeps "
                  f"basis optimal: [{F(lo)}, {F(hi)}]")
        self._say("Interpretation: set c_B(new) = c_B(old) + Δ within this interval to preserve "
                  "basis.")

    def RHSRange(self, k: int):
        """(shadow price, deltaLower, deltaUpper, current RHS) of DisplayRangeRHS (:402-420)."""
        sCol = self.SlackColForConstraint(k)
        col = self._col(sCol)
        rhs = self._col(self.numCols - 1)
        lo, hi = -math.inf, math.inf
        for i in range(1, self.numRows):
            coeff, bi = float(col[i]), float(rhs[i])
            if coeff > EPS:
                lo = max(lo, -bi / coeff)
            elif coeff < -EPS:
                hi = min(hi, -bi / coeff)
        return float(col[0]), lo, hi, float(rhs[k])

    def DisplayRangeRHS(self, k: int) -> None:  # :396-424
        if k < 1 or k >= self.numRows:
            self._say("Invalid constraint index.")
            return
        y, lo, hi, cur = self.RHSRange(k)
        self._say(f"Shadow Price y_{k} = {F(y)}")
        self._say(f"Allowable RHS change Δ for constraint {k}: [{F(lo)}, {F(hi)}]")
        self._say(f"So b_{k} may vary within [{F(cur + lo)}, {F(cur + hi)}] without changing the "
                  "basis.")

    def DisplayRangeNonBasicColumn(self, row: int, colRaw: int) -> None:  # :473-499
        if row < 1 or row >= self.numRows:
            self._say("Invalid row.")
            return
        col = colRaw - 1
        if col < 0 or col >= self.numCols - 1 or col in self.basicVars:
            self._say("Invalid column, or column is basic.")
            return
        aij = float(self._s.read_block(row, 1, col, 1)[0, 0])
        cbar = float(self._s.read_block(0, 1, col, 1)[0, 0])
        yi = float(self.ShadowPrices()[row - 1])
        lo, hi = -math.inf, math.inf
        if yi > EPS:
            lo = max(lo, -cbar / yi)
        elif yi < -EPS:
            hi = min(hi, -cbar / yi)
        lab = self.ColLabel(col)
        self._say(f"Allowable Δ for a[{row},{lab}] keeping basis optimal: [{F(lo)}, {F(hi)}]")
        self._say(f"So a[{row},{lab}] may vary within [{F(aij + lo)}, {F(aij + hi)}].")

    def DisplayShadowPrices(self) -> None:  # :662-668
        self._say("Shadow Prices y (Z−C on slack columns):")
        for i, v in enumerate(self.ShadowPrices()):
            self._say(f"  Constraint {i + 1}: y_{i + 1} = {F(float(v))}")

    def RecoverObjectiveC(self) -> np.ndarray:  # :226-246; the column sums run on the device
        m = self.numRows - 1
        n = self.numCols - m - 1
        y = self.ShadowPrices()
        aty = self._s.column_fold(y, None, n)
        row0 = self._s.read_block(0, 1, 0, n)[0] if n > 0 else np.zeros(0)
        return aty - row0

    def PerformDuality(self) -> None:  # :671-702
        y = self.ShadowPrices()
        chat = self.RecoverObjectiveC()
        zstar = float(self._s.read_block(0, 1, self.numCols - 1, 1)[0, 0])
        self._say("Dual (derived from final tableau; tableau stores Z−C):")
        self._say("  For max with ≤-type rows: minimize b^T y, s.t. A^T y ≥ c, y ≥ 0.")
        self._say(f"  y* = [{', '.join(F(float(v)) for v in y)}]")
        self._say(f"  ĉ (consistent with tableau) = [{', '.join(F(float(v)) for v in chat)}]")
        self._say(f"  Z* (from tableau) = {F(zstar)}")
        self._say("  Note: b here equals B^{-1}b (tableau RHS), so we do not compare b^T y to Z* "
                  "numerically.")

    # -- edits -------------------------------------------------------------------------------
    def ResolveAll(self) -> int:  # :203-208 (private in the C#)
        oc = self._s.resolve_all()
        self.LastOutcome = oc
        if oc in _MESSAGES:
            raise InvalidOperationException(_MESSAGES[oc])
        return oc

    def ChangeNonBasicReducedCost(self, idxRaw: int, newCbar: float) -> int:  # :300-321
        index = idxRaw - 1
        if index < 0 or index >= self.numCols - 1:
            self._say("Invalid index or variable is basic.")
            self.LastOutcome = N.LPR_SENS_INVALID_INDEX
            return self.LastOutcome
        label = self.ColLabel(index)
        oc = self._s.change_nonbasic_cbar(index, newCbar)
        if oc == N.LPR_SENS_INVALID_INDEX:
            self._say("Invalid index or variable is basic.")
            self.LastOutcome = oc
            return oc
        self._say(f"Updated reduced cost for {label}: c̄ = {F(newCbar)}")
        return self._finish(oc, "After nonbasic c̄ change (resolved)")

    def ChangeBasic(self, colRaw: int, delta: float) -> int:  # :362-393
        col = colRaw - 1
        if col < 0 or col >= self.numCols - 1 or col not in self.basicVars:
            self._say("Invalid index or variable is non-basic.")
            self.LastOutcome = N.LPR_SENS_INVALID_INDEX
            return self.LastOutcome
        label = self.ColLabel(col)
        oc = self._s.change_basic(col, delta)
        if oc == N.LPR_SENS_INVALID_INDEX:
            self._say("Error: Basic variable not found in tableau.")
            self.LastOutcome = oc
            return oc
        self._say(f"Applied Δ = {F(delta)} to c_B for {label}. Objective row updated; Z adjusted "
                  "by Δ*x_B.")
        return self._finish(oc, "After c_B change (resolved)")

    def ChangeRHS(self, k: int, newB: float) -> int:  # :427-470
        oc = self._s.change_rhs(k, newB)
        self.LastOutcome = oc
        if oc == N.LPR_SENS_INVALID_INDEX:
            self._say("Invalid constraint index.")
        elif oc == N.LPR_SENS_ROLLED_BACK:
            self._say("This RHS change makes the model infeasible for the current basis. "
                      "Use option 5 (RHS range) to see the allowable interval.")
        else:
            self.PrintTableau("After RHS change (resolved)")
        return oc

    def ChangeNonBasicColumn(self, row: int, colRaw: int, newVal: float) -> int:  # :502-531
        if row < 1 or row >= self.numRows:
            self._say("Invalid row.")
            self.LastOutcome = N.LPR_SENS_INVALID_INDEX
            return self.LastOutcome
        oc = self._s.change_nonbasic_column(row, colRaw - 1, newVal)
        if oc == N.LPR_SENS_INVALID_INDEX:
            self._say("Invalid column, or column is basic.")
            self.LastOutcome = oc
            return oc
        return self._finish(oc, "After a_ij change (resolved)")

    def AddNewActivity(self, cNew: float, aNew: Sequence[float]) -> int:  # :534-584
        m = self.numRows - 1
        n = self.numCols - m - 1
        a = [float(v) for v in aNew]
        if len(a) != m:
            raise ValueError(f"aNew needs one coefficient per constraint row ({m})")
        y = self.ShadowPrices()
        yTa = 0.0
        for i in range(m):  # only for the message; the device recomputes it in the same order
            yTa = yTa + float(y[i]) * a[i]
        cbarNew = yTa - cNew
        oc = self._s.add_activity(cNew, a)
        self._say(f"Added new variable x{n + 1}: c = {F(cNew)}, y^T a = {F(yTa)}, c̄ = "
                  f"{F(cbarNew)}. Resolving...")
        return self._finish(oc, "After adding variable (resolved)")

    def AddNewConstraint(self, tech: Sequence[float], rhs: float) -> int:  # :587-607
        return self.AddNewConstraintNonInteractive(tech, rhs)

    def AddNewConstraintNonInteractive(self, tech: Sequence[float], rhs: float) -> int:  # :609-659
        t = [float(v) for v in tech]
        if len(t) != self.numCols - 1:
            raise IndexError("Index was outside the bounds of the array.")  # tech[j], :638
        oc = self._s.add_constraint(t, rhs)
        if oc == N.LPR_SENS_INDEX_OUT_OF_RANGE:
            return self._finish(oc, "")
        self._say(f"Added new constraint (row {self.numRows - 1}). Resolving...")
        return self._finish(oc, "After adding constraint (resolved)")
