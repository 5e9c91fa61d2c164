"""Second, independent restatement of the reference's solvers in pure Python (TEST ONLY).

Written directly from the C# text (never from oracle/*.c) so that two readings of the reference can
be compared bit-for-bit: pivot logs, bases and result bits of this module must equal those of the C
oracle (tests/test_oracle_*.py).  Python floats are IEEE binary64 and ``a - f * b`` rounds the
product before the difference, which is the C# semantics (no FMA).

Citations are relative to LPR_381_Group_V22/ in the reference tree.
"""
from __future__ import annotations

import math
import sys
from typing import List, Optional, Sequence, Tuple

DBL_MAX = sys.float_info.max


# --------------------------------------------------------------------------------------------
# Simplex/PrimalSimplexSolver.cs
# --------------------------------------------------------------------------------------------
class PyConstraint:
    def __init__(self, coefficients: Sequence[float], relation: str, rhs: float):
        self.Coefficients = list(coefficients)
        self.Relation = relation
        self.RHS = rhs


class PyPrimal:
    def __init__(self, objective: Sequence[float], constraints: Sequence[PyConstraint],
                 is_max: bool = True):
        n = len(objective)
        processed = []
        for c in constraints:  # :34-51
            if c.Relation == ">=":
                processed.append(PyConstraint([-v for v in c.Coefficients], "<=", -c.RHS))
            else:
                processed.append(PyConstraint(list(c.Coefficients), "<=", c.RHS))
        m = len(processed)
        cols = n + m + 1
        rows = m + 1
        t = [[0.0] * cols for _ in range(rows)]
        for i in range(n):  # :61-62
            t[0][i] = -objective[i] if is_max else objective[i]
        self.basic: List[int] = []
        for i in range(m):  # :65-83
            c = processed[i]
            for j in range(n):
                if j < len(c.Coefficients):
                    t[i + 1][j] = c.Coefficients[j]
            t[i + 1][n + i] = 1.0
            self.basic.append(n + i)
            t[i + 1][cols - 1] = c.RHS
        self.t = t
        self.n = n
        self.m = m
        self.log: List[Tuple[int, int]] = []
        self.status: Optional[str] = None
        self.FinalZ = 0.0
        self.SolutionVector: Optional[List[float]] = None

    def find_entering(self) -> int:  # :152-167
        col = -1
        most_negative = 0.0
        row0 = self.t[0]
        for j in range(len(row0) - 1):
            if row0[j] < most_negative:
                most_negative = row0[j]
                col = j
        return col

    def find_leaving(self, e: int) -> int:  # :169-191
        leaving = -1
        min_ratio = DBL_MAX
        rhs = len(self.t[0]) - 1
        for i in range(1, len(self.t)):
            a = self.t[i][e]
            if a > 1e-9:
                ratio = _div(self.t[i][rhs], a)
                if ratio >= 0 and ratio < min_ratio:
                    min_ratio = ratio
                    leaving = i
        return leaving

    def pivot(self, r: int, e: int) -> None:  # :193-211
        t = self.t
        p = t[r][e]
        prow = t[r]
        for j in range(len(prow)):
            prow[j] = _div(prow[j], p)
        for i in range(len(t)):
            if i != r:
                row = t[i]
                f = row[e]
                for j in range(len(row)):
                    row[j] = row[j] - f * prow[j]

    def solve(self, max_pivots: int = 0) -> str:  # :102-150
        it = 0
        while True:
            e = self.find_entering()
            if e == -1:
                self.status = "optimal"
                self.FinalZ = self.t[0][-1]
                self.SolutionVector = self.extract_solution()
                break
            r = self.find_leaving(e)
            if r == -1:
                self.status = "unbounded"
                break
            if max_pivots > 0 and it >= max_pivots:
                self.status = "limit"
                break
            it += 1
            self.log.append((r, e))
            self.pivot(r, e)
            self.basic[r - 1] = e
        return self.status

    def extract_solution(self) -> List[float]:  # :213-252
        sol = [0.0] * self.n
        rows = len(self.t)
        rhs = len(self.t[0]) - 1
        for j in range(self.n):
            basic_row = -1
            is_basic = True
            for i in range(1, rows):
                v = self.t[i][j]
                if abs(v - 1.0) < 1e-9:
                    if basic_row == -1:
                        basic_row = i
                    else:
                        is_basic = False
                        break
                elif abs(v) > 1e-9:
                    is_basic = False
                    break
            if is_basic and basic_row != -1:
                sol[j] = self.t[basic_row][rhs]
        return sol


def _div(a: float, b: float) -> float:
    """IEEE division with C#'s results for x/0 (Python raises instead)."""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0.0:
            return math.nan
        neg = (math.copysign(1.0, a) < 0) != (math.copysign(1.0, b) < 0)
        return -math.inf if neg else math.inf


# --------------------------------------------------------------------------------------------
# Program.cs glue that shapes the solver inputs
# --------------------------------------------------------------------------------------------
def program_option1_constraints(n: int, constraints: List[PyConstraint]) -> List[PyConstraint]:
    """Program.cs:114-124 (and :372-382): append n rows "x_i <= 1" whose coefficient list has
    n + 3 entries ([i] = 1 and [n + 1] = 1); the solver only reads the first n."""
    out = list(constraints)
    vec_len = n + 3
    for i in range(n):
        coeffs = [0.0] * vec_len
        coeffs[i] = 1.0
        coeffs[vec_len - 2] = 1.0
        out.append(PyConstraint(coeffs, "<=", 1.0))
    return out


def program_option2_constraints(n: int, sign_restrictions: List[str],
                                constraints: List[PyConstraint]) -> List[PyConstraint]:
    """Program.cs:511-535 AddUpperBoundConstraints."""
    out = [PyConstraint(list(c.Coefficients), c.Relation, c.RHS) for c in constraints]
    if not sign_restrictions:
        return out
    for j in range(n):
        sr = sign_restrictions[min(j, len(sign_restrictions) - 1)] or ""
        s = sr.replace(" ", "")
        is_bin = "bin" in s.lower()
        has_upper = ("≤1" in s) or ("<=1" in s)
        if is_bin or has_upper:
            coeffs = [0.0] * n
            coeffs[j] = 1.0
            out.append(PyConstraint(coeffs, "<=", 1.0))
    return out


def parse_model_text(text: str):
    """IO/InputFileParser.cs:27-65 on the text of a model file."""
    lines = text.splitlines()
    assert len(lines) >= 3
    head = lines[0].strip().split(" ")
    ptype = head[0].lower()
    obj = [float(tok) for tok in head[1:]]
    cons = []
    for ln in lines[1:-1]:
        parts = [p for p in ln.strip().split(" ") if p]
        coeffs = [float(parts[j]) for j in range(len(obj))]
        cons.append(PyConstraint(coeffs, parts[len(obj)], float(parts[len(obj) + 1])))
    signs = lines[-1].strip().split(" ")
    return ptype, obj, cons, signs
