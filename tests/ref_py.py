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
# Simplex/RevisedPrimalSimplexSolver.cs
# --------------------------------------------------------------------------------------------
REV_EPS = 1e-9


def _mat_vec(M, v):  # :398-410
    out = []
    for row in M:
        s = 0.0
        for j in range(len(row)):
            s += row[j] * v[j]
        out.append(s)
    return out


def _vec_mat(v, M):  # :412-424
    rows = len(M)
    cols = len(M[0])
    out = []
    for j in range(cols):
        s = 0.0
        for i in range(rows):
            s += v[i] * M[i][j]
        out.append(s)
    return out


def _dot(a, b):  # :443-448
    s = 0.0
    for i in range(len(a)):
        s += a[i] * b[i]
    return s


def _mat_mul_skip(A, B):  # :426-441
    rA, cA, cB = len(A), len(A[0]), len(B[0])
    R = [[0.0] * cB for _ in range(rA)]
    for i in range(rA):
        for k in range(cA):
            aik = A[i][k]
            if abs(aik) < REV_EPS:
                continue
            Bk = B[k]
            Ri = R[i]
            for j in range(cB):
                Ri[j] += aik * Bk[j]
    return R


class PyRevised:
    """ctor :41-80, Solve :82-251, UpdateBInverse :264-275, ExtractSolution :277-287."""

    def __init__(self, objective, constraints, is_min):
        if not objective or not constraints:
            raise ValueError("empty")
        self.n = n = len(objective)
        self.m = m = len(constraints)
        self.is_min = is_min
        self.cOrig = list(objective)
        self.c = [-v for v in objective] if is_min else list(objective)
        self.A = []
        self.b = []
        for i, con in enumerate(constraints):
            if len(con.Coefficients) != n:
                raise ValueError(f"Constraint {i + 1} has incorrect number of coefficients.")
            self.A.append(list(con.Coefficients))
            self.b.append(con.RHS)
        self.basic = [n + i for i in range(m)]
        self.nonbasic = list(range(n))
        self.Binv = [[1.0 if i == j else 0.0 for j in range(m)] for i in range(m)]
        self.cB = [0.0] * m
        self.xB = [0.0] * m
        self.log = []
        self.status = None
        self.FinalZ = 0.0
        self.SolutionVector = []
        self.snapshots = []  # CaptureSnapshot (:294-387) as dicts of numbers, when capture=True

    def _z_original(self):  # ComputeOriginalZFromCurrentBasis :253-262
        x = [0.0] * self.n
        for i in range(self.m):
            v = self.basic[i]
            if v < self.n:
                x[v] = 0.0 if 0.0 > self.xB[i] else self.xB[i]
        return _dot(self.cOrig, x)

    def _capture(self, title, y, rcX, rcS, entering, rc_pre, u_pre, ratios_pre, basis_pre,
                 leaving_row, leaving_var, z_original):
        self.snapshots.append(dict(
            title=title, y=list(y), rcX=list(rcX), rcS=list(rcS), entering=entering,
            rc_pre=rc_pre, u_pre=list(u_pre), ratios_pre=list(ratios_pre),
            basis_pre=list(basis_pre), leaving_row=leaving_row, leaving_var=leaving_var,
            z_working=_dot(self.cB, self.xB), z_original=z_original, xB=list(self.xB),
            basis_post=list(self.basic), BInvA=_mat_mul_skip(self.Binv, self.A),
            BInv=[list(r) for r in self.Binv]))

    def solve(self, max_iter=0, capture=False):
        n, m = self.n, self.m
        it = 0
        while True:
            self.xB = _mat_vec(self.Binv, self.b)
            if any(v < -REV_EPS for v in self.xB):
                self.status = "infeasible_basis"
                return self.status
            y = _vec_mat(self.cB, self.Binv)
            rcX = [self.c[j] - _dot(y, [self.A[i][j] for i in range(m)]) for j in range(n)]
            rcS = [-y[k] for k in range(m)]
            entering = -1
            best = -math.inf
            for v in sorted(self.nonbasic):
                rc = rcX[v] if v < n else rcS[v - n]
                if rc > REV_EPS:
                    if entering == -1 or rc > best + REV_EPS or \
                            (abs(rc - best) <= REV_EPS and v < entering):
                        best = rc
                        entering = v
            if entering == -1:
                x = [0.0] * n
                for i in range(m):
                    v = self.basic[i]
                    if v < n:
                        x[v] = 0.0 if 0.0 > self.xB[i] else self.xB[i]  # Math.Max(0.0, xB)
                self.SolutionVector = x
                self.FinalZ = _dot(self.cOrig, x)
                self.status = "optimal"
                if capture:  # :127-143
                    self._capture("Optimal", y, rcX, rcS, -1, 0.0, [0.0] * m, [math.inf] * m,
                                  self.basic, -1, -1, self.FinalZ)
                return self.status
            if max_iter > 0 and it >= max_iter:
                self.status = "limit"
                return self.status
            if entering < n:
                u = _mat_vec(self.Binv, [self.A[i][entering] for i in range(m)])
            else:
                u = [self.Binv[i][entering - n] for i in range(m)]
            leaving = -1
            best_ratio = DBL_MAX
            ratios = [math.inf] * m
            for i in range(m):
                if u[i] > REV_EPS:
                    ratio = self.xB[i] / u[i]
                    ratios[i] = ratio
                    if ratio < best_ratio - REV_EPS or \
                            (abs(ratio - best_ratio) <= REV_EPS and
                             (leaving == -1 or self.basic[i] < self.basic[leaving])):
                        best_ratio = ratio
                        leaving = i
            if leaving == -1:
                self.status = "unbounded"
                return self.status
            leaving_var = self.basic[leaving]
            if leaving_var == entering:
                self.status = "entering_already_basic"
                return self.status
            self.log.append((leaving, entering, leaving_var))
            basis_pre = list(self.basic)
            rc_pre = rcX[entering] if entering < n else rcS[entering - n]
            self.basic[leaving] = entering
            self.nonbasic.remove(entering)
            if leaving_var not in self.nonbasic:
                self.nonbasic.append(leaving_var)
            self.cB[leaving] = self.c[entering] if entering < n else 0.0
            pivot = u[leaving]
            if abs(pivot) < REV_EPS:
                self.status = "pivot_too_small"
                return self.status
            E = [[1.0 if i == j else 0.0 for j in range(m)] for i in range(m)]
            for i in range(m):
                E[i][leaving] = (1.0 / pivot) if i == leaving else (-u[i] / pivot)
            self.Binv = _mat_mul_skip(E, self.Binv)
            if capture:  # :217-247
                self.xB = _mat_vec(self.Binv, self.b)
                y2 = _vec_mat(self.cB, self.Binv)
                rcX2 = [self.c[j] - _dot(y2, [self.A[i][j] for i in range(m)]) for j in range(n)]
                rcS2 = [-y2[k] for k in range(m)]
                self._capture(f"Iteration {it + 1}", y2, rcX2, rcS2, entering, rc_pre, u, ratios,
                              basis_pre, leaving, leaving_var, self._z_original())
            it += 1


# --------------------------------------------------------------------------------------------
# CaptureSnapshot's text (:294-387) and NumFormat.N3 (:451-466), restated for the tests
# --------------------------------------------------------------------------------------------
def py_round3_away(x: float) -> float:
    """Math.Round(x, 3, MidpointRounding.AwayFromZero) on .NET Framework (Math.InternalRound):
    scale by 10^3, split off the fraction, bump when |fraction| >= 0.5, unscale."""
    if math.isnan(x) or math.isinf(x) or abs(x) >= 1e16:
        return x
    v = x * 1000.0
    frac, whole = math.modf(v)
    if abs(frac) >= 0.5:
        whole += math.copysign(1.0, frac)
    return whole / 1000.0


def py_round_even(x: float) -> float:
    """Math.Round(double): to the nearest integer, ties to even."""
    if math.isnan(x) or math.isinf(x):
        return x
    f = math.floor(x)
    d = x - f
    if d > 0.5 or (d == 0.5 and f % 2.0 != 0.0):
        f += 1.0
    return math.copysign(f, x) if f == 0.0 else f


def py_n3(x: float) -> str:
    from decimal import ROUND_HALF_UP, Decimal
    if abs(x) < 1e-12:
        x = 0.0
    r = py_round3_away(x)
    ri = py_round_even(r)
    if abs(r - ri) < 1e-12:
        # double.ToString() of an integral value: 15 significant digits, no exponent below 1e15
        if ri == 0:
            return "0"
        if abs(ri) < 1e15:
            return "%d" % int(ri)
        # at 1e15 the general format switches to scientific: 15 significant digits, then E+xx
        from decimal import Context, ROUND_HALF_EVEN
        tup = Context(prec=15, rounding=ROUND_HALF_EVEN).create_decimal(Decimal(abs(ri))).as_tuple()
        digits = "".join(str(k) for k in tup.digits)
        exp10 = len(digits) - 1 + tup.exponent
        digits = digits.rstrip("0") or "0"
        mant = digits[0] + ("." + digits[1:] if len(digits) > 1 else "")
        return ("-" if ri < 0 else "") + mant + ("E+%02d" % exp10)
    # ToString("0.###"): the 15-significant-digit decimal image, rounded half away at 3 decimals
    d = Decimal("%.15g" % r).quantize(Decimal("0.001"), rounding=ROUND_HALF_UP)
    t = format(d, "f").rstrip("0").rstrip(".")
    return "0" if t in ("-0", "") else t


def py_var_label(idx: int, n: int) -> str:  # :289-292
    return f"x{idx + 1}" if idx < n else f"S{idx - n + 1}"


def py_snapshot_text(snap: dict, n: int, m: int, is_min: bool) -> str:
    nl = "\r\n"
    t = "\t"
    out = [snap["title"], nl, "Current Tableau (Revised Simplex)", nl,
           "Problem type: " + ("MIN (solving by MAX of -c)" if is_min else "MAX"), nl, nl,
           "Dual prices (y = c_B^T B^{-1}):", nl, t.join(py_n3(v) for v in snap["y"]), nl, nl,
           "Reduced costs:", nl, "  x: ", t.join(py_n3(v) for v in snap["rcX"]), nl,
           "  s: ", t.join(py_n3(v) for v in snap["rcS"]), nl, nl]
    e = snap["entering"]
    if e >= 0:
        el = py_var_label(e, n)
        out += [f"Entering variable (chosen pre-pivot): {el}  (reduced cost pre = "
                f"{py_n3(snap['rc_pre'])})", nl,
                "Direction u = B^{-1} a_enter (pre-pivot):", nl,
                t.join(py_n3(v) for v in snap["u_pre"]), nl, nl,
                "Ratio test (xB_i / u_i; \u221e if u_i \u2264 0)  [labels = pre-pivot basis]:", nl]
        for i in range(m):
            rv = snap["ratios_pre"][i]
            out += [py_var_label(snap["basis_pre"][i], n), ": ",
                    "\u221e" if rv == math.inf else py_n3(rv), nl]
        if snap["leaving_row"] >= 0 and snap["leaving_var"] >= 0:
            out += [f"Pivot (pre\u2192post): {py_var_label(snap['leaving_var'], n)}  \u2192  {el}"
                    f"    (pivot = {py_n3(snap['u_pre'][snap['leaving_row']])})", nl, nl]
    out += [f"Working objective Z_working (maxified): {py_n3(snap['z_working'])}", nl,
            f"Original objective Z_original ({'MIN' if is_min else 'MAX'}): "
            f"{py_n3(snap['z_original'])}", nl, nl]
    out += ["Table", t] + [f"x{j + 1}{t}" for j in range(n)] + [f"S{j + 1}{t}" for j in range(m)] \
        + ["RHS", nl]
    out += ["Z~", t] + [py_n3(v) + t for v in snap["rcX"]] + [py_n3(v) + t for v in snap["rcS"]] \
        + [py_n3(snap["z_working"]), nl]
    for i in range(m):
        out += [py_var_label(snap["basis_post"][i], n), t]
        out += [py_n3(v) + t for v in snap["BInvA"][i]]
        out += [py_n3(v) + t for v in snap["BInv"][i]]
        out += [py_n3(snap["xB"][i]), nl]
    out += ["Basic Variables: ", ", ".join(py_var_label(v, n) for v in snap["basis_post"]), nl]
    return "".join(out)


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
