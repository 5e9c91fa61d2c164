"""Independent pure-Python restatement of the Branch & Bound path (TEST ONLY), written from the C#
text of IntegerProgramming/BranchBoundSimplexSolver.cs and BranchAndBoundAdapter.cs -- not from
oracle/oracle_bb.c -- on List<List<double>>-style lists, exceptions included (Python exceptions
stand for the .NET ones the callers' try/catch swallow).  Must agree bit-for-bit with the C oracle.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

INF = math.inf


# ---- .NET Framework rounding -----------------------------------------------------------------
def to_int32(x: float) -> int:
    """`(int)d` (:870-871) under .NET Framework 4.7.2 x64 (cvttsd2si): 0x80000000 out of range."""
    return int(x) if -2147483649.0 < x < 2147483648.0 else -2147483648


def round_int(x: float) -> float:
    """Math.Round(double) (COMDouble::Round)."""
    if x != x or x in (INF, -INF):
        return x
    if abs(x) < 9.2e18 and x == float(int(x)):
        return x
    t = x + 0.5
    f = math.floor(t)
    if f == t and math.fmod(t, 2.0) != 0:
        f -= 1.0
    return math.copysign(f, x)


def round4(x: float) -> float:
    """Math.Round(double, 4) (Math.InternalRound, ToEven)."""
    if abs(x) < 1e16:
        x = x * 10000.0
        x = round_int(x)
        x = x / 10000.0
    return x


def _div(a: float, b: float) -> float:
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0.0:
            return math.nan
        neg = (math.copysign(1.0, a) < 0) != (math.copysign(1.0, b) < 0)
        return -INF if neg else INF


def _index_of(lst, v) -> int:
    """List<double>.IndexOf (Double.Equals: NaN == NaN, +0 == -0)."""
    for i, x in enumerate(lst):
        if x == v or (x != x and v != v):
            return i
    return -1


def _clean(tab):
    for row in tab:
        for i in range(len(row)):
            if row[i] == 0.0:
                row[i] = 0.0


class DualSimplexSolverBB:
    def __init__(self):
        self.pivotColumns: List[int] = []
        self.pivotRows: List[int] = []
        self.trace: List[Tuple[int, int, int]] = []  # (phase, row, col)

    def PerformDualPivot(self, tableau):  # :115-201
        rhs = [row[-1] for row in tableau]
        neg = [x for x in rhs if x < 0]
        if not neg:
            return tableau, None
        minRhs = min(neg)
        pr = _index_of(rhs, minRhs)
        thetas = []
        for i in range(len(tableau[pr]) - 1):
            if tableau[pr][i] < 0:
                thetas.append(abs(_div(tableau[0][i], tableau[pr][i])))
            else:
                thetas.append(INF)
        if all(x == 0 or x == INF for x in thetas):
            minPos = 0
        else:
            pos = [x for x in thetas if x > 0]
            minPos = min(pos) if pos else INF
        pc = _index_of(thetas, minPos)
        if pc < 0:
            return tableau, None  # ArgumentOutOfRangeException caught (:165-172)
        p = tableau[pr][pc]
        new = [[0.0] * len(r) for r in tableau]
        for j in range(len(tableau[pr])):
            v = _div(tableau[pr][j], p)
            if v == 0.0:
                v = 0.0
            new[pr][j] = v
        for i in range(len(tableau)):
            if i == pr:
                continue
            for j in range(len(tableau[i])):
                new[i][j] = tableau[i][j] - (tableau[i][pc] * new[pr][j])
        self.pivotColumns.append(pc)
        self.pivotRows.append(pr)
        self.trace.append((0, pr, pc))
        return new, thetas

    def PerformPrimalPivot(self, tableau):  # :203-279, isMinimization false
        objrow = tableau[0][:-1]
        cands = [x for x in objrow if x < 0 and x != 0]
        if not cands:
            return None, None
        pv = min(cands)
        pc = _index_of(tableau[0], pv)
        thetas = []
        for i in range(1, len(tableau)):
            a = tableau[i][pc]
            thetas.append(_div(tableau[i][-1], a) if a != 0 else INF)
        if all(t < 0 for t in thetas):
            return None, None
        if not any(t > 0 and t != INF for t in thetas):
            if _index_of(thetas, 0) >= 0:
                minTheta = 0.0
            else:
                return None, None
        else:
            minTheta = min(t for t in thetas if t > 0 and t != INF)
        if minTheta == INF and _index_of(thetas, 0) < 0:
            return None, None
        pr = _index_of(thetas, minTheta) + 1
        p = tableau[pr][pc]
        if p == 0:
            return None, None
        new = [[0.0] * len(r) for r in tableau]
        for j in range(len(tableau[pr])):
            v = _div(tableau[pr][j], p)
            if v == 0.0:
                v = 0.0
            new[pr][j] = v
        for i in range(len(tableau)):
            if i == pr:
                continue
            for j in range(len(tableau[i])):
                new[i][j] = tableau[i][j] - (tableau[i][pc] * new[pr][j])
        self.pivotColumns.append(pc)
        self.pivotRows.append(pr)
        self.trace.append((1, pr, pc))
        return new, thetas

    def DoDualSimplex(self, tableauOverride):  # :289-468, override mode
        tableaux = [tableauOverride]
        self.pivotColumns = []
        self.pivotRows = []
        self.trace = []
        while True:
            _clean(tableaux[-1])
            if all(row[-1] >= -1e-9 for row in tableaux[-1]):
                break
            new, th = self.PerformDualPivot(tableaux[-1])
            if th is None:
                return tableaux, None
            _clean(new)
            tableaux.append(new)
        if not all(v >= 0 for v in tableaux[-1][0][:-1]):
            while True:
                _clean(tableaux[-1])
                if all(v >= 0 for v in tableaux[-1][0][:-1]):
                    break
                new, th = self.PerformPrimalPivot(tableaux[-1])
                if th is None:
                    break  # thetaCol.ToList() -> NullReferenceException -> catch -> break
                tableaux.append(new)
            if not all(row[-1] >= 0 for row in tableaux[-1]):
                tableaux.pop()
                self.pivotColumns.pop()  # IndexError == ArgumentOutOfRangeException
                self.pivotRows.pop()
                self.trace.append((2, -1, -1))
        optimalValue = tableaux[-1][0][-1]  # IndexError if tableaux is empty
        return tableaux, optimalValue


class BranchAndBound:
    def __init__(self, nvars: int, node_cap: int = 20):
        self.nvars = nvars
        self.epsilon = 1e-6
        self.solver = DualSimplexSolverBB()
        self.node_cap = node_cap
        self.records = []   # dicts: parent, kind, depth, var, bound, status, z
        self.pop_order = []
        self.trace = []     # (node id, phase, row, col)

    def IsInteger(self, v):
        r = round4(v)
        return abs(r - round_int(r)) <= self.epsilon

    def RoundTableau(self, t):
        return [[round4(v) for v in row] for row in t]

    def IdentifyBasicVariables(self, t):  # :642-692
        ncols = len(t[-1])
        basic = []
        for k in range(ncols):
            vals = [round4(t[i][k]) for i in range(len(t))]
            s = 0.0
            for v in vals:
                s += v
            s = round4(s)
            if abs(s - 1.0) <= self.epsilon:
                basic.append(k)
        cols = [[round4(t[j][i]) for j in range(len(t))] for i in range(ncols) if i in basic]
        zipped = list(zip(cols, basic))
        zipped.sort(key=lambda cs: _index_of(cs[0], 1.0) if _index_of(cs[0], 1.0) >= 0
                    else len(cs[0]))  # list.sort is stable like OrderBy
        return [s for _, s in zipped]

    def AddConstraint(self, con, base):  # :694-803, one constraint
        working = self.RoundTableau([list(r) for r in base])
        basic = self.IdentifyBasicVariables(working)
        updated = [list(r) for r in working]
        for i in range(len(working)):
            updated[i].insert(len(updated[i]) - 1, 0.0)
        new = [0.0] * (len(working[0]) + 1)
        for i in range(len(con) - 2):
            new[i] = round4(con[i])
        new[-1] = round4(con[-2])
        slack = ((len(new) - 1) - 1) + 0
        new[slack] = -1.0 if con[-1] == 1 else 1.0
        updated.append(new)
        updated = self.RoundTableau(updated)
        out = [list(r) for r in updated]
        crow = len(updated) - 1
        for colIndex in basic:
            coefficient = round4(out[crow][colIndex])
            if abs(coefficient) > self.epsilon:
                pivotRow = None
                for rowIndex in range(len(out) - 1):
                    if abs(round4(out[rowIndex][colIndex]) - 1.0) <= self.epsilon:
                        pivotRow = rowIndex
                        break
                if pivotRow is not None:
                    reverse = int(con[-1]) == 1
                    for col in range(len(out[0])):
                        pv = round4(out[pivotRow][col])
                        cv = round4(out[crow][col])
                        if reverse:
                            nv = pv - coefficient * cv
                        else:
                            nv = cv - coefficient * pv
                        out[crow][col] = round4(nv)
        return self.RoundTableau(out)

    def _decision(self, t):  # :807-827 / :899-921
        vals = []
        for i in range(self.nvars):
            found = False
            for j in range(len(t)):
                v = round4(t[j][i])
                if abs(v - 1.0) <= self.epsilon:
                    vals.append(round4(t[j][-1]))
                    found = True
                    break
            if not found:
                vals.append(0.0)
        return vals

    def Execute(self, initial, enable_pruning=False):  # :1006-1233
        root = self.RoundTableau(initial)
        optimalSolution: Optional[List[float]] = None
        optimalValue = -INF
        optimalNode = -1
        branchCount = 0
        self.records = [dict(parent=-1, kind=0, depth=0, var=-1, bound=0.0, status=0,
                             z=round4(root[0][-1]))]
        stack = [(root, 0, 0)]
        iteration = 0
        capped = False
        while stack:
            iteration += 1
            if iteration > self.node_cap:
                capped = True
                break
            tab, depth, nid = stack.pop()
            self.pop_order.append(nid)
            branchCount += 1
            tab = self.RoundTableau(tab)
            objVal = round4(tab[0][-1])
            if enable_pruning and optimalSolution is not None and objVal <= optimalValue:
                continue
            vals = self._decision(tab)
            if all(self.IsInteger(v) for v in vals) and objVal > optimalValue:
                optimalValue = objVal
                optimalSolution = vals
                optimalNode = nid
            best = -1
            bestValue = None
            minDist = INF
            for i, v in enumerate(vals):
                if not self.IsInteger(v):
                    d = abs((v - math.floor(v)) - 0.5)
                    if d < minDist:
                        minDist = d
                        best = i
                        bestValue = v
            if best == -1:
                continue
            upperInt = to_int32(math.ceil(bestValue))
            lowerInt = to_int32(math.floor(bestValue))
            kids = []
            for side, (bound, typ) in enumerate(((lowerInt, 0), (upperInt, 1))):
                con = [1.0 if i == best else 0.0 for i in range(self.nvars)] + [float(bound),
                                                                                float(typ)]
                rid = len(self.records)
                try:
                    adj = self.AddConstraint(con, tab)
                    tabs, opt = self.solver.DoDualSimplex(adj)
                    for ph, r, c in self.solver.trace:
                        self.trace.append((rid, ph, r, c))
                    if opt is None:
                        self.records.append(dict(parent=nid, kind=side + 1, depth=depth + 1,
                                                 var=best, bound=float(bound), status=1, z=0.0))
                        continue
                    last = self.RoundTableau(tabs[-1])
                    self.records.append(dict(parent=nid, kind=side + 1, depth=depth + 1,
                                             var=best, bound=float(bound), status=0,
                                             z=round4(last[0][-1])))
                    kids.append((last, depth + 1, rid))
                except IndexError:
                    for ph, r, c in self.solver.trace:
                        self.trace.append((rid, ph, r, c))
                    self.records.append(dict(parent=nid, kind=side + 1, depth=depth + 1,
                                             var=best, bound=float(bound), status=2, z=0.0))
            for k in reversed(kids):
                stack.append(k)
        return dict(x=optimalSolution, z=optimalValue, best_node=optimalNode,
                    processed=branchCount, capped=capped)
