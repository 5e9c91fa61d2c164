"""Independent pure-Python restatement (TEST ONLY) of the re-solve half of
SensitivityAnalysis/SensitivityAnalyzer.cs, written from the C# text.  Exceptions of the C# are
Python exceptions here; ``code_of`` maps them to the return codes of oracle/oracle_sens.c."""
from __future__ import annotations

import math

EPS = 1e-9


class Unbounded(Exception):
    code = 1


class Infeasible(Exception):
    code = 2


class ZeroPivot(Exception):
    code = 3


class IterLimit(Exception):
    code = 5


class PySens:
    def __init__(self, final_tableau, solution, z, basic):  # :22-39
        self.t = [list(map(float, r)) for r in final_tableau]
        self.sol = list(solution)
        self.z = z
        self.basic = list(basic)
        self.t[0][-1] = z
        self.log = []
        self._rebuild()

    @property
    def R(self):
        return len(self.t)

    @property
    def C(self):
        return len(self.t[0])

    def _is_pivot_col(self, prow, col):  # :79-84
        return all(i == prow or abs(self.t[i][col]) <= EPS for i in range(1, self.R))

    def _basic_row(self, col):  # :69-77
        for i in range(1, self.R):
            if abs(self.t[i][col] - 1.0) < EPS and self._is_pivot_col(i, col):
                return i
        return -1

    def _rebuild(self):  # :706-723
        m = self.R - 1
        self.basic = [-1] * m
        for i in range(1, m + 1):
            for j in range(self.C - 1):
                if abs(self.t[i][j] - 1.0) < EPS and self._is_pivot_col(i, j):
                    self.basic[i - 1] = j
                    break

    def _optimal(self):  # :86-96
        bs = set(self.basic)
        return all(j in bs or self.t[0][j] >= -EPS for j in range(self.C - 1))

    def _pivot(self, enter, leave, kind):  # :98-119
        t = self.t
        piv = t[leave][enter]
        if abs(piv) < EPS:
            raise ZeroPivot()
        self.log.append((kind, leave, enter))
        for j in range(self.C):
            t[leave][j] = t[leave][j] / piv
        for i in range(self.R):
            if i == leave:
                continue
            f = t[i][enter]
            if abs(f) < EPS:
                continue
            for j in range(self.C):
                t[i][j] = t[i][j] - f * t[leave][j]
        if 0 <= leave - 1 < len(self.basic):
            self.basic[leave - 1] = enter

    def _reoptimize(self, max_iter=10000):  # :121-166
        it = 0
        while not self._optimal():
            if it > max_iter:
                raise IterLimit()
            it += 1
            enter, most = -1, 0.0
            for j in range(self.C - 1):
                if j in self.basic:
                    continue
                if self.t[0][j] < most:
                    most, enter = self.t[0][j], j
            if enter == -1:
                break
            leave, best = -1, math.inf
            for i in range(1, self.R):
                a = self.t[i][enter]
                if a > EPS:
                    ratio = self.t[i][-1] / a
                    if ratio < best - EPS:
                        best, leave = ratio, i
            if leave == -1:
                raise Unbounded()
            self._pivot(enter, leave, 1)
        self.z = self.t[0][-1]
        self.sol = []
        for j in range(self.C - 1):
            r = self._basic_row(j)
            self.sol.append(0.0 if r == -1 else self.t[r][-1])

    def _dual(self, max_iter=10000):  # :168-201
        it = 0
        while True:
            leave, most = -1, 0.0
            for i in range(1, self.R):
                b = self.t[i][-1]
                if b < most - EPS:
                    most, leave = b, i
            if leave == -1:
                break
            if it > max_iter:
                raise IterLimit()
            it += 1
            enter, best = -1, math.inf
            for j in range(self.C - 1):
                a = self.t[leave][j]
                if a < -EPS:
                    ratio = self.t[0][j] / (-a)
                    if ratio < best - EPS:
                        best, enter = ratio, j
            if enter == -1:
                raise Infeasible()
            self._pivot(enter, leave, 0)

    def resolve_all(self):  # :203-208
        self._rebuild()
        self._dual()
        self._reoptimize()

    def _y(self, k):  # shadow price of constraint k (1-based), :212-222
        m = self.R - 1
        n = self.C - m - 1
        return self.t[0][n + (k - 1)]

    # ---- edits ----
    def change_nonbasic_cbar(self, index, new):  # :300-321
        if index < 0 or index >= self.C - 1 or index in self.basic:
            return -1
        self.t[0][index] = new
        self.resolve_all()
        return 0

    def change_basic(self, col, delta):  # :362-393
        if col < 0 or col >= self.C - 1 or col not in self.basic:
            return -1
        r = self._basic_row(col)
        if r < 0:
            return -1
        for j in range(self.C - 1):
            self.t[0][j] = self.t[0][j] + delta * self.t[r][j]
        self.t[0][-1] = self.t[0][-1] + delta * self.t[r][-1]
        self.z = self.t[0][-1]
        self.resolve_all()
        return 0

    def change_rhs(self, k, new_b):  # :427-470
        if k < 1 or k >= self.R:
            return -1
        snap = [list(r) for r in self.t]
        bsnap = list(self.basic)
        old_z = self.z
        delta = new_b - self.t[k][-1]
        m = self.R - 1
        n = self.C - m - 1
        s_col = n + (k - 1)
        for i in range(1, self.R):
            self.t[i][-1] = self.t[i][-1] + delta * self.t[i][s_col]
        self.t[0][-1] = self.t[0][-1] + self._y(k) * delta
        self.z = self.t[0][-1]
        try:
            self._dual()
            self._reoptimize()
            return 0
        except (Unbounded, Infeasible, ZeroPivot, IterLimit):
            self.t = snap
            self.z = old_z
            self.basic = bsnap
            return 8

    def change_nonbasic_column(self, row, col, new):  # :502-531
        if row < 1 or row >= self.R or col < 0 or col >= self.C - 1 or col in self.basic:
            return -1
        delta = new - self.t[row][col]
        self.t[row][col] = new
        self.t[0][col] = self.t[0][col] + self._y(row) * delta
        self.resolve_all()
        return 0

    def add_activity(self, c_new, a_new):  # :534-584
        m = self.R - 1
        n = self.C - m - 1
        yta = 0.0
        for i in range(m):
            yta = yta + self._y(i + 1) * a_new[i]
        cbar = yta - c_new
        nt = []
        for i in range(self.R):
            row = self.t[i]
            nt.append(row[:n] + [cbar if i == 0 else a_new[i - 1]] + row[n:self.C - 1] + [row[-1]])
        self.t = nt
        self.basic = [b + 1 if b >= n else b for b in self.basic]
        self.resolve_all()
        return 0

    def add_constraint(self, tech, rhs):  # :609-659
        old_m = self.R - 1
        old_nm = self.C - 1
        if len(tech) != old_nm:
            return -1
        new_slack = self.C - 1
        nt = [r[:-1] + [0.0] + [r[-1]] for r in self.t]
        new_row = [0.0] * (self.C + 1)
        for j in range(old_nm):
            coeff = -tech[j]
            for pos in range(old_m):
                bc = self.basic[pos]
                if bc < 0:
                    return 9  # tech[-1]: IndexOutOfRangeException in the C#
                coeff = coeff + tech[bc] * self.t[pos + 1][j]
            new_row[j] = coeff
        ax = 0.0
        for j in range(min(len(tech), len(self.sol))):
            ax = ax + tech[j] * self.sol[j]
        new_row[new_slack] = 1.0
        new_row[-1] = rhs - ax
        nt.append(new_row)
        nt[0][new_slack] = 0.0
        self.t = nt
        self.basic.append(new_slack)
        self.resolve_all()
        return 0


def run(fn, *args):
    """Call an edit; exceptions become the oracle's return codes."""
    try:
        return fn(*args)
    except (Unbounded, Infeasible, ZeroPivot, IterLimit) as ex:
        return ex.code
