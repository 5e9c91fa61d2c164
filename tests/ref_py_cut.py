"""Independent pure-Python restatement (TEST ONLY) of the cutting-plane side path, written from the
C# text: Simplex/DualSimplex.cs, Simplex/PrimalSimplexSolver2.cs,
IntegerProgramming/CuttingPlaneSolver.cs.  objectiveRow is a list, constraintRows a list of lists,
as in the reference.  Must agree bit-for-bit with oracle/oracle_cut.c."""
from __future__ import annotations

import math

EPS = 1e-9
INF = math.inf


class PivotTooSmall(Exception):
    pass


def _pivot(obj, rows, pr, pc):  # DualSimplex.cs:150-178
    prow = rows[pr]
    piv = prow[pc]
    if abs(piv) <= EPS:
        raise PivotTooSmall()
    for j in range(len(obj)):
        prow[j] = prow[j] / piv
    for r in range(len(rows)):
        if r == pr:
            continue
        f = rows[r][pc]
        if abs(f) > EPS:
            for j in range(len(obj)):
                rows[r][j] = rows[r][j] - f * prow[j]
    of = obj[pc]
    if abs(of) > EPS:
        for j in range(len(obj)):
            obj[j] = obj[j] - of * prow[j]


def dual_solve(obj, rows, max_iters=10_000, print_steps=True, log=None, hard_cap=0):
    """DualSimplexSolver.Solve :14-114 -> 'ok' | 'infeasible' | 'limit'."""
    width = len(obj)
    it = 0
    done = 0
    while True:
        pivot_row = -1
        most_neg = 0.0
        for r in range(len(rows)):
            rhs = rows[r][width - 1]
            if rhs < most_neg - EPS or (abs(rhs - most_neg) <= EPS and pivot_row != -1 and r < pivot_row):
                most_neg = rhs
                pivot_row = r
        if pivot_row == -1:
            return "ok"
        pivot_col = -1
        best = INF
        for j in range(width - 1):
            a = rows[pivot_row][j]
            if a < -EPS:
                num = obj[j]
                if abs(num) > EPS:
                    ratio = abs(num / a)
                    if ratio < best - EPS or (abs(ratio - best) <= EPS and (pivot_col == -1 or j < pivot_col)):
                        best = ratio
                        pivot_col = j
        if pivot_col == -1:
            return "infeasible"
        if hard_cap and done >= hard_cap:
            return "limit"
        if print_steps:
            it += 1
        if log is not None:
            log.append((0, pivot_row, pivot_col))
        _pivot(obj, rows, pivot_row, pivot_col)
        done += 1
        if it >= max_iters:
            return "limit"


def primal2_solve(obj, rows, max_iters=10_000, print_steps=False, log=None, hard_cap=0):
    """PrimalSimplexSolver2.Solve :46-97 on a tableau t (row 0 = objective) -> 'ok' |
    'unbounded' | 'limit'.  Works on copies like the C# (its own double[,]) and writes back."""
    t = [list(obj)] + [list(r) for r in rows]
    R, C = len(t), len(obj)
    rhs = C - 1
    it = 0
    done = 0
    status = None
    while True:
        pc = -1
        most_neg = 0.0
        for j in range(rhs):
            c = t[0][j]
            if c < most_neg - EPS or (abs(c - most_neg) <= EPS and pc != -1 and j < pc):
                most_neg = c
                pc = j
        if pc == -1:
            status = "ok"
            break
        best_row = -1
        best = INF
        for i in range(1, R):
            a = t[i][pc]
            if a > EPS:
                ratio = t[i][rhs] / a
                second = True if (abs(ratio - best) <= EPS and best_row == -1) else (i < best_row)
                if (ratio > EPS and ratio < best - EPS) or second:
                    best = ratio
                    best_row = i
        if best_row == -1:
            status = "unbounded"
            break
        if hard_cap and done >= hard_cap:
            status = "limit"
            break
        if print_steps:
            it += 1
        if log is not None:
            log.append((1, best_row, pc))
        piv = t[best_row][pc]
        if abs(piv) <= EPS:
            raise PivotTooSmall()
        for j in range(C):
            t[best_row][j] = t[best_row][j] / piv
        for i in range(R):
            if i == best_row:
                continue
            f = t[i][pc]
            if abs(f) <= EPS:
                continue
            for j in range(C):
                t[i][j] = t[i][j] - f * t[best_row][j]
        done += 1
        if it >= max_iters:
            status = "limit"
            break
    obj[:] = t[0]
    for i in range(len(rows)):
        rows[i][:] = t[i + 1]
    return status


def _frac(a):  # CuttingPlaneSolver.cs:12-17
    f = a - math.floor(a)
    if abs(f) < EPS or abs(1 - f) < EPS:
        return 0.0
    return f


def cutting_plane(obj, rows, max_cuts=64, log=None, hard_cap=0):
    """CuttingPlaneSolver.CuttingPlaneSolution :64-229 (recursion unrolled).  Mutates obj / rows
    in place (rows grows by one list per cut).  Returns (exit code, cuts) with the exit codes of
    oracle_cut.c."""
    cuts = 0
    while True:
        fractional = []
        for i, row in enumerate(rows):
            fr = _frac(row[-1])
            if fr > EPS:
                fractional.append((i, row, row[-1], fr))
        if not fractional:
            return 1, cuts
        if cuts >= max_cuts:
            return 6, cuts
        fractional.sort(key=lambda t: abs(t[3] - 0.5))  # stable, like an insertion sort
        chosen = fractional[0]
        n = len(chosen[1])
        cut = [-_frac(chosen[1][j]) for j in range(n)]
        rows.append(cut)
        cuts += 1
        cut_idx = len(rows) - 1
        pc = -1
        best = INF
        for j in range(n - 1):
            a = cut[j]
            if a < -EPS:
                num = obj[j]
                if abs(num) > EPS:
                    ratio = abs(num / a)
                    if ratio < best - EPS or (abs(ratio - best) <= EPS and (pc == -1 or j < pc)):
                        best = ratio
                        pc = j
        if pc == -1:
            return 2, cuts
        if abs(rows[cut_idx][pc]) <= EPS:
            return 3, cuts
        if log is not None:
            log.append((2, cut_idx, pc))
        _pivot(obj, rows, cut_idx, pc)
        need_dual = any(r[-1] < -EPS for r in rows)
        need_primal = any(obj[j] < -EPS for j in range(n - 1))
        try:
            if need_dual:
                if dual_solve(obj, rows, print_steps=True, log=log, hard_cap=hard_cap) != "ok":
                    return 4, cuts
                need_primal = any(obj[j] < -EPS for j in range(n - 1))
            if need_primal:
                primal2_solve(obj, rows, print_steps=True, log=log, hard_cap=hard_cap)
        except PivotTooSmall:
            return 7, cuts
        if all(obj[j] >= -EPS for j in range(n - 1)) and not any(r[-1] < -EPS for r in rows):
            if any(_frac(r[-1]) > EPS for r in rows):
                continue
            return 0, cuts
        return 5, cuts
