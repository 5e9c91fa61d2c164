"""Instances for the cutting-plane side path (TEST ONLY): tableaux (row 0 = objective row)."""
from __future__ import annotations

import numpy as np

import bb_cases
import lp_cases


def primal2_tableaux(oracle):
    """Initial tableaux of LPs with b >= 0 (PrimalSimplexSolver2 assumes a feasible basis)."""
    out = []
    for (m, n, seed) in [(4, 8, 0), (16, 32, 1), (40, 17, 4)]:
        obj, cons, _ = lp_cases.random_dense(m, n, seed)
        o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
        T, _ = oracle.primal_build(o, A, rel, rhs, True, ncoef)
        out.append((f"dense_{m}x{n}_s{seed}", T))
    for (m, n, seed) in [(6, 6, 0), (12, 9, 1), (24, 30, 2)]:
        obj, cons, _ = lp_cases.tie_heavy(m, n, seed)
        cons = [type(c)(c.Coefficients, "<=", abs(c.RHS)) for c in cons]
        o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
        T, _ = oracle.primal_build(o, A, rel, rhs, True, ncoef)
        out.append((f"ties_{m}x{n}_s{seed}", T))
    obj, cons, _ = lp_cases.unbounded_lp()
    o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
    T, _ = oracle.primal_build(o, A, rel, rhs, True, ncoef)
    out.append(("unbounded", T))
    return out


def dual_tableaux(oracle):
    """Dual-feasible tableaux with negative right-hand sides: an optimal tableau plus a violated
    branching row (what AddConstraint hands to a dual simplex), and >= rows negated by the ctor."""
    out = []
    for name, (obj, cons) in bb_cases.all_bb_cases()[:8]:
        st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
        vals = [0.0] * n
        for k in range(n):
            for j in range(T.shape[0]):
                if abs(T[j, k] - 1.0) <= 1e-6:
                    vals[k] = T[j, -1]
                    break
        frac = [k for k in range(n) if abs(vals[k] - round(vals[k])) > 1e-6]
        if not frac:
            continue
        k = frac[0]
        for side, bound in ((0, np.floor(vals[k])), (1, np.ceil(vals[k]))):
            con = np.zeros(n + 2)
            con[k] = 1.0
            con[n] = bound
            con[n + 1] = float(side)
            out.append((f"{name}_side{side}", oracle.bb_add_constraint(T, con)))
    # >= rows: the PrimalSimplexSolver ctor negates them, leaving negative RHS with a dual
    # feasible Z row when the objective is a minimisation-style (all reduced costs >= 0)
    rng = np.random.RandomState(5)
    for (m, n) in [(5, 7), (12, 9)]:
        A = rng.randint(1, 9, size=(m, n)).astype(float)
        b = rng.randint(5, 40, size=m).astype(float)
        c = rng.randint(1, 9, size=n).astype(float)
        T = np.zeros((m + 1, n + m + 1))
        T[0, :n] = c
        T[1:, :n] = -A
        T[1:, n:n + m] = np.eye(m)
        T[1:, -1] = -b
        out.append((f"cover_{m}x{n}", T))
    return out


def cutting_plane_tableaux(oracle):
    """Optimal LP tableaux with fractional right-hand sides (what CuttingPlaneSolution expects)."""
    out = []
    for name, (obj, cons) in bb_cases.all_bb_cases():
        st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
        out.append((name, T))
    return out
