"""Edit scripts for the sensitivity re-solve tests (TEST ONLY): a solved LP + a list of
(operation, args) applied in order to the same analyzer, as a user of the reference's sub-menu
(Program.cs:158-294) would."""
from __future__ import annotations

import numpy as np

import lp_cases


def solved_lp(oracle, m, n, seed, integer=False):
    if integer:
        obj, cons, _ = lp_cases.tie_heavy(m, n, seed)
        cons = [type(c)(c.Coefficients, "<=", abs(c.RHS) + 1.0) for c in cons]
    else:
        obj, cons, _ = lp_cases.random_dense(m, n, seed)
    o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
    T, basis = oracle.primal_build(o, A, rel, rhs, True, ncoef)
    st, piv, log = oracle.primal_solve(T, basis)
    assert st == 0
    x, z = oracle.extract_solution(T, n)
    return T, x, z, basis


def scripts(oracle):
    """[(name, (T, x, z, basis), [(op, args), ...])]"""
    out = []
    for (m, n, seed, integer) in [(4, 6, 0, False), (8, 12, 1, False), (6, 6, 2, True),
                                  (12, 9, 3, True), (16, 24, 4, False)]:
        base = solved_lp(oracle, m, n, seed, integer)
        T, x, z, basis = base
        rng = np.random.RandomState(100 + seed)
        C = T.shape[1]
        bset = set(int(b) for b in basis)
        nonbasic = [j for j in range(C - 1) if j not in bset]
        basic = [j for j in range(n) if j in bset] or [int(basis[0])]
        ops = [
            ("resolve_all", ()),
            ("change_nonbasic_cbar", (nonbasic[0], -2.5)),          # makes a reduced cost negative
            ("change_basic", (basic[0], 0.75)),
            ("change_rhs", (1, float(T[1, -1]) + 3.0)),
            ("change_rhs", (min(2, m), -5.0)),                       # may be rolled back
            ("change_nonbasic_column", (1, nonbasic[-1], 0.5)),
            ("add_activity", (float(rng.uniform(5, 9)), rng.uniform(0.1, 1.0, size=m).tolist())),
            ("add_constraint", (None, 1.0)),                         # tech filled in by the test
            ("change_nonbasic_cbar", (10 ** 6, 1.0)),                # invalid index
            ("change_rhs", (1, 0.0)),
            ("add_constraint_infeasible", ()),                       # code 2, state left mid-way
            ("add_activity_unbounded", ()),                          # code 1
        ]
        out.append((f"lp_{m}x{n}_s{seed}", base, ops))
    return out


def make_tech(width: int, seed: int):
    rng = np.random.RandomState(seed)
    t = np.zeros(width)
    k = max(1, width // 3)
    t[:k] = rng.randint(1, 4, size=k)
    return t.tolist()


def materialize(op, args, T, k):
    """Fill in the arguments that depend on the analyzer's current shape."""
    R, C = T.shape
    if op == "add_constraint":
        return op, (make_tech(C - 1, 7 + k), args[1])
    if op == "add_constraint_infeasible":       # sum of all columns <= -1 with x >= 0
        return "add_constraint", ([1.0] * (C - 1), -1.0)
    if op == "add_activity_unbounded":          # profitable activity that uses nothing
        return "add_activity", (50.0, [-1.0] * (R - 1))
    return op, args
