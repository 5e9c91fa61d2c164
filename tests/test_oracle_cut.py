"""CPU tests: the C oracle of the cutting-plane side path (DualSimplex.cs, PrimalSimplexSolver2.cs,
CuttingPlaneSolver.cs) against the independent Python restatement.  PARITY UNPINNED by the
reference (this code is not even reachable from its menu, Program.cs:417-428)."""
import numpy as np
import pytest

import cut_cases
import ref_py_cut as rp

DUAL = {0: "ok", 1: "infeasible", 5: "limit"}
PRIM = {0: "ok", 1: "unbounded", 5: "limit"}


def split(T):
    return list(map(float, T[0])), [list(map(float, r)) for r in T[1:]]


def join(obj, rows):
    return np.array([obj] + rows, dtype=np.float64)


def test_dual_simplex(oracle):
    cases = cut_cases.dual_tableaux(oracle)
    assert len(cases) >= 8
    seen = set()
    for name, T0 in cases:
        T = T0.copy()
        rc, piv, log = oracle.dual_solve(T, print_steps=True, hard_cap=2000)
        obj, rows = split(T0)
        plog = []
        st = rp.dual_solve(obj, rows, print_steps=True, log=plog, hard_cap=2000)
        assert st == DUAL[rc], name
        assert plog == log, name
        assert join(obj, rows).tobytes() == T.tobytes(), name
        seen.add(st)
    assert "ok" in seen


def test_dual_simplex_max_iters_is_inert_without_print_steps(oracle):
    """DualSimplex.cs:94,108: `iter` only advances inside `if (printSteps)`."""
    name, T0 = cut_cases.dual_tableaux(oracle)[0]
    a, b = T0.copy(), T0.copy()
    rc_a, piv_a, _ = oracle.dual_solve(a, max_iters=1, print_steps=False, hard_cap=2000)
    rc_b, piv_b, _ = oracle.dual_solve(b, max_iters=1, print_steps=True, hard_cap=2000)
    assert piv_b == 1 and rc_b == 5
    assert rc_a in (0, 1) and piv_a >= piv_b


def test_primal_simplex_solver2(oracle):
    seen = set()
    for name, T0 in cut_cases.primal2_tableaux(oracle):
        T = T0.copy()
        rc, piv, log = oracle.primal2_solve(T, print_steps=False, hard_cap=3000)
        obj, rows = split(T0)
        plog = []
        st = rp.primal2_solve(obj, rows, print_steps=False, log=plog, hard_cap=3000)
        assert st == PRIM[rc], name
        assert plog == log, name
        assert join(obj, rows).tobytes() == T.tobytes(), name
        seen.add(st)
    assert {"ok", "unbounded"} <= seen


def test_primal2_objective_matches_primal_solver_on_nondegenerate_lp(oracle):
    """Different tie rules, same optimum: on a non-degenerate LP both primal solvers must reach the
    same objective value (sanity of the restatement, 1e-9 relative)."""
    import lp_cases
    obj, cons, _ = lp_cases.random_dense(16, 32, 1)
    o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
    T, basis = oracle.primal_build(o, A, rel, rhs, True, ncoef)
    T2 = T.copy()
    oracle.primal_solve(T, basis)
    rc, piv, log = oracle.primal2_solve(T2)
    assert rc == 0 and abs(T[0, -1] - T2[0, -1]) <= 1e-9 * abs(T[0, -1])


@pytest.mark.parametrize("max_cuts", [1, 6])
def test_cutting_plane(oracle, max_cuts):
    exits = set()
    for name, T0 in cut_cases.cutting_plane_tableaux(oracle):
        rc, cuts, T, log = oracle.cutting_plane(T0, max_cuts=max_cuts, hard_cap=2000)
        obj, rows = split(T0)
        plog = []
        prc, pcuts = rp.cutting_plane(obj, rows, max_cuts=max_cuts, log=plog, hard_cap=2000)
        assert (prc, pcuts) == (rc, cuts), name
        assert plog == log, name
        assert join(obj, rows).tobytes() == T.tobytes(), name
        exits.add(rc)
    assert len(exits) >= 2, exits
