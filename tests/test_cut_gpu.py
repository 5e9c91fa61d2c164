"""GPU parity tests of the cutting-plane side path (lpr_dual_solve, lpr_primal2_solve,
lpr_cutting_plane) against the C oracle: status / exit code, pivot log, tableau bits."""
import numpy as np
import pytest

import cut_cases

pytestmark = pytest.mark.gpu

DUAL_STATUS = {0: 0, 1: 2, 3: 3, 5: 5}    # oracle rc -> lpr_status (false = INFEASIBLE_BASIS)
PRIM_STATUS = {0: 0, 1: 1, 3: 3, 5: 5}    # false = UNBOUNDED


def test_dual_simplex_matches_oracle(engine, oracle):
    from lpr_381_group_v22_amd import Tableau
    for name, T0 in cut_cases.dual_tableaux(oracle):
        T = T0.copy()
        rc, piv, log = oracle.dual_solve(T, print_steps=True, hard_cap=2000)
        tab = Tableau.from_array(engine, T0)
        res = tab.dual_solve(print_steps=True, hard_cap=2000)
        assert res.status == DUAL_STATUS[rc], name
        assert res.pivots == piv, name
        assert tab.cut_log() == log, name
        assert tab.read().tobytes() == T.tobytes(), name
        tab.destroy()


def test_dual_max_iters_only_counts_with_print_steps(engine, oracle):
    from lpr_381_group_v22_amd import Tableau
    name, T0 = cut_cases.dual_tableaux(oracle)[0]
    for ps in (False, True):
        T = T0.copy()
        rc, piv, log = oracle.dual_solve(T, max_iters=1, print_steps=ps, hard_cap=2000)
        tab = Tableau.from_array(engine, T0)
        res = tab.dual_solve(max_iters=1, print_steps=ps, hard_cap=2000)
        assert res.status == DUAL_STATUS[rc] and res.pivots == piv, (name, ps)
        assert tab.read().tobytes() == T.tobytes()
        tab.destroy()


def test_primal_simplex_solver2_matches_oracle(engine, oracle):
    from lpr_381_group_v22_amd import Tableau
    for name, T0 in cut_cases.primal2_tableaux(oracle):
        T = T0.copy()
        rc, piv, log = oracle.primal2_solve(T, print_steps=False, hard_cap=3000)
        tab = Tableau.from_array(engine, T0)
        res = tab.primal2_solve(print_steps=False, hard_cap=3000)
        assert res.status == PRIM_STATUS[rc], name
        assert res.pivots == piv, name
        assert tab.cut_log() == log, name
        assert tab.read().tobytes() == T.tobytes(), name
        assert res.z == T[0, -1]
        tab.destroy()


@pytest.mark.parametrize("max_cuts", [1, 6])
def test_cutting_plane_matches_oracle(engine, oracle, max_cuts):
    from lpr_381_group_v22_amd import Tableau
    exits = set()
    for name, T0 in cut_cases.cutting_plane_tableaux(oracle):
        rc, cuts, T, log = oracle.cutting_plane(T0, max_cuts=max_cuts, hard_cap=2000)
        tab = Tableau.from_array(engine, T0)
        ex, ncuts = tab.cutting_plane(max_cuts=max_cuts, hard_cap=2000)
        assert (ex, ncuts) == (rc, cuts), name
        assert tab.cut_log() == log, name
        got = tab.read()
        assert got.shape == T.shape, name
        assert got.tobytes() == T.tobytes(), name
        exits.add(ex)
        tab.destroy()
    assert len(exits) >= 2


def test_cut_path_then_primal_path_on_grown_tableau(engine, oracle):
    """After cuts have grown the tableau the ordinary solver entry points still work on it."""
    from lpr_381_group_v22_amd import Tableau
    name, T0 = cut_cases.cutting_plane_tableaux(oracle)[0]
    rc, cuts, T, log = oracle.cutting_plane(T0, max_cuts=2, hard_cap=2000)
    tab = Tableau.from_array(engine, T0)
    tab.cutting_plane(max_cuts=2, hard_cap=2000)
    st, piv, plog = oracle.primal_solve(T, None, 50)
    res = tab.solve(max_pivots=50)
    assert res.status == st and res.pivots == piv
    assert tab.read().tobytes() == T.tobytes()
    tab.destroy()


def test_solve_cut_solve_cut_solve_on_one_handle(engine, oracle):
    """lpr_primal_solve (one-pivot path: captured graph) / lpr_cutting_plane alternating on one
    handle: every cut adds a row (and may or may not re-allocate), so a graph captured for the old
    row count must not be replayed; each leg is checked against the oracle."""
    from lpr_381_group_v22_amd import Tableau
    checked = 0
    for name, T0 in cut_cases.cutting_plane_tableaux(oracle)[:4]:
        T = T0.copy()
        tab = Tableau.from_array(engine, T0)
        for leg in range(3):
            st, piv, plog = oracle.primal_solve(T, None, 40)
            res = tab.solve(max_pivots=40, variant=0x7fff, batch=4)   # two-kernel graph path
            assert res.status == st and res.pivots == piv, (name, leg)
            assert tab.read().tobytes() == T.tobytes(), (name, leg)
            rc, cuts, T, log = oracle.cutting_plane(T, max_cuts=1, hard_cap=2000)
            ex, ncuts = tab.cutting_plane(max_cuts=1, hard_cap=2000)
            assert (ex, ncuts) == (rc, cuts), (name, leg)
            got = tab.read()
            assert got.shape == T.shape and got.tobytes() == T.tobytes(), (name, leg)
            checked += 1
        # and through the K-pivot path on the grown tableau
        st, piv, plog = oracle.primal_solve(T, None, 40)
        res = tab.solve(max_pivots=40, variant=0x4008, block=4)
        assert res.status == st and res.pivots == piv, name
        assert tab.read().tobytes() == T.tobytes(), name
        tab.destroy()
    assert checked >= 9
