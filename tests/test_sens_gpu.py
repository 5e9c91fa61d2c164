"""GPU parity tests of the sensitivity re-solve (lpr_sens_*, csrc/sens_engine.hip) against the C
oracle (oracle/oracle_sens.c): outcome code, pivot log, tableau / basis / solution / Z bits after
every edit of a script, the device-to-device hand-over from a solved primal tableau, the column
folds and the host mirror's text."""
import math

import numpy as np
import pytest

import sens_cases

pytestmark = pytest.mark.gpu


def _same_state(dev, orc, tag):
    T, basic, sol = dev.read()
    st = orc.state()
    assert (T.shape == st["T"].shape), tag
    assert T.tobytes() == st["T"].tobytes(), tag
    assert basic.tolist() == st["basic"], tag
    assert sol.tobytes() == st["sol"].tobytes(), tag
    assert dev.shape()[4] == st["z"] or (math.isnan(st["z"]) and math.isnan(dev.shape()[4])), tag
    assert dev.log() == orc.log(), tag


def _run_script(engine, oracle, name, base, ops):
    from lpr_381_group_v22_amd.engine import SensState
    T, x, z, basis = base
    o = oracle.sens(T, x, z, basis)
    d = SensState.create(engine, T, x, z)
    _same_state(d, o, (name, "ctor"))
    codes = []
    for k, (op, args) in enumerate(ops):
        op, args = sens_cases.materialize(op, args, o.state()["T"], k)
        rc = getattr(o, op)(*args)
        oc = getattr(d, op)(*args)
        assert oc == rc, (name, k, op, oc, rc)
        _same_state(d, o, (name, k, op))
        codes.append(rc)
    d.destroy()
    return codes


def test_edit_scripts_match_oracle(engine, oracle):
    seen = set()
    for name, base, ops in sens_cases.scripts(oracle):
        seen |= set(_run_script(engine, oracle, name, base, ops))
    assert {0, -1, 8, 1, 2} <= seen, seen


def test_larger_instances_match_oracle(engine, oracle):
    """A few hundred rows: many dual pivots per RHS edit, tableau grown twice."""
    for (m, n, seed) in [(96, 160, 11), (200, 120, 12)]:
        base = sens_cases.solved_lp(oracle, m, n, seed)
        T, x, z, basis = base
        rng = np.random.RandomState(seed)
        bset = set(int(b) for b in basis)
        nonbasic = [j for j in range(T.shape[1] - 1) if j not in bset]
        ops = [("change_rhs", (3, float(T[3, -1]) * 0.25)),
               ("change_nonbasic_cbar", (nonbasic[2], -1.0)),
               ("add_activity", (3.0, rng.uniform(0.05, 0.5, size=m).tolist())),
               ("add_constraint", (None, 2.0)),
               ("change_basic", (int(basis[0]), -0.5)),
               ("change_rhs", (m // 2, float(T[m // 2, -1]) + 10.0)),
               ("resolve_all", ())]
        _run_script(engine, oracle, f"big_{m}x{n}", base, ops)


def test_hand_over_from_a_solved_primal_tableau(engine, oracle):
    import lp_cases
    from lpr_381_group_v22_amd import Tableau
    from lpr_381_group_v22_amd.engine import SensState
    m, n = 24, 36
    obj, cons, _ = lp_cases.random_dense(m, n, 5)
    o_, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
    T, basis = oracle.primal_build(o_, A, rel, rhs, True, ncoef)
    st, piv, log = oracle.primal_solve(T, basis)
    assert st == 0
    x, z = oracle.extract_solution(T, n)
    tab = Tableau.from_lp(engine, o_, A, rel, rhs, is_max=True, ncoef=ncoef)
    assert tab.solve().status == 0
    d = SensState.from_tableau(tab, n)
    o = oracle.sens(T, x, z, basis)
    _same_state(d, o, "hand-over")
    assert d.change_rhs(2, float(T[2, -1]) * 0.5) == o.change_rhs(2, float(T[2, -1]) * 0.5)
    _same_state(d, o, "hand-over + rhs")
    # the analyzer owns a copy: the primal tableau is untouched
    assert tab.read().tobytes() == T.tobytes()
    d.destroy()
    tab.destroy()


def test_basic_row_and_column_fold(engine, oracle):
    from lpr_381_group_v22_amd.engine import SensState
    T, x, z, basis = sens_cases.solved_lp(oracle, 40, 56, 9)
    d = SensState.create(engine, T, x, z)
    R, C = T.shape
    for col in range(C - 1):
        exp = -1
        for i in range(1, R):
            if abs(T[i, col] - 1.0) < 1e-9 and all(
                    k == i or abs(T[k, col]) <= 1e-9 for k in range(1, R)):
                exp = i
                break
        assert d.basic_row(col) == exp, col
    rng = np.random.RandomState(3)
    w = rng.uniform(-1, 1, size=R - 1)
    init = rng.uniform(-1, 1, size=C - 1)
    got = d.column_fold(w, init, C - 1)
    exp = init.copy()
    for i in range(R - 1):          # same order, every product rounded on its own
        exp = exp + w[i] * T[i + 1, :C - 1]
    assert got.tobytes() == exp.tobytes()
    got0 = d.column_fold(w, None, 7)
    exp0 = np.zeros(7)
    for i in range(R - 1):
        exp0 = exp0 + w[i] * T[i + 1, :7]
    assert got0.tobytes() == exp0.tobytes()
    d.destroy()


def test_host_mirror_text_and_ranges(engine, oracle):
    from lpr_381_group_v22_amd.sensitivity_analyzer import (F, InvalidOperationException,
                                                            SensitivityAnalyzer)
    T, x, z, basis = sens_cases.solved_lp(oracle, 4, 6, 0)
    sa = SensitivityAnalyzer(T, list(x), z, [int(b) for b in basis], engine=engine)
    o = oracle.sens(T, x, z, basis)
    R, C = T.shape
    m, n = R - 1, C - R
    bset = set(o.state()["basic"])
    nonbasic = [j for j in range(C - 1) if j not in bset]
    j = nonbasic[0]
    sa.Out.clear()
    sa.DisplayRangeNonBasic(j + 1)
    assert sa.Out[0] == f"Reduced Cost for {sa.ColLabel(j)}: {F(T[0, j])}"
    # RHS range of constraint 1 from the oracle's tableau
    y, lo, hi, cur = sa.RHSRange(1)
    s_col = n
    elo, ehi = -math.inf, math.inf
    for i in range(1, R):
        c, b = T[i, s_col], T[i, -1]
        if c > 1e-9:
            elo = max(elo, -b / c)
        elif c < -1e-9:
            ehi = min(ehi, -b / c)
    assert (y, lo, hi, cur) == (T[0, s_col], elo, ehi, T[1, -1])
    # duality: c-hat from the device fold equals the sequential host sum
    chat = sa.RecoverObjectiveC()
    exp = np.zeros(n)
    for i in range(m):
        exp = exp + T[i + 1, :n] * T[0, n + i]
    exp = exp - T[0, :n]
    assert chat.tobytes() == exp.tobytes()
    sa.PerformDuality()
    assert any(line.startswith("  y* = [") for line in sa.Out)
    # edits through the mirror follow the oracle and print the resolved tableau
    sa.Out.clear()
    assert sa.ChangeRHS(1, float(T[1, -1]) + 3.0) == o.change_rhs(1, float(T[1, -1]) + 3.0)
    assert sa.Out[0] == "\n=== After RHS change (resolved) ==="
    assert sa.CurrentTableau.tobytes() == o.state()["T"].tobytes()
    assert sa.CurrentZ == o.state()["z"]
    sa.Out.clear()
    assert sa.ChangeRHS(2, -5.0) == 8 == o.change_rhs(2, -5.0)
    assert sa.Out[0].startswith("This RHS change makes the model infeasible")
    assert sa.ChangeNonBasicReducedCost(10 ** 6, 1.0) == -1
    assert sa.Out[-1] == "Invalid index or variable is basic."
    # an infeasible new constraint raises the C#'s exception
    width = sa.numCols - 1
    assert o.add_constraint([1.0] * width, -1.0) == 2
    with pytest.raises(InvalidOperationException, match="Infeasible after RHS change"):
        sa.AddNewConstraintNonInteractive([1.0] * width, -1.0)
    assert sa.CurrentTableau.tobytes() == o.state()["T"].tobytes()
    # and an activity that uses nothing is unbounded
    rows = sa.numRows
    assert o.add_activity(50.0, [-1.0] * (rows - 1)) == 1
    with pytest.raises(InvalidOperationException, match="Unbounded during re-optimization"):
        sa.AddNewActivity(50.0, [-1.0] * (rows - 1))


def test_full_size_resolve_properties(engine):
    """BASELINE-size tableau (m=4096, n=8192): solve on the device, hand over, push one RHS far
    outside its range and check the invariants the re-solve must restore."""
    from lpr_381_group_v22_amd import Tableau
    from lpr_381_group_v22_amd.engine import SensState
    m, n = 4096, 8192
    tab = Tableau.synthetic(engine, m, n, 7)
    res = tab.solve()
    assert res.status == 0
    d = SensState.from_tableau(tab, n)
    tab.destroy()
    R, C = m + 1, n + m + 1
    rhs0 = d.read_block(0, R, C - 1, 1)[:, 0]
    z0 = rhs0[0]
    assert d.resolve_all() == 0 and d.shape()[5] == 0      # optimal stays optimal, no pivots
    k = int(np.argmax(rhs0[1:])) + 1
    oc = d.change_rhs(k, -1.0)                              # slack row forced negative
    assert oc in (0, 8)
    rhs = d.read_block(0, R, C - 1, 1)[:, 0]
    row0 = d.read_block(0, 1, 0, C - 1)[0]
    _, basic, sol = d.read(tableau=False)
    if oc == 0:
        assert d.shape()[5] > 0
        assert rhs[1:].min() >= -1e-9
        nb = np.ones(C - 1, dtype=bool)
        nb[basic[basic >= 0]] = False
        assert row0[nb].min() >= -1e-9
        assert d.shape()[4] == rhs[0]
        assert sol.shape[0] == C - 1
        # every basic column's solution entry is its row's RHS
        for i in (0, m // 2, m - 1):
            if basic[i] >= 0:
                assert sol[basic[i]] == rhs[i + 1]
    else:
        assert rhs.tobytes() == rhs0.tobytes() and d.shape()[4] == z0
    d.destroy()


def test_division_is_ieee_in_the_dual_pivot(engine, oracle):
    """The pair tools/fuzz_side_gpu.py found (seed 170796): the normalised pivot row of a dual
    pivot holds a / b with the exact quotient 2e-16 ulp from a midpoint; the hardware's division
    sequence gives 0.35, IEEE (and the C#) 0.35000000000000003 (engine_common.hpp: ieee_div)."""
    from lpr_381_group_v22_amd.engine import SensState
    a, b = -float.fromhex("0x1.6666666666663p+0"), -float.fromhex("0x1.ffffffffffffbp+1")
    T = np.array([[0.0, 1.0, 0.0, 0.0, 0.0],
                  [b, a, 1.0, 0.0, -1.0],
                  [0.5 * b, 2.0, 0.0, 1.0, 7.0]])
    x = np.zeros(4)
    o = oracle.sens(T, x, 0.0, np.zeros(2, dtype=np.int32))
    d = SensState.create(engine, T, x, 0.0)
    assert o.resolve_all() == d.resolve_all() == 0
    assert o.log() == d.log() == [(0, 1, 0)]
    assert o.state()["T"][1, 1] == 0.35000000000000003
    _same_state(d, o, "hard division")
    d.destroy()
