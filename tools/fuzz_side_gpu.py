#!/usr/bin/env python3
"""Time-bounded randomised comparison of the two side paths on the device with the C oracle:
  * cutting-plane path (rows f3): lpr_dual_solve, lpr_primal2_solve, lpr_cutting_plane -- exit code,
    pivot count, (row, column) log, every byte of the tableau;
  * sensitivity re-solve (row f4): random edit scripts on one analyzer (indices in and out of range,
    basic and non-basic columns, edits that are rolled back or leave the state mid-way) -- outcome
    code and tableau / basis / solution / Z / pivot log after EVERY edit.
    python tools/fuzz_side_gpu.py [seconds] [first seed]"""
import math
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import bb_cases  # noqa: E402
import lp_cases  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
import lpr_381_group_v22_amd as pkg  # noqa: E402
from lpr_381_group_v22_amd import Tableau  # noqa: E402
from lpr_381_group_v22_amd.engine import SensState  # noqa: E402

DUAL_STATUS = {0: 0, 1: 2, 3: 3, 5: 5}
PRIM_STATUS = {0: 0, 1: 1, 3: 3, 5: 5}


def fail(*what):
    print("MISMATCH", *what)
    sys.exit(1)


def random_lp(rng, integer):
    m, n = int(rng.randint(2, 14)), int(rng.randint(2, 18))
    seed = int(rng.randint(0, 1 << 30))
    if integer:
        obj, cons, _ = lp_cases.tie_heavy(m, n, seed)
        cons = [type(c)(c.Coefficients, "<=", abs(c.RHS) + float(rng.randint(0, 3))) for c in cons]
    else:
        obj, cons, _ = lp_cases.random_dense(m, n, seed)
    return m, n, obj, cons


def build(oracle, obj, cons):
    o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
    return oracle.primal_build(o, A, rel, rhs, True, ncoef)


def cut_case(oracle, eng, rng, seed):
    kind = int(rng.randint(0, 3))
    if kind == 0:      # PrimalSimplexSolver2 on an initial tableau
        m, n, obj, cons = random_lp(rng, bool(rng.randint(0, 2)))
        T0, _ = build(oracle, obj, cons)
        ps, cap = bool(rng.randint(0, 2)), int(rng.choice([3, 40, 3000]))
        T = T0.copy()
        rc, piv, log = oracle.primal2_solve(T, print_steps=ps, hard_cap=cap)
        tab = Tableau.from_array(eng, T0)
        res = tab.primal2_solve(print_steps=ps, hard_cap=cap)
        if res.status != PRIM_STATUS[rc] or res.pivots != piv or tab.cut_log() != log:
            fail(seed, "primal2", res.status, rc, res.pivots, piv)
        if tab.read().tobytes() != T.tobytes():
            fail(seed, "primal2 tableau")
        tab.destroy()
        return "primal2"
    # an optimal tableau of a small integer programme
    nv, mc = int(rng.randint(3, 10)), int(rng.randint(1, 5))
    gen = bb_cases.random_binary_program if rng.randint(0, 2) else bb_cases.fractional_program
    obj, cons = gen(nv, mc, int(rng.randint(0, 1 << 30)))
    st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
    if st != 0:
        return None
    if kind == 1:      # DualSimplexSolver on it + one branching row (violated or not)
        k = int(rng.randint(0, n))
        con = np.zeros(n + 2)
        con[k] = 1.0
        con[n] = float(rng.randint(-1, 3))
        con[n + 1] = float(rng.randint(0, 2))
        T0 = oracle.bb_add_constraint(T, con)
        ps, mi = bool(rng.randint(0, 2)), int(rng.choice([1, 5, 1000]))
        Tw = T0.copy()
        rc, piv, log = oracle.dual_solve(Tw, max_iters=mi, print_steps=ps, hard_cap=2000)
        tab = Tableau.from_array(eng, T0)
        res = tab.dual_solve(max_iters=mi, print_steps=ps, hard_cap=2000)
        if res.status != DUAL_STATUS[rc] or res.pivots != piv or tab.cut_log() != log:
            fail(seed, "dual", res.status, rc, res.pivots, piv)
        if tab.read().tobytes() != Tw.tobytes():
            fail(seed, "dual tableau")
        tab.destroy()
        return "dual"
    mc_ = int(rng.choice([1, 2, 6]))
    rc, cuts, Tw, log = oracle.cutting_plane(T, max_cuts=mc_, hard_cap=2000)
    tab = Tableau.from_array(eng, T)
    ex, ncuts = tab.cutting_plane(max_cuts=mc_, hard_cap=2000)
    got = tab.read()
    if (ex, ncuts) != (rc, cuts) or tab.cut_log() != log:
        fail(seed, "cutting plane", ex, rc, ncuts, cuts)
    if got.shape != Tw.shape or got.tobytes() != Tw.tobytes():
        fail(seed, "cutting plane tableau")
    tab.destroy()
    return "cut%d" % rc


def same_state(d, o, tag):
    T, basic, sol = d.read()
    st = o.state()
    if T.shape != st["T"].shape or T.tobytes() != st["T"].tobytes():
        fail(tag, "tableau")
    if basic.tolist() != st["basic"] or sol.tobytes() != st["sol"].tobytes():
        fail(tag, "basis / solution")
    z = d.shape()[4]
    if not (z == st["z"] or (math.isnan(st["z"]) and math.isnan(z))):
        fail(tag, "z", z, st["z"])
    if d.log() != o.log():
        fail(tag, "pivot log")


def sens_case(oracle, eng, rng, seed):
    m, n, obj, cons = random_lp(rng, bool(rng.randint(0, 2)))
    T, basis = build(oracle, obj, cons)
    st, piv, log = oracle.primal_solve(T, basis)
    if st != 0:
        return None
    x, z = oracle.extract_solution(T, n)
    o = oracle.sens(T, x, z, basis)
    d = SensState.create(eng, T, x, z)
    same_state(d, o, (seed, "ctor"))
    codes = []
    for k in range(int(rng.randint(3, 10))):
        R, C = o.state()["T"].shape
        op = int(rng.randint(0, 7))
        if op == 0:
            name, args = "resolve_all", ()
        elif op == 1:
            name, args = "change_nonbasic_cbar", (int(rng.randint(-1, C + 1)),
                                                  float(np.round(rng.uniform(-5, 5), 3)))
        elif op == 2:
            name, args = "change_basic", (int(rng.randint(-1, C + 1)),
                                          float(np.round(rng.uniform(-3, 3), 3)))
        elif op == 3:
            name, args = "change_rhs", (int(rng.randint(-1, R + 1)),
                                        float(np.round(rng.uniform(-10, 30), 2)))
        elif op == 4:
            name, args = "change_nonbasic_column", (int(rng.randint(-1, R + 1)),
                                                    int(rng.randint(-1, C + 1)),
                                                    float(np.round(rng.uniform(-2, 2), 3)))
        elif op == 5:
            a = np.round(rng.uniform(-0.5, 1.0, size=R - 1), 3)
            name, args = "add_activity", (float(np.round(rng.uniform(0, 10), 2)), a.tolist())
        else:
            t = np.where(rng.uniform(size=C - 1) < 0.4, rng.randint(0, 4, size=C - 1), 0)
            name, args = "add_constraint", (t.astype(float).tolist(),
                                            float(np.round(rng.uniform(-2, 10), 2)))
        rc = getattr(o, name)(*args)
        oc = getattr(d, name)(*args)
        if oc != rc:
            fail(seed, k, name, args, "code", oc, rc)
        same_state(d, o, (seed, k, name, args))
        codes.append(rc)
        if o.state()["T"].shape[0] > 40 or o.state()["T"].shape[1] > 80:
            break
    d.destroy()
    return codes


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    oracle = Oracle()
    eng = pkg.Engine(0)
    t_end = time.time() + budget
    kinds, codes, edits = {}, {}, 0
    while time.time() < t_end:
        rng = np.random.RandomState(seed)
        if seed % 2:
            k = cut_case(oracle, eng, rng, seed)
            if k:
                kinds[k] = kinds.get(k, 0) + 1
        else:
            cs = sens_case(oracle, eng, rng, seed)
            if cs is not None:
                kinds["sens"] = kinds.get("sens", 0) + 1
                edits += len(cs)
                for c in cs:
                    codes[c] = codes.get(c, 0) + 1
        seed += 1
        if (seed - seed0) % 200 == 0:
            print(f"{seed - seed0} cases, seed {seed}: {kinds}, {edits} edits, codes {codes}", flush=True)
    print(f"OK: seeds {seed0}..{seed - 1}: {kinds}; {edits} sensitivity edits with outcome codes "
          f"{codes}: identical to the oracle")


if __name__ == "__main__":
    main()
