#!/usr/bin/env python3
"""Time-bounded randomised comparison of the Branch & Bound path on the device with the C oracle:
node records, pop order, every dual / primal pivot of every child (incl. dropped last tableaux),
incumbent bits -- lpr_bb_run (the reference's DFS) and, on the same instance, the level-synchronous
driver against the oracle-backed evaluator:  python tools/fuzz_bb_gpu.py [seconds] [first seed]"""
import struct
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import bb_cases  # noqa: E402
from oracle_evaluator import OracleEvaluator  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
import lpr_381_group_v22_amd as pkg  # noqa: E402
from lpr_381_group_v22_amd import BranchBoundTree, solve_level_sync_native, solve_level_synchronous  # noqa: E402


def bits(x):
    return struct.pack(">d", float(x)).hex()


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
oracle = Oracle()
eng = pkg.Engine(0)
t_end = time.time() + budget
cases = drops = primal = 0
while time.time() < t_end:
    rng = np.random.RandomState(seed)
    n, mc = int(rng.randint(3, 14)), int(rng.randint(1, 6))
    gen = bb_cases.random_binary_program if rng.randint(0, 2) else bb_cases.fractional_program
    obj, cons = gen(n, mc, int(rng.randint(0, 1 << 30)))
    st, T, nn = bb_cases.primal_final_tableau(oracle, obj, cons)
    if st != 0:
        seed += 1
        continue
    cap = int(rng.choice([20, 20, 40, 7]))
    ref = oracle.bb_solve(T, nn, node_cap=cap, rec_cap=1 << 12, piv_cap=1 << 18)
    tree = BranchBoundTree.from_array(eng, T, nn, max_depth=max(cap, 20))
    res, x = tree.run(node_cap=cap)
    tag = (seed, n, mc, cap)
    assert res.status == ref["status"] and bool(res.found) == ref["found"], tag
    assert tree.pop_order() == ref["pop_order"] and tree.records() == ref["records"], tag
    assert tree.trace() == ref["trace"], tag
    if ref["found"]:
        assert bits(res.z) == bits(ref["z"]) and [bits(v) for v in x] == [bits(v) for v in ref["x"]], tag
    tree.destroy()
    drops += sum(1 for t in ref["trace"] if t[1] == 2)
    primal += sum(1 for t in ref["trace"] if t[1] == 1)
    # level-synchronous driver, 4 levels, against the oracle-backed evaluator
    t2 = BranchBoundTree.from_array(eng, T, nn, max_depth=24)
    got = solve_level_sync_native(t2, max_levels=4)
    t2.destroy()
    want = solve_level_synchronous(OracleEvaluator(oracle, T, nn), nn, max_levels=4)
    for k in ("processed", "pivots", "levels", "found", "status"):
        assert got[k] == want[k], (tag, k, got[k], want[k])
    if want["found"]:
        assert bits(got["z"]) == bits(want["z"]) and got["path"] == tuple(want["path"]), tag
    cases += 1
    seed += 1
    if cases % 50 == 0:
        print(f"{cases} cases, seed {seed}, {primal} primal pivots, {drops} dropped tableaux", flush=True)
print(f"OK: {cases} instances, seeds {seed0}..{seed - 1}: records, pop order, pivot traces ({primal} primal "
      f"pivots, {drops} dropped tableaux), incumbents identical to the oracle")
eng.close()
