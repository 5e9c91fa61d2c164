"""GPU parity tests at the sizes of BASELINE.json's remaining configs:
  configs[3]  Branch & Bound with ~256 live LP sub-problems (level-synchronous, batched children)
  configs[4]  degenerate / tie-saturated LP, m = n = 2048 (anti-cycling, wavefront arg-min stress)
(configs[0] sample LP, configs[1] m=512 and configs[2] m=4096 live in test_primal_gpu.py /
test_revised_gpu.py.)"""
import hashlib
import struct

import numpy as np
import pytest

import bb_cases
from oracle_evaluator import OracleEvaluator

pytestmark = pytest.mark.gpu


def bits(x):
    return struct.pack(">d", float(x)).hex()


def degenerate_tableau(m: int, n: int, seed: int):
    """Tie-saturated LP: small-integer coefficients, a third of the right-hand sides zero (every
    early ratio test is a many-way tie at ratio 0), objective coefficients drawn from 4 values
    (many-way ties in the entering scan).  All values stay far inside 2^+-50."""
    rng = np.random.RandomState(seed)
    h = m // 2
    A0 = rng.randint(1, 7, size=(h, n)).astype(np.float64)
    b0 = (rng.randint(20, 60, size=h) * n // 8).astype(np.float64)
    # every constraint appears twice: each ratio test has an exact two-way tie at its minimum and
    # the twin row is left with a zero right-hand side (a degenerate vertex) after the pivot
    A = np.vstack([A0, A0])
    b = np.concatenate([b0, b0])
    c = rng.randint(1, 5, size=n).astype(np.float64)
    T = np.zeros((m + 1, n + m + 1))
    T[0, :n] = -c
    T[1:, :n] = A
    T[1:, n:n + m] = np.eye(m)
    T[1:, -1] = b
    basis = (n + np.arange(m)).astype(np.int32)
    return T, basis


def test_config4_degenerate_m2048_fixed_pivot_budget(engine, oracle):
    from lpr_381_group_v22_amd import Tableau
    m = n = 2048
    K = 160
    T, basis = degenerate_tableau(m, n, 4)
    tab = Tableau.from_array(engine, T, basis)
    st, piv, log = oracle.primal_solve(T, basis, K)
    res = tab.solve(max_pivots=K)
    assert res.status == st and res.pivots == piv
    assert tab.pivot_log().tolist() == log.tolist(), "pivot indices differ from the oracle"
    assert tab.basis().tolist() == basis.tolist()
    assert piv >= 100, "fixture must need a long pivot sequence"
    assert int((T[1:, -1] == 0).sum()) >= 1, "fixture must sit on a degenerate vertex"
    got = tab.read()
    assert hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(T.tobytes()).hexdigest()
    tab.destroy()


def test_config4_klee_minty_bounded_full_path(engine, oracle):
    """Klee-Minty-style cube (growth base 2, d = 14): Dantzig's rule walks an exponentially long
    path; every pivot index must match the oracle."""
    import lp_cases
    from lpr_381_group_v22_amd import Constraint, PrimalSimplexSolver
    obj, cons, is_max = lp_cases.klee_minty_bounded(14)
    o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
    T, basis = oracle.primal_build(o, A, rel, rhs, is_max, ncoef)
    st, piv, log = oracle.primal_solve(T, basis, 100000, log_cap=1 << 17)
    s = PrimalSimplexSolver(obj, [Constraint(list(c.Coefficients), c.Relation, c.RHS)
                                  for c in cons], is_max, engine=engine, snapshots="none")
    s.Solve(max_pivots=100000)
    assert s.Status == st == 0
    assert s.PivotLog.tolist() == log.tolist()
    assert s.GetFinalTableau().tobytes() == T.tobytes()
    assert bits(s.FinalZ) == bits(T[0, -1])


def test_config3_bb_hundreds_of_live_subproblems(engine, oracle):
    """~256 live LP sub-problems: the frontier of a 9-level tree evaluated level by level, all
    children of a level in ONE batched expand on the GPU; same driver with the oracle-backed
    evaluator as the checker."""
    from lpr_381_group_v22_amd import BranchBoundTree, solve_level_synchronous
    obj, cons = bb_cases.random_binary_program(40, 5, 77)
    st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
    levels = 9
    tree = BranchBoundTree.from_array(engine, T, n, max_depth=levels + 2)

    class Counting:
        def __init__(self, inner):
            self.inner = inner
            self.max_batch = 0

        def node_info(self, ids):
            return self.inner.node_info(ids)

        def expand(self, p, v, b, k):
            self.max_batch = max(self.max_batch, len(p))
            return self.inner.expand(p, v, b, k)

        def release(self, ids):
            return self.inner.release(ids)

    gpu_ev = Counting(tree)
    got = solve_level_synchronous(gpu_ev, n, max_levels=levels)
    tree.destroy()
    # the same through the library's own driver (lpr_bb_solve_level_sync)
    from lpr_381_group_v22_amd import solve_level_sync_native
    tree2 = BranchBoundTree.from_array(engine, T, n, max_depth=levels + 2)
    nat = solve_level_sync_native(tree2, max_levels=levels)
    tree2.destroy()
    assert (nat["processed"], nat["pivots"], nat["levels"], nat["found"]) == \
        (got["processed"], got["pivots"], got["levels"], got["found"])
    assert bits(nat["z"]) == bits(got["z"]) and nat["path"] == (
        tuple(got["path"]) if got["path"] is not None else None)
    ref = solve_level_synchronous(OracleEvaluator(oracle, T, n), n, max_levels=levels)
    assert gpu_ev.max_batch >= 128, f"largest batch was only {gpu_ev.max_batch} children"
    assert got["processed"] == ref["processed"] and got["pivots"] == ref["pivots"]
    assert got["levels"] == ref["levels"]
    assert got["found"] == ref["found"] and bits(got["z"]) == bits(ref["z"])
    if ref["found"]:
        assert [bits(v) for v in got["x"]] == [bits(v) for v in ref["x"]]
        assert tuple(got["path"]) == tuple(ref["path"])


class _Recording:
    """Evaluator wrapper that keeps what every level saw: z and decision values of every node."""

    def __init__(self, inner):
        self.inner, self.levels, self.max_batch = inner, [], 0

    def node_info(self, ids):
        zs, vals = self.inner.node_info(ids)
        self.levels.append((np.array(zs, dtype=np.float64).copy(),
                            np.array(vals, dtype=np.float64).copy()))
        return zs, vals

    def expand(self, p, v, b, k):
        self.max_batch = max(self.max_batch, len(p))
        return self.inner.expand(p, v, b, k)

    def release(self, ids):
        return self.inner.release(ids)


def test_config3_bench_instance_six_levels_vs_oracle(engine, oracle):
    """Branch & Bound at the size bench.py runs -- the 512-variable binary
    programme of bench.bb_instance (577 x 1089 root, BranchBoundSimplexSolver.cs:289-468,694-803
    on M-sized tableaux) -- six levels (127 nodes scored, 126 children solved, 64 live at the
    widest level) against the oracle-backed evaluator: z and all 512 decision values of EVERY node
    of EVERY level bit for bit, node / pivot counts, and the library's own driver
    (lpr_bb_solve_level_sync) against both."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import bb_instance
    import lpr_381_group_v22_amd as pkg
    from lpr_381_group_v22_amd import (BranchBoundTree, Constraint, solve_level_sync_native,
                                       solve_level_synchronous)
    nv, nc, levels = 512, 64, 6
    c, A, b = bb_instance(nv, nc, 7)
    cons = [Constraint(A[i].tolist(), "<=", float(b[i])) for i in range(nc)]
    for i in range(nv):  # Program.cs:372-382
        co = [0.0] * (nv + 3)
        co[i] = 1.0
        co[nv + 1] = 1.0
        cons.append(Constraint(co, "<=", 1.0))
    primal = pkg.PrimalSimplexSolver(c.tolist(), cons, True, engine=engine, snapshots="none")
    primal.Solve()
    T = primal.tableau.read()
    assert T.shape == (577, 1089)
    tree = BranchBoundTree.from_tableau(primal.tableau, nv, max_depth=levels + 2)
    gpu = _Recording(tree)
    got = solve_level_synchronous(gpu, nv, max_levels=levels)
    tree.destroy()
    cpu = _Recording(OracleEvaluator(oracle, T, nv))
    ref = solve_level_synchronous(cpu, nv, max_levels=levels)
    assert len(gpu.levels) == len(cpu.levels) == levels + 1
    for lv, ((gz, gv), (cz, cv)) in enumerate(zip(gpu.levels, cpu.levels)):
        assert gz.shape == cz.shape and gz.tobytes() == cz.tobytes(), f"z differs at level {lv}"
        assert gv.tobytes() == cv.tobytes(), f"decision values differ at level {lv}"
    assert gpu.max_batch == cpu.max_batch >= 64
    for k in ("processed", "pivots", "levels", "found", "status"):
        assert got[k] == ref[k], k
    assert got["processed"] == 127 and bits(got["z"]) == bits(ref["z"])
    tree2 = BranchBoundTree.from_tableau(primal.tableau, nv, max_depth=levels + 2)
    nat = solve_level_sync_native(tree2, max_levels=levels)
    tree2.destroy()
    for k in ("processed", "pivots", "levels", "found", "status"):
        assert nat[k] == ref[k], k
    assert bits(nat["z"]) == bits(ref["z"])
    if ref["found"]:
        assert [bits(v) for v in nat["x"]] == [bits(v) for v in ref["x"]]
        assert nat["path"] == tuple(ref["path"])
