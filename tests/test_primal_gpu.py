"""GPU parity tests of the primal-simplex hot path: the HIP engine, called through the C ABI,
against the CPU oracle on the same inputs.  Bar: bit-exact tableau, identical pivot log / basis /
status, bit-exact Z and x (the path is IEEE binary64 with no re-association)."""
import hashlib
import struct

import numpy as np
import pytest

import lp_cases

pytestmark = pytest.mark.gpu

STATUS = {0: "optimal", 1: "unbounded", 5: "limit"}


def bits(x):
    return struct.pack(">d", float(x)).hex()


def oracle_run(oracle, obj, cons, is_max, max_pivots=0):
    o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
    T, basis = oracle.primal_build(o, A, rel, rhs, is_max, ncoef)
    T0 = T.copy()
    st, piv, log = oracle.primal_solve(T, basis, max_pivots)
    x, z = oracle.extract_solution(T, len(obj))
    return dict(T0=T0, T=T, basis=basis, status=st, pivots=piv, log=log, x=x, z=z)


def to_constraints(cons):
    from lpr_381_group_v22_amd import Constraint
    return [Constraint(list(c.Coefficients), c.Relation, c.RHS) for c in cons]


@pytest.mark.parametrize("name,case", lp_cases.all_cases(), ids=[c[0] for c in lp_cases.all_cases()])
def test_solver_mirror_matches_oracle(engine, oracle, name, case):
    from lpr_381_group_v22_amd import PrimalSimplexSolver
    obj, cons, is_max = case
    ref = oracle_run(oracle, obj, cons, is_max, max_pivots=5000)
    s = PrimalSimplexSolver(obj, to_constraints(cons), is_max, engine=engine, snapshots="none")
    assert s.tableau.read().tobytes() == ref["T0"].tobytes(), "constructor (a-P0) differs"
    s.Solve(max_pivots=5000)
    assert s.Status == ref["status"]
    assert s.PivotLog.tolist() == ref["log"].tolist()
    assert s.BasicVariables == ref["basis"].tolist()
    assert s.GetFinalTableau().tobytes() == ref["T"].tobytes(), "final tableau not bit-identical"
    if ref["status"] == 0:
        assert bits(s.FinalZ) == bits(ref["z"])
        assert [bits(v) for v in s.SolutionVector] == [bits(v) for v in ref["x"]]
        assert s.FinalTableau.tobytes() == ref["T"].tobytes()
    elif ref["status"] == 1:  # PrimalSimplexSolver.cs:129-135: no throw, FinalZ stays 0, x null
        assert s.FinalZ == 0.0 and s.SolutionVector is None and s.FinalTableau is not None


def test_golden_fixture_on_gpu(engine):
    import json
    import os
    from lpr_381_group_v22_amd import PrimalSimplexSolver
    with open(os.path.join(os.path.dirname(__file__), "golden", "primal_golden.json")) as f:
        gold = json.load(f)
    cases = dict(lp_cases.all_cases())
    for name, g in gold.items():
        obj, cons, is_max = cases[name]
        s = PrimalSimplexSolver(obj, to_constraints(cons), is_max, engine=engine,
                                snapshots="none")
        s.Solve(max_pivots=5000)
        assert STATUS[s.Status] == g["status"], name
        assert s.PivotLog.tolist() == g["log"], name
        assert s.BasicVariables == g["basis"], name
        T = s.GetFinalTableau()
        assert bits(T[0, -1]) == g["z_bits"], name
        assert hashlib.sha256(T.tobytes()).hexdigest() == g["tableau_sha256"], name
        x, _ = s.tableau.extract_solution(len(obj))
        assert [bits(v) for v in x] == g["x_bits"], name


def test_single_step_entry_points(engine, oracle):
    """lpr_select_entering / lpr_select_leaving / lpr_pivot == FindEnteringVariable /
    FindLeavingVariable / Pivot, one call at a time, including the snapshot-"all" host loop."""
    from lpr_381_group_v22_amd import Tableau
    for case in (lp_cases.sample_option1(), lp_cases.tie_heavy(24, 30, 2),
                 lp_cases.random_dense(16, 32, 1)):
        obj, cons, is_max = case
        o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
        T, basis = oracle.primal_build(o, A, rel, rhs, is_max, ncoef)
        tab = Tableau.from_array(engine, T, basis)
        for _ in range(200):
            e = oracle.find_entering(T)
            assert tab.select_entering() == e
            if e < 0:
                break
            r = oracle.find_leaving(T, e)
            assert tab.select_leaving(e) == r
            if r < 0:
                break
            oracle.pivot(T, r, e)
            basis[r - 1] = e
            tab.pivot(r, e)
            assert tab.read().tobytes() == T.tobytes()
            assert tab.basis().tolist() == basis.tolist()
        tab.destroy()


def test_snapshot_policy_all_equals_batched(engine):
    from lpr_381_group_v22_amd import PrimalSimplexSolver
    obj, cons, is_max = lp_cases.sample_option1()
    a = PrimalSimplexSolver(obj, to_constraints(cons), is_max, engine=engine, snapshots="all")
    b = PrimalSimplexSolver(obj, to_constraints(cons), is_max, engine=engine, snapshots="none")
    a.Solve()
    b.Solve()
    assert a.GetFinalTableau().tobytes() == b.GetFinalTableau().tobytes()
    assert a.PivotLog.tolist() == b.PivotLog.tolist() == [[5, 3], [7, 5], [3, 1], [4, 2], [1, 0],
                                                           [1, 4]]
    assert bits(a.FinalZ) == bits(15.4) and a.BasicVariables == [4, 7, 1, 2, 3, 11, 5]
    # initial + one per pivot + final block (PrimalSimplexSolver.cs:86,148,119-122)
    assert len(a.IterationSnapshots) == 1 + 6 + 1
    assert "Iteration 6 - After pivot:" in a.IterationSnapshots[6]
    assert "Z = 15.400000" in a.IterationSnapshots[-1]


def test_synthetic_generator_matches_oracle(engine, oracle):
    from lpr_381_group_v22_amd import Tableau
    for (m, n, seed) in [(5, 8, 7), (64, 128, 0), (100, 37, 3)]:
        T, basis = oracle.gen_dense_tableau(m, n, seed)
        tab = Tableau.synthetic(engine, m, n, seed)
        assert tab.read().tobytes() == T.tobytes()
        assert tab.basis().tolist() == basis.tolist()
        tab.destroy()


def test_all_update_variants_and_paths_give_identical_bits(engine, oracle):
    """Tile shape, serpentine sweep, graph replay vs eager+events: same bits, same log."""
    from lpr_381_group_v22_amd import Tableau
    m, n, seed = 200, 333, 11
    T, basis = oracle.gen_dense_tableau(m, n, seed)
    st, piv, log = oracle.primal_solve(T, basis, 120)
    want = T.tobytes()
    for variant in list(range(1, 12)) + [0x100 + v for v in range(1, 12)]:
        for timed in (False, True):
            tab = Tableau.synthetic(engine, m, n, seed)
            res = tab.solve(max_pivots=120, variant=variant, time_kernels=timed, batch=32)
            assert res.status == st and res.pivots == piv
            assert tab.pivot_log().tolist() == log.tolist()
            assert tab.read().tobytes() == want, (variant, timed)
            if timed:
                launches, total_ms, avg_ms = tab.kernel_stats()
                assert 0 < launches <= piv and total_ms > 0 and avg_ms > 0
            tab.destroy()


def test_resume_after_pivot_limit(engine, oracle):
    from lpr_381_group_v22_amd import Tableau
    m, n, seed = 48, 96, 2
    T, basis = oracle.gen_dense_tableau(m, n, seed)
    st, piv, log = oracle.primal_solve(T, basis)
    tab = Tableau.synthetic(engine, m, n, seed)
    total = 0
    while True:
        res = tab.solve(max_pivots=7, batch=3)
        total += res.pivots
        assert res.total_pivots == total
        if res.status != 5:
            break
        assert res.pivots == 7
    assert res.status == st and total == piv
    assert tab.pivot_log().tolist() == log.tolist()
    assert tab.read().tobytes() == T.tobytes()
    tab.destroy()


def test_config1_m512_n1024_full_solve(engine, oracle):
    """BASELINE configs[1]: dense random LP m=512 n=1024, solved to optimality on both sides."""
    from lpr_381_group_v22_amd import Tableau
    m, n, seed = 512, 1024, 0
    T, basis = oracle.gen_dense_tableau(m, n, seed)
    st, piv, log = oracle.primal_solve(T, basis, 20000)
    x_ref, z_ref = oracle.extract_solution(T, n)
    tab = Tableau.synthetic(engine, m, n, seed)
    res = tab.solve(max_pivots=20000)
    assert res.status == st == 0 and res.pivots == piv
    assert tab.pivot_log(1 << 16).tolist() == log.tolist()
    assert tab.basis().tolist() == basis.tolist()
    assert tab.read().tobytes() == T.tobytes()
    x, z = tab.extract_solution(n)
    assert x.tobytes() == x_ref.tobytes() and bits(z) == bits(z_ref)
    tab.destroy()


def test_north_star_size_first_pivots_and_invariants(engine, oracle):
    """m=4096, n=8192 (4097 x 12289, 402.8 MB): the first pivots bit-for-bit against the oracle,
    then size-independent properties of the tableau after a longer run."""
    from lpr_381_group_v22_amd import Tableau
    m, n, seed, K = 4096, 8192, 0, 4
    T, basis = oracle.gen_dense_tableau(m, n, seed)
    tab = Tableau.synthetic(engine, m, n, seed)
    assert tab.read_block(0, 3, 0, n + m + 1).tobytes() == T[:3].tobytes()
    assert tab.read_block(4000, 97, 8000, 4289).tobytes() == \
        np.ascontiguousarray(T[4000:, 8000:]).tobytes()
    st, piv, log = oracle.primal_solve(T, basis, K)
    res = tab.solve(max_pivots=K)
    assert res.status == st == 5 and res.pivots == K
    assert tab.pivot_log().tolist() == log.tolist()
    got = tab.read()
    assert hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(T.tobytes()).hexdigest()
    del got, T
    # longer run: invariants that hold for any correct Gauss-Jordan pivot sequence
    z_prev = res.z
    res = tab.solve(max_pivots=60)
    assert res.pivots == 60 and res.z >= z_prev
    b = tab.basis()
    R, C = tab.rows, tab.cols
    logk = tab.pivot_log()
    r_last, e_last = logk[-1]
    col = tab.read_block(0, R, int(e_last), 1)[:, 0]
    unit = np.zeros(R)
    unit[r_last] = 1.0
    assert col.tobytes() == unit.tobytes(), "pivot column must be exactly e_r after the pivot"
    assert b[r_last - 1] == e_last
    rhs = tab.read_block(1, R - 1, C - 1, 1)[:, 0]
    assert (rhs >= 0).all(), "ratio test keeps the basis primal feasible"
    # every structural basic column is (numerically) a unit column, slack ones exactly identity
    for row in (1, R // 2, R - 1):
        cidx = int(b[row - 1])
        colv = tab.read_block(0, R, cidx, 1)[:, 0]
        assert abs(colv[row] - 1.0) < 1e-6 and np.abs(np.delete(colv, row)).max() < 1e-6
    tab.destroy()


def test_edge_shapes(engine, oracle):
    from lpr_381_group_v22_amd import Constraint, PrimalSimplexSolver
    # no constraints at all: R = 1; a positive objective is unbounded, a non-positive one optimal
    s = PrimalSimplexSolver([1.0, 2.0], [], True, engine=engine, snapshots="none")
    s.Solve()
    assert s.Status == 1 and s.SolutionVector is None
    s = PrimalSimplexSolver([-1.0, 0.0], [], True, engine=engine, snapshots="none")
    s.Solve()
    assert s.Status == 0 and s.FinalZ == 0.0 and s.SolutionVector == [0.0, 0.0]
    # a single variable, a single row
    s = PrimalSimplexSolver([3.0], [Constraint([2.0], "<=", 5.0)], True, engine=engine)
    s.Solve()
    assert s.Status == 0 and s.FinalZ == 7.5 and s.SolutionVector == [2.5]
    assert s.PivotLog.tolist() == [[1, 0]] and s.BasicVariables == [0]
    # minimisation keeps +c in the Z row (PrimalSimplexSolver.cs:62): nothing negative -> optimal
    s = PrimalSimplexSolver([3.0, 1.0], [Constraint([1.0, 1.0], "<=", 2.0)], False, engine=engine)
    s.Solve()
    assert s.Status == 0 and s.PivotLog.shape[0] == 0 and s.FinalZ == 0.0


@pytest.mark.parametrize("m,n,seed", [(6, 20000, 1), (3000, 10, 2), (1, 1, 3), (2, 5000, 4),
                                      (1500, 1, 5)])
def test_extreme_aspect_ratios(engine, oracle, m, n, seed):
    """Rows wider than one k_pivot_head trip (ld > 16384), very tall/thin tableaux, 1x1."""
    from lpr_381_group_v22_amd import Tableau
    T, basis = oracle.gen_dense_tableau(m, n, seed)
    st, piv, log = oracle.primal_solve(T, basis, 3000)
    tab = Tableau.synthetic(engine, m, n, seed)
    res = tab.solve(max_pivots=3000)
    assert res.status == st and res.pivots == piv
    assert tab.pivot_log().tolist() == log.tolist()
    assert tab.basis().tolist() == basis.tolist()
    assert tab.read().tobytes() == T.tobytes()
    tab.destroy()


@pytest.mark.parametrize("m,n,seed", [(40, 70, 1), (300, 500, 2), (513, 200, 3)])
def test_fused_and_pipelined_paths_agree(engine, oracle, m, n, seed):
    """Small tableaux take the single-launch k_pivot_fused path (ping-pong buffers); variant
    0x7fff forces the two-kernel pipelined path, 0x7ffe the fused one.  Same bits either way,
    including odd pivot counts (live tableau left in the second buffer) and resumed solves."""
    from lpr_381_group_v22_amd import Tableau
    T, basis = oracle.gen_dense_tableau(m, n, seed)
    st, piv, log = oracle.primal_solve(T, basis, 301)
    for variant in (0x7fff, 0x7ffe, 0):
        tab = Tableau.synthetic(engine, m, n, seed)
        r1 = tab.solve(max_pivots=7, variant=variant, batch=4)      # odd count, partial batches
        r2 = tab.solve(max_pivots=294, variant=variant)
        assert r1.pivots + r2.pivots == piv and r2.status == st
        assert tab.pivot_log().tolist() == log.tolist()
        assert tab.basis().tolist() == basis.tolist()
        assert tab.read().tobytes() == T.tobytes(), hex(variant)
        x, z = tab.extract_solution(n)
        xr, zr = oracle.extract_solution(T, n)
        assert x.tobytes() == xr.tobytes() and z == zr
        tab.destroy()


def test_handles_outlive_their_engine_safely():
    """lpr_engine_close releases what the caller forgot and orphans the handles: a later call on one
    of them is LPR_BAD_ARGUMENT (no crash, the exit-139 of round 1), and destroying it is safe."""
    import lpr_381_group_v22_amd as pkg
    from lpr_381_group_v22_amd import _native as N
    from lpr_381_group_v22_amd.engine import RevisedState, SensState
    eng = pkg.Engine(0)
    tab = pkg.Tableau.synthetic(eng, 24, 40, 1)
    assert tab.solve().status == 0
    big = pkg.Tableau.synthetic(eng, 700, 900, 2)      # K-pivot path: owns streams and scratch
    big.solve(max_pivots=40)
    rev = RevisedState.synthetic(eng, 16, 24, 0)
    rev.solve(max_pivots=3)
    sens = SensState.from_tableau(tab, 40)
    eng.close()
    for call in (lambda: tab.solve(), lambda: tab.read(), lambda: tab.basis(),
                 lambda: tab.pivot_log(), lambda: big.solve(max_pivots=3),
                 lambda: tab.extract_solution(40), lambda: tab.select_entering(),
                 lambda: rev.solve(max_pivots=1), lambda: rev.binv(),
                 lambda: sens.resolve_all()):
        with pytest.raises(N.EngineError) as ei:
            call()
        assert ei.value.status == N.LPR_BAD_ARGUMENT
    for h in (tab, big, rev, sens):
        h.destroy()
        h.destroy()  # idempotent on the Python side
    eng.close()


def test_abi_argument_errors_are_statuses_not_crashes(engine):
    """Bad arguments across the ABI come back as LPR_BAD_ARGUMENT with a message (never a crash,
    never an exception across the boundary)."""
    import ctypes as C
    import lpr_381_group_v22_amd as pkg
    from lpr_381_group_v22_amd import _native as N
    lib = N.lib
    BAD = N.LPR_BAD_ARGUMENT
    h = C.c_void_p()
    assert lib.lpr_engine_open(99, C.byref(h)) == BAD and b"out of range" in lib.lpr_last_error()
    assert lib.lpr_tableau_synthetic(engine._h, 0, 5, 1, C.byref(h)) == BAD
    assert lib.lpr_tableau_create(engine._h, 3, 4, None, None, C.byref(h)) == BAD
    tab = pkg.Tableau.synthetic(engine, 6, 9, 0)
    assert lib.lpr_primal_solve(tab._h, None, None) == BAD
    assert lib.lpr_select_leaving(tab._h, 10 ** 6, C.byref(C.c_int32())) == BAD
    assert lib.lpr_pivot(tab._h, 0, 0) == BAD                      # row 0 is the Z row
    assert lib.lpr_tableau_read_block(tab._h, 0, 99, 0, 1, None) == BAD
    assert lib.lpr_debug_head_stamps(tab._h, None, -1, None) == BAD
    assert lib.lpr_tableau_destroy(None) == BAD
    tab.destroy()
    assert lib.lpr_revised_step(None, None) == BAD
    assert lib.lpr_revised_create(engine._h, 0, 0, None, None, 0, None, 0, C.byref(h)) == BAD
    assert lib.lpr_bb_solve_level_sync(None, None, None, None, None) == BAD
    assert lib.lpr_comm_init(engine._h, 3, 2, (C.c_uint8 * 128)(), C.byref(h)) == BAD
    assert lib.lpr_comm_init_custom(0, 0, N.ALLREDUCE_MAX_FN(), N.ALLGATHER_FN(), None,
                                    C.byref(h)) == BAD
    assert lib.lpr_comm_destroy(None) == BAD and lib.lpr_comm_all_reduce_max(None, None, 1) == BAD


# a / b whose exact quotient lies 2e-16 ulp from the midpoint of two doubles: the device's own fp64
# division rounds it the other way (0.35 instead of 0.35000000000000003; tools/div_probe.hip)
HARD_A, HARD_B = float.fromhex("0x1.6666666666663p+0"), float.fromhex("0x1.ffffffffffffbp+1")


def _hard_division_tableau(rows=2, pad_cols=0):
    """Z row [-1, 0, ..., 0]; row 1 = [b, a, ..., 1]: the first pivot is (1, 0) and normalises a / b
    and 1 / b; further rows (factor 1 / 2) then carry the quotient on."""
    C = 3 + pad_cols
    T = np.zeros((rows, C))
    T[0, 0] = -1.0
    T[1, 0], T[1, 1], T[1, -1] = HARD_B, HARD_A, 1.0
    for i in range(2, rows):
        T[i, 0], T[i, 1], T[i, -1] = 0.5 * HARD_B, float(i), 10.0 * i
    return T


@pytest.mark.parametrize("rows,pad,variant,block", [(2, 0, 0, 0), (5, 3, 0, 0), (5, 3, 0x7fff, 1),
                                                    (40, 70, 0x2008, 0), (40, 70, 0x4008, 4),
                                                    (40, 70, 0x3008, 4), (40, 70, 0x6008, 4),
                                                    (40, 70, 0x5008, 4)],
                         ids=["2x3", "5x6", "5x6-graph", "small-heads", "seq", "ov2", "inplace",
                              "ov"])
def test_division_is_ieee_even_where_the_hardware_sequence_is_not(engine, oracle, rows, pad,
                                                                  variant, block):
    """Every `/` of the pivot loop (row normalisation :199, ratio test :180) goes through
    ieee_div: the quotient of the hard pair must be the host's, whatever path runs."""
    from lpr_381_group_v22_amd import Tableau
    assert HARD_A / HARD_B == 0.35000000000000003
    T0 = _hard_division_tableau(rows, pad)
    T = T0.copy()
    basis = np.arange(T.shape[1] - T.shape[0], T.shape[1] - 1, dtype=np.int32)
    st, piv, log = oracle.primal_solve(T, basis, 1)
    assert piv == 1 and T[1, 1] == 0.35000000000000003
    tab = Tableau.from_array(engine, T0)
    res = tab.solve(max_pivots=1, variant=variant, block=block)
    assert res.pivots == 1
    got = tab.read()
    assert got[1, 1] == 0.35000000000000003
    assert got.tobytes() == T.tobytes()
    tab.destroy()
