"""GPU parity tests of the revised primal simplex (lpr_revised_*) against the CPU oracle.
Bar: identical status / pivot log / basis, bit-exact B^-1, x_B, x and Z (all sums keep the C#'s
sequential order on the device); tolerance only for the MFMA product B^-1 * A."""
import struct

import numpy as np
import pytest

import lp_cases
from test_oracle_revised import flat, revised_cases

pytestmark = pytest.mark.gpu


def bits(x):
    return struct.pack(">d", float(x)).hex()


def to_constraints(cons):
    from lpr_381_group_v22_amd import Constraint
    return [Constraint(list(c.Coefficients), c.Relation, c.RHS) for c in cons]


@pytest.mark.parametrize("name,case", revised_cases(), ids=[c[0] for c in revised_cases()])
def test_solver_mirror_matches_oracle(engine, oracle, name, case):
    from lpr_381_group_v22_amd import RevisedPrimalSimplexSolver, SolverException
    obj, cons, is_min = case
    A, b = flat(cons)
    ref = oracle.revised_solve(obj, A, b, is_min, max_iter=3000)
    s = RevisedPrimalSimplexSolver(obj, to_constraints(cons), is_min, engine=engine)
    raised = None
    try:
        s.Solve(max_pivots=3000)
    except SolverException as ex:
        raised = ex
    assert s.Status == ref["status"]
    if ref["status"] in (1, 2, 3, 4):  # the C# throws (:91, :179, :183, :267)
        assert raised is not None and raised.status == ref["status"]
    else:
        assert raised is None
    assert s.PivotLog.tolist() == ref["log"].tolist()
    assert s.BasicVariables == ref["basis"].tolist()
    assert s.state.binv().tobytes() == ref["Binv"].tobytes(), "B^-1 not bit-identical"
    assert s.state.xb().tobytes() == ref["xB"].tobytes(), "x_B not bit-identical"
    if ref["status"] == 0:
        assert bits(s.FinalZ) == bits(ref["z"])
        assert [bits(v) for v in s.SolutionVector] == [bits(v) for v in ref["x"]]


def test_sample_model_option2(engine):
    """data/TextFile.txt through Program.cs option 2."""
    from lpr_381_group_v22_amd import RevisedPrimalSimplexSolver
    from test_oracle_revised import sample_option2
    obj, cons, is_min = sample_option2()
    s = RevisedPrimalSimplexSolver(obj, to_constraints(cons), is_min, engine=engine)
    s.Solve()
    assert bits(s.FinalZ) == bits(15.399999999999999)
    assert bits(s.SolutionVector[4]) == bits(0.19999999999999973)
    assert s.PivotLog.tolist() == [[4, 3, 10], [6, 5, 12], [2, 1, 8], [3, 2, 9], [0, 0, 6],
                                   [0, 4, 0]]


def test_constructor_argument_errors(engine):
    from lpr_381_group_v22_amd import Constraint, RevisedPrimalSimplexSolver
    with pytest.raises(ValueError, match="Objective cannot be null or empty"):
        RevisedPrimalSimplexSolver([], [Constraint([1.0], "<=", 1.0)], False, engine=engine)
    with pytest.raises(ValueError, match="Constraints cannot be null or empty"):
        RevisedPrimalSimplexSolver([1.0], [], False, engine=engine)
    with pytest.raises(ValueError, match="Constraint 2 has incorrect number of coefficients"):
        RevisedPrimalSimplexSolver([1.0, 2.0], [Constraint([1.0, 1.0], "<=", 1.0),
                                                Constraint([1.0], "<=", 1.0)], False,
                                   engine=engine)


@pytest.mark.parametrize("m,n,seed", [(64, 128, 0), (200, 333, 1), (257, 100, 2)])
def test_synthetic_lp_full_solve(engine, oracle, m, n, seed):
    from lpr_381_group_v22_amd import RevisedState
    c, A, b = oracle.gen_dense_lp(m, n, seed)
    ref = oracle.revised_solve(c, A, b, False, max_iter=20000)
    st = RevisedState.synthetic(engine, m, n, seed)
    res = st.solve(max_pivots=20000)
    assert res.status == ref["status"] == 0
    assert res.iterations == ref["iterations"]
    assert st.log().tolist() == ref["log"].tolist()
    assert st.basis().tolist() == ref["basis"].tolist()
    assert st.binv().tobytes() == ref["Binv"].tobytes()
    x, z = st.solution()
    assert x.tobytes() == ref["x"].tobytes() and bits(z) == bits(ref["z"]) == bits(res.z)
    st.destroy()


def _explicit(engine, oracle, c, A, b, max_iter):
    from lpr_381_group_v22_amd import RevisedState
    ref = oracle.revised_solve(c, A, b, False, max_iter=max_iter)
    st = RevisedState.create(engine, c, A, b, False)
    res = st.solve(max_pivots=max_iter)
    assert res.status == ref["status"] and res.iterations == ref["iterations"]
    assert st.log().tolist() == ref["log"].tolist()
    assert st.basis().tolist() == ref["basis"].tolist()
    assert st.binv().tobytes() == ref["Binv"].tobytes()
    assert st.xb().tobytes() == ref["xB"].tobytes()
    st.destroy()
    return ref


def test_entering_fold_with_thousands_of_prefix_records(engine, oracle):
    """Reduced costs that GROW with the index: every one of the n = 3000 candidates is a strict
    prefix record of the fold (more than the 1024 the one-wave replay keeps in LDS), so the
    block-wide next-take search runs; and the mirror image (costs that shrink: one record)."""
    rng = np.random.RandomState(5)
    m, n = 24, 3000
    A = rng.uniform(0.5, 2.0, size=(m, n))
    b = rng.uniform(50.0, 100.0, size=m)
    grow = 1.0 + 1e-3 * np.arange(n)
    ref = _explicit(engine, oracle, grow, A, b, 40)
    assert ref["log"][0][1] == n - 1  # the last (largest) reduced cost enters first
    _explicit(engine, oracle, grow[::-1].copy(), A, b, 40)
    # steps of 1e-12: a candidate is "better by more than EPS" only every ~1000 indices
    flat_c = 1.0 + 1e-12 * np.arange(n)
    ref = _explicit(engine, oracle, flat_c, A, b, 40)
    assert ref["log"][0][1] == 2002


def test_ratio_fold_beyond_the_lds_replay(engine, oracle):
    """m = 4300 rows (> 4096: the ratio test falls back to the block-wide next-take search) with
    many ties inside the EPS band (equal right-hand sides and repeated rows), few columns."""
    rng = np.random.RandomState(6)
    m, n = 4300, 24
    base = rng.uniform(0.5, 2.0, size=(43, n))
    A = np.repeat(base, 100, axis=0)[:m].copy()       # every row a hundred times: tied ratios
    A += rng.uniform(0.0, 2e-10, size=A.shape)         # ... inside the EPS band, not equal
    b = np.full(m, 10.0)
    c = rng.uniform(1.0, 2.0, size=n)
    ref = _explicit(engine, oracle, c, A, b, 6)
    assert ref["iterations"] >= 3


def test_resume_after_limit(engine, oracle):
    from lpr_381_group_v22_amd import RevisedState
    m, n, seed = 40, 80, 3
    c, A, b = oracle.gen_dense_lp(m, n, seed)
    ref = oracle.revised_solve(c, A, b, False)
    st = RevisedState.synthetic(engine, m, n, seed)
    total = 0
    while True:
        res = st.solve(max_pivots=5, batch=2)
        total += res.iterations
        if res.status != 5:
            break
        assert res.iterations == 5
    assert res.status == 0 and total == ref["iterations"]
    assert st.log().tolist() == ref["log"].tolist()
    assert st.binv().tobytes() == ref["Binv"].tobytes()
    st.destroy()


def test_update_binverse_zero_skip_and_negative_zero(engine, oracle):
    """UpdateBInverse goes through MultiplyMatrices' `|a_ik| < EPS -> continue` (:436): rows whose
    eta factor is below 1e-9 keep their values (with -0 -> +0), and 1/p below 1e-9 zeroes the
    pivot row.  Degenerate integer LPs hit the first rule; the tie-heavy family covers it."""
    from lpr_381_group_v22_amd import RevisedPrimalSimplexSolver
    for seed in range(5):
        obj, cons, _ = lp_cases.tie_heavy(20, 16, 100 + seed)
        A, b = flat(cons)
        b = np.abs(b)  # keep the slack basis feasible so that the solve runs
        cons = [type(c)(c.Coefficients, c.Relation, float(bb)) for c, bb in zip(cons, b)]
        ref = oracle.revised_solve(obj, A, b, False, max_iter=500)
        s = RevisedPrimalSimplexSolver(obj, to_constraints(cons), False, engine=engine)
        try:
            s.Solve(max_pivots=500)
        except Exception:
            pass
        assert s.Status == ref["status"]
        assert s.PivotLog.tolist() == ref["log"].tolist()
        assert s.state.binv().tobytes() == ref["Binv"].tobytes()


@pytest.mark.parametrize("m,n,seed,iters", [(96, 200, 0, 40), (130, 70, 1, 25), (300, 517, 2, 100)])
def test_binv_a_mfma_product(engine, oracle, m, n, seed, iters):
    """B^-1 * A (CaptureSnapshot :360) on the fp64 matrix cores vs the oracle's literal i-k-j loop
    with zero-skip.  Tolerance (stated): |err| <= 1e-9 * (|B^-1| |A|)_ij + 1e-12 -- the MFMA path
    accumulates with FMAs in a different association than the C#; the product only feeds the 3-dp
    printed tableau."""
    from lpr_381_group_v22_amd import RevisedState
    c, A, b = oracle.gen_dense_lp(m, n, seed)
    st = RevisedState.synthetic(engine, m, n, seed)
    res = st.solve(max_pivots=iters)
    Binv = st.binv()
    want = oracle.matmul_skip(Binv, A)
    got, ms = st.binv_a()
    assert ms > 0
    bound = 1e-9 * (np.abs(Binv) @ np.abs(A)) + 1e-12
    assert (np.abs(got - want) <= bound).all(), float(np.abs(got - want).max())
    # structure: the columns of basic structural variables are unit vectors of B^-1 A
    basis = st.basis()
    for row, v in enumerate(basis):
        if v < n:
            col = got[:, v]
            e = np.zeros(m)
            e[row] = 1.0
            assert np.abs(col - e).max() < 1e-7
    st.destroy()


def test_binv_a_zero_skip_is_applied(engine, oracle):
    """Entries of B^-1 below 1e-9 must not contribute (MultiplyMatrices :436).  Badly scaled
    coefficients leave such entries in B^-1 after a few pivots; with columns of A as large as 1e6
    a product that ignored the skip would miss the stated tolerance."""
    from lpr_381_group_v22_amd import RevisedState
    m, n = 32, 48
    rng = np.random.RandomState(0)
    A = 10.0 ** rng.uniform(-6, 6, size=(m, n))
    b = 10.0 ** rng.uniform(0, 3, size=m)
    c = rng.rand(n)
    st = RevisedState.create(engine, c, A, b, False)
    got, _ = st.binv_a()
    assert got.tobytes() == oracle.matmul_skip(np.eye(m), A).tobytes()  # I * A is exact
    ref = oracle.revised_solve(c, A, b, False, max_iter=12)
    st.solve(max_pivots=12)
    Binv = st.binv()
    assert Binv.tobytes() == ref["Binv"].tobytes()
    tiny = (np.abs(Binv) < 1e-9) & (Binv != 0.0)
    assert tiny.any(), "fixture no longer produces sub-EPS entries in B^-1"
    want = oracle.matmul_skip(Binv, A)
    got, _ = st.binv_a()
    Bz = np.where(np.abs(Binv) < 1e-9, 0.0, Binv)
    bound = 1e-9 * (np.abs(Bz) @ np.abs(A)) + 1e-12
    assert (np.abs(got - want) <= bound).all()
    st.destroy()


def test_config3_m4096_n8192_iterations_and_product(engine, oracle):
    """BASELINE configs[2] at full size (m=4096, n=8192): the first iterations against the oracle
    (about 3 s of one core each) -- status, pivot log, basis, x_B and all of B^-1 bit for bit -- then,
    after 150 more iterations on the device have filled B^-1 in, the MFMA product B^-1 * A
    (CaptureSnapshot :360) against the oracle's literal loop on a 64-row slice, with the stated
    tolerance |err| <= 1e-9 * (|B^-1| |A|)_ij + 1e-12; then two lpr_revised_step from that DENSE
    state against the oracle's loop body, every vector bit for bit."""
    from lpr_381_group_v22_amd import RevisedState
    m, n, seed, iters = 4096, 8192, 0, 3
    c, A, b = oracle.gen_dense_lp(m, n, seed)
    ref = oracle.revised_solve(c, A, b, False, max_iter=iters)
    st = RevisedState.synthetic(engine, m, n, seed)
    res = st.solve(max_pivots=iters)
    assert res.status == ref["status"] == 5 and res.iterations == ref["iterations"] == iters
    assert st.log().tolist() == ref["log"].tolist()
    assert st.basis().tolist() == ref["basis"].tolist()
    assert st.xb().tobytes() == ref["xB"].tobytes()
    assert st.binv().tobytes() == ref["Binv"].tobytes()
    del ref
    res = st.solve(max_pivots=150)
    assert res.iterations == 150
    Binv = st.binv()
    got, ms = st.binv_a()
    assert ms > 0 and got.shape == (m, n)
    rows = np.r_[0:16, 1000:1016, 2048:2064, 4080:4096]
    dense = (np.abs(Binv[rows]) >= 1e-9).sum(axis=1)
    assert dense.max() > 50, "B^-1 rows of the slice are still (almost) unit rows"
    want = oracle.matmul_skip(Binv[rows], A)
    bound = 1e-9 * (np.abs(Binv[rows]) @ np.abs(A)) + 1e-12
    err = np.abs(got[rows] - want)
    assert (err <= bound).all(), float(err.max())
    # columns of basic structural variables are unit vectors of B^-1 A (all rows)
    for row, v in enumerate(st.basis()):
        if v < n and row % 97 == 0:
            e = np.zeros(m)
            e[row] = 1.0
            assert np.abs(got[:, v] - e).max() < 1e-6
    del got, want
    # ---- the order-faithful sums with DENSE operands against the oracle.
    # From this filled-in state (B^-1, basis read back) the oracle runs the loop body of Solve()
    # (:89-215, orc_revised_iterate_from: ~2 s per pass) and lpr_revised_step must agree bit for
    # bit: entering, direction u = B^-1 a_e (:149-151), ratios (:154-176), leaving row, the
    # updated B^-1 (:264-275), and the post-pivot x_B / y / reduced costs (:218-227), which are
    # the next pass's x_B = B^-1 b (:89), y = c_B B^-1 (:93), rc (:96-102) of the oracle.
    c, A, b = oracle.gen_dense_lp(m, n, seed)
    nnz = int((np.abs(Binv) >= 1e-9).sum())
    assert nnz > 40 * m, "B^-1 has not filled in"
    basis0 = st.basis()
    for rep in range(2):
        o0 = oracle.revised_iterate_from(c, A, b, Binv, basis0, False)
        assert o0["status"] == 5
        info = st.step()
        y, rc, u, ratios, bpre, xb = st.snapshot()
        assert info.status == 5 and info.entering == o0["entering"], rep
        assert info.leaving_row == o0["leaving_row"], rep
        assert info.leaving_var == int(basis0[o0["leaving_row"]]), rep
        assert bpre.tolist() == basis0.tolist()
        assert u.tobytes() == o0["u"].tobytes(), rep
        assert ratios.tobytes() == o0["ratios"].tobytes(), rep
        e = o0["entering"]
        rc_pre = o0["rcX"][e] if e < n else o0["rcS"][e - n]
        assert bits(info.entering_rc_pre) == bits(rc_pre), rep
        Binv = st.binv()
        assert Binv.tobytes() == o0["Binv"].tobytes(), rep
        basis0 = st.basis()
        assert basis0.tolist() == o0["basis"].tolist(), rep
        o1 = oracle.revised_iterate_from(c, A, b, Binv, basis0, False)  # pre-values of the next pass
        assert xb.tobytes() == o1["xB"].tobytes(), rep
        assert y.tobytes() == o1["y"].tobytes(), rep
        assert rc[:n].tobytes() == o1["rcX"].tobytes() and rc[n:].tobytes() == o1["rcS"].tobytes()
    st.destroy()


SNAP_KEYS = ("y", "u_pre", "ratios_pre", "xB")


@pytest.mark.parametrize("name,case", revised_cases()[:9], ids=[c[0] for c in revised_cases()[:9]])
def test_iteration_snapshots_match_oracle(engine, oracle, name, case):
    """IterationSnapshots (RevisedPrimalSimplexSolver.cs:36, CaptureSnapshot :294-387): one
    lpr_revised_step per iteration; every number the C# prints -- post-pivot y / reduced costs /
    x_B, pre-pivot direction, ratios and basis, the entering reduced cost, Z_working, Z_original,
    B^-1 A in the C#'s own summation order, B^-1 -- is the oracle's bit for bit, and the text of
    every block equals the test-side restatement of the C#'s StringBuilder code."""
    from lpr_381_group_v22_amd import RevisedPrimalSimplexSolver, SolverException
    from lpr_381_group_v22_amd.engine import RevisedState
    from ref_py import PyRevised, py_snapshot_text
    obj, cons, is_min = case
    A, b = flat(cons)
    tr = oracle.revised_trace(obj, A, b, is_min, max_iter=60, cap=64)
    n, m = len(obj), len(cons)
    st = RevisedState.create(engine, obj, A, b, is_min)
    got = 0
    while got < 60:
        info = st.step()
        if info.status not in (0, 5):
            break
        a = tr["snapshots"][got]
        y, rc, u, ratios, bpre, xb = st.snapshot()
        assert info.entering == a["entering"], got
        assert bits(info.z_working) == bits(a["z_working"]), got
        assert bits(info.z_original) == bits(a["z_original"]), got
        assert y.tobytes() == a["y"].tobytes() and xb.tobytes() == a["xB"].tobytes(), got
        assert rc[:n].tobytes() == a["rcX"].tobytes() and rc[n:].tobytes() == a["rcS"].tobytes()
        assert st.basis().tolist() == a["basis_post"].tolist()
        assert st.binv_a_exact().tobytes() == a["BInvA"].tobytes(), got
        assert st.binv().tobytes() == a["BInv"].tobytes(), got
        if info.status == 5:
            assert (info.leaving_row, info.leaving_var) == (a["leaving_row"], a["leaving_var"])
            assert bits(info.entering_rc_pre) == bits(a["rc_pre"])
            assert u.tobytes() == a["u_pre"].tobytes(), got
            assert ratios.tobytes() == a["ratios_pre"].tobytes(), got
            assert bpre.tolist() == a["basis_pre"].tolist()
        got += 1
        if info.status == 0:
            break
    assert got == min(tr["count"], 61 if tr["status"] == 0 else 60)
    st.destroy()
    # the mirror class builds the same text as the restatement of CaptureSnapshot
    p = PyRevised(obj, cons, is_min)
    p.solve(max_iter=60, capture=True)
    s = RevisedPrimalSimplexSolver(obj, to_constraints(cons), is_min, engine=engine,
                                   snapshots="all")
    try:
        s.Solve(max_pivots=60)
    except SolverException:
        pass
    want = [py_snapshot_text(q, n, m, is_min) for q in p.snapshots]
    assert len(s.IterationSnapshots) == len(want)
    for k, (g, w) in enumerate(zip(s.IterationSnapshots, want)):
        assert g == w, (name, k)


def test_step_and_batched_solve_mix(engine, oracle):
    """lpr_revised_step and lpr_revised_solve on one handle: same state as the oracle after any
    interleaving."""
    from lpr_381_group_v22_amd.engine import RevisedState
    m, n, seed = 40, 64, 5
    c, A, b = oracle.gen_dense_lp(m, n, seed)
    ref = oracle.revised_solve(c, A, b, False)
    st = RevisedState.synthetic(engine, m, n, seed)
    k = 0
    while True:
        info = st.step()
        if info.status != 5:
            break
        k += 1
        res = st.solve(max_pivots=3)
        k += res.iterations
        if res.status != 5:
            break
    assert k == ref["iterations"]
    assert st.log().tolist() == ref["log"].tolist()
    assert st.binv().tobytes() == ref["Binv"].tobytes()
    x, z = st.solution()
    assert x.tobytes() == ref["x"].tobytes() and bits(z) == bits(ref["z"])
    st.destroy()


@pytest.mark.parametrize("m,n,iters", [(64, 15000, 25), (4200, 300, 4), (17, 33, 0), (100, 31, 0),
                                       (5, 3, 0)],
                         ids=["wide-beyond-the-ring", "tall-beyond-lds-ratio", "ragged", "n<32", "tiny"])
def test_fused_iteration_outside_its_fast_paths(engine, oracle, m, n, iters):
    """The three-launch iteration of lpr_revised_solve (csrc/revised_fused.hip) where its tails
    leave their fast paths: n + m above what the entering fold can stage in the ring's LDS (and
    above the fold's register budget: the block-wide next-take search), m above the 4 096 rows the
    ratio replay keeps in LDS, shapes that are not multiples of the 16-row / 32-column strips.
    Status, log, basis, B^-1 and x_B bits against the oracle."""
    from lpr_381_group_v22_amd import RevisedState
    c, A, b = oracle.gen_dense_lp(m, n, 5)
    ref = oracle.revised_solve(c, A, b, False, max_iter=iters)
    st = RevisedState.synthetic(engine, m, n, 5)
    res = st.solve(max_pivots=iters)
    assert res.status == ref["status"] and res.iterations == ref["iterations"]
    assert st.log().tolist() == ref["log"].tolist()
    assert st.basis().tolist() == ref["basis"].tolist()
    assert st.binv().tobytes() == ref["Binv"].tobytes()
    if ref["status"] in (0, 5):
        assert st.xb().tobytes() == ref["xB"].tobytes()
    st.destroy()
