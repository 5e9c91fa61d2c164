"""GPU parity tests of the K-pivots-per-sweep paths (opts.block): deciding K pivots ahead from
O(R + C) data and applying them in one sweep must give the same status, pivot log, basis and
tableau BITS as the oracle's one-pivot-at-a-time loop, for every K, every stop reason and every
way a pivot limit can cut a block.  Two implementations: csrc/overlap_kernels.hip (default on
large tableaux: out-of-place sweep with the next block's loop heads inside the same launch,
variant 0x50tr; the same as two concurrent kernels on two streams, variant 0x30tr; and variant 0x40tr: all loop heads of a block in one persistent
launch, then the sweep in place) and csrc/block_kernels.hip (in place, one launch per loop head,
variant 0x60tr)."""
import hashlib

import numpy as np
import pytest

import lp_cases

pytestmark = pytest.mark.gpu

SEQ, OV, INPLACE, OV2 = 0x4008, 0x5008, 0x6008, 0x3008
# bits 16..18 of the variant (K-pivot paths): the loop heads are normally confined to one XCD and
# hand off through its L2 when the launch finds them there; SPREAD = one workgroup per group
# anywhere on the chip with memory-side hand-offs (the round-1 form), MEMSIDE = confined but
# memory-side hand-offs, STAMPS = the diagnostic build.  Same bits every way.
STAMPS, SPREAD, MEMSIDE = 0x10000, 0x20000, 0x40000
NOAVOID, NOHINT, DEVHAND = 0x80000, 0x100000, 0x200000  # sweep / hand-over variants (ABI header)
# SWEEPDEV = the sweep of a step asks the heads' completion word as it starts instead of waiting
# for an event on its stream (opt-in, like DEVHAND for the other direction)
SWEEPDEV = 0x400000
NAMES = {SEQ: "seq", OV: "ov", INPLACE: "inplace", OV2: "ov2",
         SEQ | SPREAD: "seq-spread", OV2 | SPREAD: "ov2-spread", OV2 | MEMSIDE: "ov2-memside",
         SEQ | STAMPS: "seq-stamps", OV2 | STAMPS: "ov2-stamps",
         OV2 | DEVHAND: "ov2-devhand", OV2 | NOAVOID: "ov2-noavoid", OV2 | NOHINT: "ov2-nohint",
         OV2 | DEVHAND | SWEEPDEV: "ov2-dev2", OV2 | SWEEPDEV: "ov2-sweepdev",
         0x3024: "ov2-2x4", 0x3028: "ov2-2x8", 0x3004: "ov2-4", 0x3010: "ov2-16",
         0x4024: "seq-2x4", 0x4028: "seq-2x8", 0x4010: "seq-16"}
# (variant, block)
BLOCKS = [(SEQ, 2), (SEQ, 5), (SEQ, 8), (SEQ, 16), (OV, 2), (OV, 3), (OV, 8), (OV, 16),
          (OV2, 2), (OV2, 7), (OV2, 16),
          (INPLACE, 2), (INPLACE, 4), (INPLACE, 8),
          (SEQ | SPREAD, 16), (OV2 | SPREAD, 16), (OV2 | MEMSIDE, 9), (SEQ | STAMPS, 16),
          (OV2 | STAMPS, 13), (OV2 | DEVHAND, 16), (OV2 | DEVHAND, 3), (OV2 | NOAVOID, 16), (OV2 | NOHINT, 6),
          (OV2 | DEVHAND | SWEEPDEV, 16), (OV2 | SWEEPDEV, 5),
          (0x3024, 16), (0x3028, 16), (0x3004, 16), (0x3010, 16),
          (0x4024, 16), (0x4028, 11), (0x4010, 16)]
IDS = [NAMES[v] + str(b) for v, b in BLOCKS]


def _build(oracle, case):
    obj, cons, is_max = case
    o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
    return oracle.primal_build(o, A, rel, rhs, is_max, ncoef)


@pytest.mark.parametrize("variant,block", BLOCKS, ids=IDS)
def test_named_cases_match_oracle(engine, oracle, variant, block):
    from lpr_381_group_v22_amd import Tableau
    statuses = set()
    for name, case in lp_cases.all_cases():
        T, basis = _build(oracle, case)
        if T.shape[0] < 2:
            continue
        tab = Tableau.from_array(engine, T, basis)
        st, piv, log = oracle.primal_solve(T, basis, 5000)
        res = tab.solve(max_pivots=5000, block=block, variant=variant)
        assert res.block == block, name
        assert res.status == st and res.pivots == piv, (name, res.status, st, res.pivots, piv)
        assert tab.pivot_log().tolist() == log.tolist(), name
        assert tab.basis().tolist() == basis.tolist(), name
        assert tab.read().tobytes() == T.tobytes(), name
        statuses.add(st)
        tab.destroy()
    assert {0, 1} <= statuses


@pytest.mark.parametrize("variant,block", BLOCKS, ids=IDS)
def test_every_pivot_limit_cuts_the_block_correctly(engine, oracle, variant, block):
    """max_pivots = 0..17 on the same LP: the limit lands on every position inside a block."""
    from lpr_381_group_v22_amd import Tableau
    m, n, seed = 40, 64, 3
    T0, b0 = oracle.gen_dense_tableau(m, n, seed)
    for limit in list(range(1, 20)) + [31, 32, 33, 35]:
        T, basis = T0.copy(), b0.copy()
        st, piv, log = oracle.primal_solve(T, basis, limit)
        tab = Tableau.synthetic(engine, m, n, seed)
        res = tab.solve(max_pivots=limit, block=block, batch=5, variant=variant)
        assert res.status == st and res.pivots == piv, (limit, res.status, st, res.pivots, piv)
        assert tab.pivot_log().tolist() == log.tolist(), limit
        assert tab.read().tobytes() == T.tobytes(), limit
        tab.destroy()


@pytest.mark.parametrize("variant,block", BLOCKS, ids=IDS)
def test_resume_after_pivot_limit(engine, oracle, variant, block):
    from lpr_381_group_v22_amd import Tableau
    m, n, seed = 48, 96, 2
    T, basis = oracle.gen_dense_tableau(m, n, seed)
    st, piv, log = oracle.primal_solve(T, basis)
    tab = Tableau.synthetic(engine, m, n, seed)
    total = 0
    while True:
        res = tab.solve(max_pivots=7, batch=3, block=block, variant=variant)
        total += res.pivots
        assert res.total_pivots == total
        if res.status != 5:
            break
        assert res.pivots == 7
    assert res.status == st and total == piv
    assert tab.pivot_log().tolist() == log.tolist()
    assert tab.basis().tolist() == basis.tolist()
    assert tab.read().tobytes() == T.tobytes()
    tab.destroy()


@pytest.mark.parametrize("variant,block,timed", [(SEQ, 3, False), (SEQ, 16, True), (OV2, 16, False), (OV2, 6, True), (OV, 2, False), (OV, 4, True), (OV, 16, False),
                                                 (OV, 11, True), (INPLACE, 8, False),
                                                 (INPLACE, 5, True)])
def test_medium_dense_lp_full_solve(engine, oracle, variant, block, timed):
    """m=200, n=333 to optimality (hundreds of pivots), graph replay and eager+events."""
    from lpr_381_group_v22_amd import Tableau
    m, n, seed = 200, 333, 11
    T, basis = oracle.gen_dense_tableau(m, n, seed)
    st, piv, log = oracle.primal_solve(T, basis, 20000)
    tab = Tableau.synthetic(engine, m, n, seed)
    res = tab.solve(max_pivots=20000, block=block, time_kernels=timed, variant=variant)
    assert res.status == st == 0 and res.pivots == piv
    assert tab.pivot_log(1 << 16).tolist() == log.tolist()
    assert tab.basis().tolist() == basis.tolist()
    assert tab.read().tobytes() == T.tobytes()
    if timed:
        launches, total_ms, avg_ms = tab.kernel_stats()
        assert launches > 0 and avg_ms > 0
    tab.destroy()


@pytest.mark.parametrize("base,tr", [(0x4000, 4), (0x4000, 8), (0x4000, 16), (0x6000, 8), (0x6000, 16), (0x6000, 32), (0x5000, 4),
                                     (0x5000, 8), (0x5000, 16)])
def test_sweep_tile_shapes_give_identical_bits(engine, oracle, base, tr):
    from lpr_381_group_v22_amd import Tableau
    m, n, seed = 150, 260, 5
    T, basis = oracle.gen_dense_tableau(m, n, seed)
    st, piv, log = oracle.primal_solve(T, basis, 90)
    tab = Tableau.synthetic(engine, m, n, seed)
    res = tab.solve(max_pivots=90, block=4, variant=base | tr)
    assert res.status == st and res.pivots == piv and res.block == 4
    assert tab.pivot_log().tolist() == log.tolist()
    assert tab.read().tobytes() == T.tobytes()
    tab.destroy()


@pytest.mark.parametrize("variant,block", [(SEQ, 8), (SEQ, 16), (OV, 8), (OV, 16), (OV2, 16), (INPLACE, 8)])
def test_repeated_rows_and_columns_inside_a_block(engine, oracle, variant, block):
    """Degenerate / tie-heavy LPs: the same row leaves twice within one block, a column re-enters,
    ties in both arg-mins -- the chains through earlier pivots must reproduce them exactly."""
    from lpr_381_group_v22_amd import Tableau
    seen_repeat = seen_col_repeat = False
    cases = [lp_cases.tie_heavy(18, 14, seed) for seed in range(6)] + \
        [lp_cases.tie_heavy(8, 12, seed) for seed in (2, 10)] + \
        [lp_cases.klee_minty_bounded(d) for d in (4, 6, 8)]
    for seed, case in enumerate(cases):
        T, basis = _build(oracle, case)
        tab = Tableau.from_array(engine, T, basis)
        st, piv, log = oracle.primal_solve(T, basis, 400)
        rows = [r for r, _ in log.tolist()]
        cols = [c for _, c in log.tolist()]
        for k in range(0, len(rows), 8):
            seen_repeat |= len(set(rows[k:k + 8])) < len(rows[k:k + 8])
            seen_col_repeat |= len(set(cols[k:k + 8])) < len(cols[k:k + 8])
        res = tab.solve(max_pivots=400, block=block, variant=variant)
        assert res.status == st and res.pivots == piv, seed
        assert tab.pivot_log().tolist() == log.tolist(), seed
        assert tab.read().tobytes() == T.tobytes(), seed
        tab.destroy()
    assert seen_repeat and seen_col_repeat, "fixtures no longer repeat a pivot row / column inside a block"


@pytest.mark.parametrize("variant", [SEQ, OV, OV2, INPLACE])
def test_wide_and_tall_shapes(engine, oracle, variant):
    """ld wider than one head trip (G * 256 double2 < ld / 2) and tall thin tableaux."""
    from lpr_381_group_v22_amd import Tableau
    # (3, 40000): more column pairs than head lanes (64 groups x 256): the lanes' "further" loops
    for (m, n, seed) in [(6, 20000, 1), (3000, 10, 2), (2, 5000, 4), (1, 1, 3), (3, 40000, 6)]:
        T, basis = oracle.gen_dense_tableau(m, n, seed)
        st, piv, log = oracle.primal_solve(T, basis, 600)
        tab = Tableau.synthetic(engine, m, n, seed)
        res = tab.solve(max_pivots=600, block=4, variant=variant)
        assert res.status == st and res.pivots == piv, (m, n)
        assert tab.pivot_log().tolist() == log.tolist(), (m, n)
        assert tab.basis().tolist() == basis.tolist(), (m, n)
        assert tab.read().tobytes() == T.tobytes(), (m, n)
        tab.destroy()


def test_north_star_size_blocked_equals_one_pivot_path(engine, oracle):
    """m=4096, n=8192: the first 8 pivots against the oracle, then 96 more against the
    one-pivot-per-sweep path (same device, different kernels): same log, same tableau hash."""
    from lpr_381_group_v22_amd import Tableau
    m, n, seed = 4096, 8192, 0
    T, basis = oracle.gen_dense_tableau(m, n, seed)
    st, piv, log = oracle.primal_solve(T, basis, 8)
    a = Tableau.synthetic(engine, m, n, seed)
    res = a.solve(max_pivots=8, block=8, variant=INPLACE)
    assert res.status == st == 5 and res.pivots == 8 and res.block == 8
    assert a.pivot_log().tolist() == log.tolist()
    assert hashlib.sha256(a.read().tobytes()).hexdigest() == \
        hashlib.sha256(T.tobytes()).hexdigest()
    del T
    b = Tableau.synthetic(engine, m, n, seed)
    b.solve(max_pivots=8, block=1)
    ra = a.solve(max_pivots=96)            # auto: fused heads + in-place sweep
    rb = b.solve(max_pivots=96, block=1)
    assert ra.block > 1 and rb.block == 1
    assert ra.pivots == rb.pivots == 96 and ra.z == rb.z
    assert a.pivot_log().tolist() == b.pivot_log().tolist()
    assert a.basis().tolist() == b.basis().tolist()
    ha = hashlib.sha256(a.read().tobytes()).hexdigest()
    hb = hashlib.sha256(b.read().tobytes()).hexdigest()
    assert ha == hb
    a.destroy()
    b.destroy()


def test_north_star_full_solve_default_equals_one_pivot_path(engine):
    """m=4096, n=8192 solved to optimality (about 122 000 pivots) twice: the default path (16 pivots
    per sweep, heads beside the sweep on a second stream) and one pivot per sweep.  Same status,
    pivot count, pivot log, basis, objective bits and tableau hash."""
    from lpr_381_group_v22_amd import Tableau
    m, n, seed = 4096, 8192, 0
    a = Tableau.synthetic(engine, m, n, seed)
    ra = a.solve()
    b = Tableau.synthetic(engine, m, n, seed)
    rb = b.solve(block=1)
    assert ra.block == 16 and rb.block == 1
    assert ra.status == rb.status == 0
    assert ra.pivots == rb.pivots > 100000
    assert ra.z == rb.z
    la, lb = a.pivot_log(cap=ra.pivots + 8), b.pivot_log(cap=rb.pivots + 8)
    assert la.shape == lb.shape and (la == lb).all()
    assert a.basis().tolist() == b.basis().tolist()
    ha = hashlib.sha256(a.read().tobytes()).hexdigest()
    hb = hashlib.sha256(b.read().tobytes()).hexdigest()
    assert ha == hb
    a.destroy()
    b.destroy()


def test_random_small_lps_random_path_block_and_limits(engine, oracle):
    """Seeded fuzz: 60 small LPs (dense / tie-heavy, some unbounded), each solved in several legs
    with a random path, block size, batch and pivot limit per leg; every leg must leave the oracle's
    state (status, pivots, log, basis, tableau bytes)."""
    from lpr_381_group_v22_amd import Tableau
    rng = np.random.RandomState(2024)
    paths = [SEQ, OV, OV2, INPLACE]
    seen = set()
    for case_no in range(60):
        m, n = int(rng.randint(2, 40)), int(rng.randint(2, 48))
        kind = rng.randint(0, 3)
        if kind == 0:
            case = lp_cases.random_dense(m, n, int(rng.randint(0, 1000)))
        elif kind == 1:
            case = lp_cases.tie_heavy(m, n, int(rng.randint(0, 1000)))
        else:  # negated constraint rows: unbounded directions appear
            obj, cons, is_max = lp_cases.random_dense(m, n, int(rng.randint(0, 1000)))
            cons = [type(c)([-v if (k + j) % 3 == 0 else v for j, v in enumerate(c.Coefficients)],
                            c.Relation, c.RHS) for k, c in enumerate(cons)]
            case = (obj, cons, is_max)
        T, basis = _build(oracle, case)
        tab = Tableau.from_array(engine, T, basis)
        total = 0
        for leg in range(6):
            limit = int(rng.choice([1, 2, 3, 5, 9, 17, 33, 0]))
            variant = int(rng.choice(paths))
            kmax = 8 if variant == INPLACE else 16
            block = int(rng.randint(2, kmax + 1))
            batch = int(rng.choice([0, 1, 3, 7, 40]))
            st, piv, log = oracle.primal_solve(T, basis, limit if limit else 100000)
            res = tab.solve(max_pivots=limit if limit else 100000, block=block, variant=variant,
                            batch=batch)
            total += piv
            tag = (case_no, leg, hex(variant), block, limit, batch)
            assert res.status == st and res.pivots == piv and res.total_pivots == total, tag
            assert tab.pivot_log().tolist()[total - piv:] == log.tolist(), tag
            assert tab.basis().tolist() == basis.tolist(), tag
            assert tab.read().tobytes() == T.tobytes(), tag
            seen.add(st)
            if st != 5:
                break
        tab.destroy()
    assert {0, 1, 5} <= seen


def test_random_wide_lps_many_head_groups(engine, oracle):
    """The same fuzz on wide LPs (1 500 - 5 000 columns, few rows): 6 - 20 head workgroups exchange
    their partials; cheap for the oracle."""
    from lpr_381_group_v22_amd import Tableau
    rng = np.random.RandomState(77)
    for case_no in range(10):
        m, n = int(rng.randint(8, 70)), int(rng.randint(1500, 5000))
        gen = lp_cases.random_dense if case_no % 2 == 0 else lp_cases.tie_heavy
        T, basis = _build(oracle, gen(m, n, int(rng.randint(0, 1000))))
        tab = Tableau.from_array(engine, T, basis)
        total = 0
        for leg in range(4):
            limit = int(rng.choice([7, 16, 31, 50, 0]))
            variant = int(rng.choice([SEQ, OV, OV2, INPLACE]))
            block = int(rng.randint(2, (8 if variant == INPLACE else 16) + 1))
            st, piv, log = oracle.primal_solve(T, basis, limit if limit else 100000)
            res = tab.solve(max_pivots=limit if limit else 100000, block=block, variant=variant)
            total += piv
            tag = (case_no, leg, hex(variant), block, limit, m, n)
            assert res.status == st and res.pivots == piv and res.total_pivots == total, tag
            assert tab.pivot_log().tolist()[total - piv:] == log.tolist(), tag
            assert tab.read().tobytes() == T.tobytes(), tag
            if st != 5:
                break
        tab.destroy()


def _north_star_oracle(oracle, pivots):
    m, n, seed = 4096, 8192, 0
    T, basis = oracle.gen_dense_tableau(m, n, seed)
    st, piv, log = oracle.primal_solve(T, basis, pivots)
    assert st == 5 and piv == pivots
    return st, log, basis, hashlib.sha256(T.tobytes()).hexdigest()


@pytest.fixture(scope="module")
def north_star_64(oracle):
    return _north_star_oracle(oracle, 64)


@pytest.mark.parametrize("variant", [0, SPREAD, MEMSIDE, DEVHAND, NOAVOID, 0x4008, 0x5008, 0x3024,
                                     0x3028, 0x3004, DEVHAND | SWEEPDEV],
                         ids=["default", "default-spread", "default-memside", "default-devhand",
                              "default-noavoid", "seq", "ov", "ov2-2x4", "ov2-2x8", "ov2-4",
                              "default-dev2"])
def test_north_star_size_64_pivots_vs_oracle(engine, north_star_64, variant):
    """BASELINE's headline size (m=4096, n=8192: 4097 x 12289, 25 loop-head workgroups): four full
    blocks of 16 -- from the second on, the next block's heads run beside the sweep -- against the
    ORACLE: status, pivot log, basis and the sha256 of all 402.8 MB of the tableau.  The default
    is the two-stream path with the heads confined to one XCD."""
    from lpr_381_group_v22_amd import Tableau
    st, log, basis, sha = north_star_64
    tab = Tableau.synthetic(engine, 4096, 8192, 0)
    res = tab.solve(max_pivots=64, variant=variant)
    assert res.status == st and res.pivots == 64 and res.block == 16
    assert tab.pivot_log().tolist() == log.tolist()
    assert tab.basis().tolist() == basis.tolist()
    assert hashlib.sha256(tab.read().tobytes()).hexdigest() == sha
    tab.destroy()


@pytest.mark.parametrize("variant", [DEVHAND | SWEEPDEV, SWEEPDEV], ids=["dev2", "sweepdev"])
def test_north_star_size_sweep_asks_the_heads_word_vs_oracle(engine, north_star_64, variant):
    """The sweep-side device hand-over only starts once a poll has told the host which XCD the
    loop heads share and from the third step of a call on: 16 pivots first (one poll), then 48 in
    a second call, whose last two sweeps follow their predecessors without an event and ask the
    heads' completion word (B.sflag[1]) themselves.  Same oracle data as the test above."""
    from lpr_381_group_v22_amd import Tableau
    st, log, basis, sha = north_star_64
    tab = Tableau.synthetic(engine, 4096, 8192, 0)
    r1 = tab.solve(max_pivots=16, variant=variant)
    assert r1.pivots == 16 and r1.block == 16
    res = tab.solve(max_pivots=48, variant=variant)
    assert res.status == st and res.pivots == 48 and res.total_pivots == 64
    assert tab.pivot_log().tolist() == log.tolist()
    assert tab.basis().tolist() == basis.tolist()
    assert hashlib.sha256(tab.read().tobytes()).hexdigest() == sha
    tab.destroy()


@pytest.mark.parametrize("m,n", [(2048, 4096), (1792, 3584)], ids=["101MB", "77MB"])
def test_mid_size_default_path_vs_oracle(engine, oracle, m, n):
    """Either side of the size where the default changes from heads-then-sweep to the two-stream
    overlap (80 MB): 80 pivots in two legs through the DEFAULT path against the oracle -- status,
    pivot log, basis, every byte of the tableau."""
    from lpr_381_group_v22_amd import Tableau
    T, basis = oracle.gen_dense_tableau(m, n, 3)
    st, piv, log = oracle.primal_solve(T, basis, 80)
    tab = Tableau.synthetic(engine, m, n, 3)
    total = 0
    for leg in (33, 47):
        res = tab.solve(max_pivots=leg)
        total += res.pivots
        assert res.block == 16
    assert total == piv and res.status == st
    assert tab.pivot_log().tolist() == log.tolist()
    assert tab.basis().tolist() == basis.tolist()
    assert tab.read().tobytes() == T.tobytes()
    tab.destroy()


def test_event_sampling_stride(engine, oracle):
    """opts.time_kernels = 4 brackets one step in four of the two-stream path: a call of 10 full
    blocks is 11 steps (the first only decides); steps 4 and 8 are sampled, the step windows run
    to the next sampled step / the closing event.  Same bits as an untimed solve."""
    from lpr_381_group_v22_amd import Tableau
    m, n = 2048, 4096
    T, basis = oracle.gen_dense_tableau(m, n, 5)
    st, piv, log = oracle.primal_solve(T, basis, 160)
    tab = Tableau.synthetic(engine, m, n, 5)
    res = tab.solve(max_pivots=160, time_kernels=4)
    assert res.status == st and res.pivots == piv == 160 and res.block == 16
    launches, total_ms, avg_ms = tab.kernel_stats()
    steps, step_ms = tab.step_stats()
    assert launches == 2 and steps == 7 and 0 < avg_ms and total_ms < step_ms
    assert tab.pivot_log().tolist() == log.tolist()
    assert tab.read().tobytes() == T.tobytes()
    tab.destroy()


def test_north_star_size_ragged_legs_vs_oracle(engine, oracle):
    """The same size in legs that cut blocks (1 + 37 + 32 pivots, the bench's probe / warm-up /
    timed pattern) with kernel timing on: the tableau after 70 pivots equals the oracle's."""
    from lpr_381_group_v22_amd import Tableau
    st, log, basis, sha = _north_star_oracle(oracle, 70)
    tab = Tableau.synthetic(engine, 4096, 8192, 0)
    total = 0
    for leg, timed in ((1, False), (37, False), (32, True)):
        res = tab.solve(max_pivots=leg, time_kernels=timed)
        total += res.pivots
        assert res.status == 5 and res.pivots == leg and res.total_pivots == total
    launches, total_ms, avg_ms = tab.kernel_stats()
    steps, step_ms = tab.step_stats()
    assert launches == 2 and steps == 2 and 0 < total_ms <= step_ms
    assert tab.pivot_log().tolist() == log.tolist()
    assert tab.basis().tolist() == basis.tolist()
    assert hashlib.sha256(tab.read().tobytes()).hexdigest() == sha
    tab.destroy()


def test_head_placement_report(engine):
    """Diagnostic build of the loop heads (variant bit 16): stamps are monotone per pivot and the
    launch reports where the lead workgroup ran and which hand-off form it chose."""
    from lpr_381_group_v22_amd import Tableau
    tab = Tableau.synthetic(engine, 4096, 8192, 0)
    res = tab.solve(max_pivots=48, variant=STAMPS)
    assert res.pivots == 48
    stamps, xcc, l2 = tab.head_stamps()
    assert 0 <= xcc < 8 and l2 in (0, 1)
    used = stamps[:48]
    assert (used[:, 0] > 0).all()
    assert (np.diff(used.astype(np.int64), axis=1) >= 0).all()
    tab.destroy()


def test_interleaved_paths_on_one_handle(engine, oracle):
    """One tableau handle through the one-pivot graph path, the fused small-tableau path (odd pivot
    counts swap its two buffers), the two-stream and in-place K-pivot paths and back: a captured
    graph must never be replayed on buffers it was not captured for."""
    from lpr_381_group_v22_amd import Tableau
    m, n, seed = 60, 90, 4
    T, basis = oracle.gen_dense_tableau(m, n, seed)
    tab = Tableau.synthetic(engine, m, n, seed)
    total = 0
    legs = [(0x7fff, 1, 6), (0x7ffe, 1, 5), (0x7fff, 1, 6), (OV2, 4, 7), (0x7fff, 1, 6),
            (0x7ffe, 1, 3), (INPLACE, 3, 5), (0x7fff, 1, 6), (SEQ, 16, 9), (0, 0, 4),
            (0x7fff, 1, 6)]
    for variant, block, limit in legs:
        st, piv, log = oracle.primal_solve(T, basis, limit)
        res = tab.solve(max_pivots=limit, variant=variant, block=block, batch=6)
        total += piv
        tag = (hex(variant), block, limit)
        assert res.status == st and res.pivots == piv and res.total_pivots == total, tag
        assert tab.pivot_log().tolist()[total - piv:] == log.tolist(), tag
        assert tab.basis().tolist() == basis.tolist(), tag
        assert tab.read().tobytes() == T.tobytes(), tag
        if st != 5:
            break
    tab.destroy()


def test_ratio_that_overflows_is_not_a_candidate(engine, oracle):
    """FindLeavingVariable (:184) takes a row only when ratio < minRatio, which starts at
    double.MaxValue: rhs / a = +inf (or exactly MaxValue) never qualifies -> "unbounded"."""
    from lpr_381_group_v22_amd import Tableau
    T = np.array([[-1.0, 0.0, 0.0, 0.0],
                  [1e-8, 1.0, 0.0, 1e308],         # a > 1e-9; 1e308 / 1e-8 = +inf
                  [2.0 ** -29, 0.0, 1.0, 2.0 ** 994 * (2.0 - 2.0 ** -52)]], dtype=np.float64)
    # second row: ratio == DBL_MAX exactly ((2 - 2^-52) * 2^1023): not < MaxValue either
    assert T[2, 3] / T[2, 0] == np.finfo(np.float64).max and T[1, 3] / T[1, 0] == np.inf
    for variant, block in ((0x7fff, 1), (0x7ffe, 1), (SEQ, 4), (OV2, 4), (OV, 4), (INPLACE, 4)):
        Tc, basis = T.copy(), np.array([1, 2], dtype=np.int32)
        st, piv, log = oracle.primal_solve(Tc, basis, 10)
        assert st == 1 and piv == 0
        tab = Tableau.from_array(engine, T, [1, 2])
        res = tab.solve(max_pivots=10, variant=variant, block=block)
        assert res.status == 1 and res.pivots == 0, hex(variant)
        assert tab.read().tobytes() == Tc.tobytes()
        tab.destroy()


@pytest.mark.parametrize("m,n,variant", [(2048, 3072, 0), (2048, 3072, 0x4008), (640, 1024, 0),
                                         (640, 1024, 0x6008)],
                         ids=["two-stream-84MB", "seq-84MB", "small-default", "blk"])
def test_pivot_log_growth_is_crossed_in_an_oracle_checked_solve(engine, oracle, m, n, variant):
    """the pivot log starts at 65 536 pairs, so its growth (the path
    whose race was fixed in 681e713: the grown capacity must reach the heads' stream in order) was
    only crossed by the 122 k-pivot device-vs-device test.  With the test hook LPR_TEST_LOG_CAP the
    log starts at 32 pairs and grows four times (32 -> 512) inside 300 pivots -- on the two-stream
    path (84 MB tableau, above the engine's 80 MB switch) in ONE call, so the growth happens
    between queued steps -- and everything is compared with the ORACLE: status, the whole pivot
    log, basis, the tableau's bytes."""
    import os
    from lpr_381_group_v22_amd import Tableau
    pivots = 300
    T, basis = oracle.gen_dense_tableau(m, n, 3)
    st, piv, log = oracle.primal_solve(T, basis, pivots)
    assert st == 5 and piv == pivots
    os.environ["LPR_TEST_LOG_CAP"] = "32"
    try:
        tab = Tableau.synthetic(engine, m, n, 3)
    finally:
        del os.environ["LPR_TEST_LOG_CAP"]
    res = tab.solve(max_pivots=pivots, variant=variant)
    assert res.status == st and res.pivots == pivots
    assert tab.pivot_log().tolist() == log.tolist(), "pivot log differs after the log has grown"
    assert tab.basis().tolist() == basis.tolist()
    assert hashlib.sha256(tab.read().tobytes()).hexdigest() == hashlib.sha256(T.tobytes()).hexdigest()
    # and in ragged legs (growth between calls as well as inside one)
    os.environ["LPR_TEST_LOG_CAP"] = "16"
    try:
        tab2 = Tableau.synthetic(engine, m, n, 3)
    finally:
        del os.environ["LPR_TEST_LOG_CAP"]
    done = 0
    for leg in (5, 40, 3, 100, 152):
        r = tab2.solve(max_pivots=leg, variant=variant)
        done += r.pivots
    assert done == pivots and tab2.pivot_log().tolist() == log.tolist()
    assert tab2.read().tobytes() == tab.read().tobytes()
    tab.destroy()
    tab2.destroy()


SMALL = 0x2000  # csrc/small_kernels.hip forced (the default wherever it fits: R <= 1024, ld <= 2048)


@pytest.mark.parametrize("m,n,pivots", [(1023, 1000, 90), (1000, 1000, 120), (700, 1300, 200),
                                        (37, 11, 0), (1, 1, 0), (3, 900, 0), (600, 2, 0)],
                         ids=["max-rows-1024", "16MB", "wide", "tiny", "1x1", "flat", "tall"])
def test_small_tableau_path_at_its_limits_vs_oracle(engine, oracle, m, n, pivots):
    """The one-workgroup loop heads of small_kernels.hip (R <= 1024, ld <= 2048) forced beyond the
    size the engine picks them for by itself, at the row limit (R = 1024), on flat / tall / 1 x 1
    shapes and in ragged legs: status, pivot log, basis and tableau bytes against the oracle."""
    from lpr_381_group_v22_amd import Tableau
    T, basis = oracle.gen_dense_tableau(m, n, 11)
    st, piv, log = oracle.primal_solve(T, basis, pivots if pivots else 100000)
    tab = Tableau.synthetic(engine, m, n, 11)
    res = tab.solve(max_pivots=pivots if pivots else 100000, variant=SMALL)
    assert res.block == 16, "the small-tableau path did not run"
    assert res.status == st and res.pivots == piv
    assert tab.pivot_log().tolist() == log.tolist()
    assert tab.basis().tolist() == basis.tolist()
    assert tab.read().tobytes() == T.tobytes()
    tab.destroy()
    if piv >= 40:  # the same pivots in ragged legs (limits inside a block, resume)
        tab = Tableau.synthetic(engine, m, n, 11)
        done = 0
        for leg in (1, 15, 16, 17, 3):
            r = tab.solve(max_pivots=leg, variant=SMALL)
            assert r.status == 5 and r.pivots == leg
            done += leg
        T2, b2 = oracle.gen_dense_tableau(m, n, 11)
        oracle.primal_solve(T2, b2, done)
        assert tab.read().tobytes() == T2.tobytes() and tab.basis().tolist() == b2.tolist()
        tab.destroy()


def test_small_tableau_path_unbounded_and_degenerate(engine, oracle):
    """Unbounded exit and tie-saturated ratio tests on the small path."""
    import lp_cases
    from lpr_381_group_v22_amd import Tableau
    for name, (obj, cons, is_max) in [("unbounded", lp_cases.unbounded_lp()),
                                      ("ties", lp_cases.tie_heavy(33, 20, 3)),
                                      ("km", lp_cases.klee_minty_bounded(9))]:
        o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
        T, basis = oracle.primal_build(o, A, rel, rhs, is_max, ncoef)
        T0, b0 = T.copy(), basis.copy()
        st, piv, log = oracle.primal_solve(T, basis, 100000)
        tab = Tableau.from_array(engine, T0, b0)
        res = tab.solve(variant=SMALL)
        assert (res.status, res.pivots) == (st, piv), name
        assert tab.pivot_log().tolist() == log.tolist(), name
        assert tab.read().tobytes() == T.tobytes(), name
        tab.destroy()
