"""CPU tests: the C oracle of PrimalSimplexSolver against (a) the independent Python restatement,
(b) the hand traces of SURVEY.md section 4, (c) scipy on well-posed LPs, (d) committed goldens.

PARITY UNPINNED by the reference (it ships no tests / golden outputs); these are the pins."""
import json
import os
import struct

import numpy as np
import pytest

import lp_cases
from ref_py import PyPrimal

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "primal_golden.json")
STATUS = {0: "optimal", 1: "unbounded", 5: "limit"}


def bits(x: float) -> str:
    return struct.pack(">d", float(x)).hex()


def run_oracle(oracle, obj, cons, is_max, max_pivots=0):
    o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
    T, basis = oracle.primal_build(o, A, rel, rhs, is_max, ncoef)
    T0 = T.copy()
    st, piv, log = oracle.primal_solve(T, basis, max_pivots)
    x, z = oracle.extract_solution(T, len(obj))
    return dict(T0=T0, T=T, basis=basis, status=st, pivots=piv, log=log, x=x, z=z)


@pytest.mark.parametrize("name,case", lp_cases.all_cases(), ids=[c[0] for c in lp_cases.all_cases()])
def test_oracle_equals_python_restatement(oracle, name, case):
    obj, cons, is_max = case
    r = run_oracle(oracle, obj, cons, is_max, max_pivots=5000)
    p = PyPrimal(obj, cons, is_max)
    assert np.array_equal(np.array(p.t), r["T0"]), "constructor differs"
    ps = p.solve(max_pivots=5000)
    assert ps == STATUS[r["status"]]
    assert [tuple(v) for v in r["log"].tolist()] == p.log
    assert r["basis"].tolist() == p.basic
    # bit-exact tableau (NaN-free cases; compare raw bytes so that -0 vs +0 would also show)
    assert np.array(p.t).tobytes() == r["T"].tobytes()
    if ps == "optimal":
        assert bits(p.FinalZ) == bits(r["z"])
        assert [bits(v) for v in p.SolutionVector] == [bits(v) for v in r["x"]]


def test_survey_hand_trace_sample(oracle):
    """data/TextFile.txt through Program.cs option 1."""
    obj, cons, is_max = lp_cases.sample_option1()
    r = run_oracle(oracle, obj, cons, is_max)
    assert r["status"] == 0
    assert bits(r["z"]) == bits(15.4)
    assert r["log"].tolist() == [[5, 3], [7, 5], [3, 1], [4, 2], [1, 0], [1, 4]]
    assert r["basis"].tolist() == [4, 7, 1, 2, 3, 11, 5]
    C = r["T"].shape[1]
    np.testing.assert_allclose(r["T"][1:, C - 1], [0.2, 1, 1, 1, 1, 0.8, 1], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r["x"], [0, 1, 1, 1, 0.2, 1], rtol=0, atol=1e-12)


def test_survey_hand_trace_readme(oracle):
    """the >= row is negated and never repaired, so the reference
    returns an infeasible 'optimum' (Z = 9, RHS -9).  A faithful restatement reproduces it."""
    obj, cons, is_max = lp_cases.readme_option1()
    r = run_oracle(oracle, obj, cons, is_max)
    assert r["status"] == 0
    assert r["z"] == 9.0
    C = r["T"].shape[1]
    assert r["T"][1:, C - 1].tolist() == [4.0, -9.0, 1.0, 1.0, 1.0]


def test_unbounded_and_limit(oracle):
    obj, cons, is_max = lp_cases.unbounded_lp()
    r = run_oracle(oracle, obj, cons, is_max)
    assert r["status"] == 1
    obj, cons, is_max = lp_cases.random_dense(16, 32, 1)
    r = run_oracle(oracle, obj, cons, is_max, max_pivots=3)
    assert r["status"] == 5 and r["pivots"] == 3 and len(r["log"]) == 3


def test_objective_against_scipy(oracle):
    """Sanity only: on well-posed LPs (<=, b >= 0) the reference's algorithm is a correct simplex,
    so its optimum must agree with HiGHS to 1e-9 relative."""
    from scipy.optimize import linprog
    for (m, n, seed) in [(4, 8, 0), (16, 32, 1), (64, 128, 3), (40, 17, 4)]:
        obj, cons, is_max = lp_cases.random_dense(m, n, seed)
        r = run_oracle(oracle, obj, cons, is_max)
        assert r["status"] == 0
        A = np.array([c.Coefficients for c in cons])
        b = np.array([c.RHS for c in cons])
        ref = linprog(-np.array(obj), A_ub=A, b_ub=b, bounds=(0, None), method="highs")
        assert ref.status == 0
        assert abs(r["z"] - (-ref.fun)) <= 1e-9 * max(1.0, abs(ref.fun))


def test_generator_spec(oracle):
    """The synthetic-LP generator: SplitMix64 keyed by (seed, stream, i, j)."""
    M = (1 << 64) - 1

    def sm(x):
        x = (x + 0x9E3779B97F4A7C15) & M
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    def u01(seed, stream, i, j):
        k = sm(seed ^ ((stream * 0xD1B54A32D192ED03) & M))
        k = sm((k + i) & M)
        k = sm((k + j) & M)
        return (k >> 11) * 2.0 ** -53

    for args in [(0, 0, 0, 0), (7, 1, 3, 0), (2, 2, 0, 11), (123456789, 0, 4095, 8191)]:
        assert oracle.u01(*args) == u01(*args)
    c, A, b = oracle.gen_dense_lp(5, 8, 7)
    assert A[3, 2] == u01(7, 0, 3, 2) and c[6] == u01(7, 2, 0, 6)
    assert b[4] == (8 * 0.25) * (1.0 + u01(7, 1, 4, 0) * 0.1)
    T, basis = oracle.gen_dense_tableau(5, 8, 7)
    T2, basis2 = oracle.primal_build(c, A, np.zeros(5, dtype=np.int8), b, True)
    assert T.tobytes() == T2.tobytes() and basis.tolist() == basis2.tolist()


def test_golden_fixture(oracle):
    """tests/golden/primal_golden.json was written by tests/golden/make_golden.py from runs in
    which oracle == Python restatement; the oracle must keep reproducing it bit for bit."""
    with open(GOLDEN) as f:
        gold = json.load(f)
    cases = dict(lp_cases.all_cases())
    assert set(gold) <= set(cases)
    for name, g in gold.items():
        obj, cons, is_max = cases[name]
        r = run_oracle(oracle, obj, cons, is_max, max_pivots=5000)
        assert STATUS[r["status"]] == g["status"], name
        assert r["log"].tolist() == g["log"], name
        assert r["basis"].tolist() == g["basis"], name
        assert bits(r["T"][0, -1]) == g["z_bits"], name
        assert [bits(v) for v in r["x"]] == g["x_bits"], name
        import hashlib
        assert hashlib.sha256(r["T"].tobytes()).hexdigest() == g["tableau_sha256"], name
