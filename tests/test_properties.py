"""Property tests (hypothesis): random small LPs -- integer data, so exact ties and degenerate
vertices are common -- through every implementation of the pivot path.
  CPU : C oracle == independent Python restatement (primal and revised), bit for bit.
  GPU : HIP engine == C oracle, bit for bit (marked gpu)."""
import struct

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import lp_cases
from ref_py import PyConstraint, PyPrimal, PyRevised

STATUS = {0: "optimal", 1: "unbounded", 2: "infeasible_basis", 3: "pivot_too_small",
          4: "entering_already_basic", 5: "limit"}


def bits(x):
    return struct.pack(">d", float(x)).hex()


@st.composite
def small_lp(draw, max_m=7, max_n=7, allow_ge=True):
    m = draw(st.integers(1, max_m))
    n = draw(st.integers(1, max_n))
    coef = st.integers(-3, 6)
    A = [[float(draw(coef)) for _ in range(n)] for _ in range(m)]
    b = [float(draw(st.integers(0, 12))) for _ in range(m)]
    c = [float(draw(st.integers(-2, 7))) for _ in range(n)]
    rels = [draw(st.sampled_from(["<=", "<=", "<=", "=", ">="] if allow_ge else ["<="]))
            for _ in range(m)]
    return c, [PyConstraint(A[i], rels[i], b[i]) for i in range(m)]


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(lp=small_lp())
def test_primal_oracle_equals_python(oracle, lp):
    obj, cons = lp
    o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
    T, basis = oracle.primal_build(o, A, rel, rhs, True, ncoef)
    st_, piv, log = oracle.primal_solve(T, basis, 200)
    p = PyPrimal(obj, cons, True)
    ps = p.solve(200)
    assert ps == STATUS[st_]
    assert p.log == [tuple(v) for v in log.tolist()]
    assert np.array(p.t).tobytes() == T.tobytes()


@settings(max_examples=100, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(lp=small_lp(allow_ge=False))
def test_revised_oracle_equals_python(oracle, lp):
    obj, cons = lp
    A = np.array([c.Coefficients for c in cons], dtype=np.float64)
    b = np.array([c.RHS for c in cons], dtype=np.float64)
    r = oracle.revised_solve(obj, A, b, False, max_iter=200)
    p = PyRevised(obj, cons, False)
    ps = p.solve(200)
    assert ps == STATUS[r["status"]]
    assert p.log == [tuple(v) for v in r["log"].tolist()]
    assert np.array(p.Binv).tobytes() == r["Binv"].tobytes()
    if ps == "optimal":
        assert bits(p.FinalZ) == bits(r["z"])


@pytest.mark.gpu
@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(lp=small_lp(max_m=9, max_n=9))
def test_primal_gpu_equals_oracle(engine, oracle, lp):
    from lpr_381_group_v22_amd import Tableau
    obj, cons = lp
    o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
    T, basis = oracle.primal_build(o, A, rel, rhs, True, ncoef)
    tab = Tableau.from_lp(engine, o, A, rel, rhs, True, ncoef)
    assert tab.read().tobytes() == T.tobytes()
    st_, piv, log = oracle.primal_solve(T, basis, 200)
    res = tab.solve(max_pivots=200)
    assert res.status == st_ and res.pivots == piv
    assert tab.pivot_log().tolist() == log.tolist()
    assert tab.read().tobytes() == T.tobytes()
    tab.destroy()


@pytest.mark.gpu
@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(lp=small_lp(max_m=8, max_n=8, allow_ge=False))
def test_revised_gpu_equals_oracle(engine, oracle, lp):
    from lpr_381_group_v22_amd import RevisedState
    obj, cons = lp
    A = np.array([c.Coefficients for c in cons], dtype=np.float64)
    b = np.array([c.RHS for c in cons], dtype=np.float64)
    r = oracle.revised_solve(obj, A, b, False, max_iter=200)
    s = RevisedState.create(engine, obj, A, b, False)
    res = s.solve(max_pivots=200)
    assert res.status == r["status"]
    assert s.log().tolist() == r["log"].tolist()
    assert s.binv().tobytes() == r["Binv"].tobytes()
    if r["status"] == 0:
        assert bits(res.z) == bits(r["z"])
    s.destroy()
