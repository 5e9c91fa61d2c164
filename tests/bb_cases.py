"""Branch & Bound instances for the parity tests (TEST ONLY).  Every instance goes through the
reference's own route (Program.cs option 3): model -> x_i <= 1 rows appended -> PrimalSimplexSolver
-> FinalTableau -> BranchAndBoundAdapter.SolveFromPrimal."""
from __future__ import annotations

import numpy as np

import lp_cases
from ref_py import PyConstraint, parse_model_text, program_option1_constraints


def knapsack_sample():
    _, obj, cons, _ = parse_model_text(lp_cases.SAMPLE_MODEL)
    return obj, program_option1_constraints(len(obj), cons)


def random_binary_program(n: int, mcons: int, seed: int):
    """max c x, A x <= b, x binary (after the x_i <= 1 rows): small positive integers so that the
    LP relaxation is fractional and the 4-decimal rounding of the reference is exercised."""
    rng = np.random.RandomState(seed)
    c = rng.randint(1, 20, size=n).astype(float)
    A = rng.randint(1, 15, size=(mcons, n)).astype(float)
    b = np.floor(A.sum(axis=1) * rng.uniform(0.3, 0.6, size=mcons))
    cons = [PyConstraint(A[i].tolist(), "<=", float(b[i])) for i in range(mcons)]
    return c.tolist(), program_option1_constraints(n, cons)


def fractional_program(n: int, mcons: int, seed: int):
    """Non-integer data: exercises Math.Round(x, 4) on values that are not short decimals."""
    rng = np.random.RandomState(seed)
    c = np.round(rng.uniform(1, 9, size=n), 3)
    A = np.round(rng.uniform(0.5, 7, size=(mcons, n)), 3)
    b = np.round(A.sum(axis=1) * rng.uniform(0.35, 0.65, size=mcons), 2)
    cons = [PyConstraint(A[i].tolist(), "<=", float(b[i])) for i in range(mcons)]
    return c.tolist(), program_option1_constraints(n, cons)


def huge_relaxation_value():
    """An instance (found by tools/fuzz_bb_gpu.py, seed 2955) where a child's relaxation puts
    1.1e15 into a decision variable: `(int)Math.Floor(v)` (:870-871) is then out of int's range and
    the bound of BOTH children becomes int.MinValue (x64 cvttsd2si)."""
    rng = np.random.RandomState(2955)
    n, mc = int(rng.randint(3, 14)), int(rng.randint(1, 6))
    gen = random_binary_program if rng.randint(0, 2) else fractional_program
    return gen(n, mc, int(rng.randint(0, 1 << 30)))


def all_bb_cases():
    cases = [("knapsack_sample", knapsack_sample())]
    for (n, mc, seed) in [(4, 1, 0), (5, 2, 1), (6, 2, 2), (8, 3, 3), (10, 2, 4), (7, 4, 5)]:
        cases.append((f"binary_{n}v{mc}c_s{seed}", random_binary_program(n, mc, seed)))
    for (n, mc, seed) in [(4, 2, 10), (6, 3, 11), (9, 2, 12)]:
        cases.append((f"frac_{n}v{mc}c_s{seed}", fractional_program(n, mc, seed)))
    cases.append(("huge_relaxation_value", huge_relaxation_value()))
    return cases


def primal_final_tableau(oracle, obj, cons):
    """Oracle PrimalSimplexSolver on the instance; returns (status, FinalTableau, n)."""
    o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
    T, basis = oracle.primal_build(o, A, rel, rhs, True, ncoef)
    st, piv, log = oracle.primal_solve(T, basis)
    return st, T, len(obj)
