"""Seeded LP families used by the parity tests (TEST ONLY)."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from ref_py import PyConstraint, parse_model_text, program_option1_constraints

# data/TextFile.txt of the reference (3 lines of model data, also at TextFile/textfile.txt)
SAMPLE_MODEL = "max +2 +3 +3 +5 +2 +4\n+11 +8 +6 +14 +10 +10 <= 40\nbin bin bin bin bin bin\n"
# README.md:29-32 of the reference
README_MODEL = "max +2 +3 +4\n+1 +2 +3 <= 10\n+3 +2 +1 >= 15\n+ + +\n"

REL_CODE = {"<=": 0, ">=": 1, "=": 2}


def flatten(objective, constraints: List[PyConstraint]):
    """(objective, A[m x n], ncoef, rel, rhs) exactly as the host mirror hands them to the ABI."""
    n = len(objective)
    m = len(constraints)
    A = np.zeros((m, max(n, 1)))
    ncoef = np.zeros(m, dtype=np.int32)
    rel = np.zeros(m, dtype=np.int8)
    rhs = np.zeros(m)
    for i, c in enumerate(constraints):
        k = min(n, len(c.Coefficients))
        A[i, :k] = c.Coefficients[:k]
        ncoef[i] = k
        rel[i] = REL_CODE.get(c.Relation, 0)
        rhs[i] = c.RHS
    return np.asarray(objective, dtype=np.float64), A[:, :n], ncoef, rel, rhs


def sample_option1():
    _, obj, cons, _ = parse_model_text(SAMPLE_MODEL)
    return obj, program_option1_constraints(len(obj), cons), True


def readme_option1():
    _, obj, cons, _ = parse_model_text(README_MODEL)
    return obj, program_option1_constraints(len(obj), cons), True


def random_dense(m: int, n: int, seed: int):
    rng = np.random.RandomState(seed)
    A = rng.rand(m, n)
    b = 1.0 + rng.rand(m) * n / 4.0
    c = rng.rand(n)
    cons = [PyConstraint(A[i].tolist(), "<=", float(b[i])) for i in range(m)]
    return c.tolist(), cons, True


def tie_heavy(m: int, n: int, seed: int):
    """Small-integer coefficients: many exact ties in both arg-min scans, zero RHS rows
    (degenerate pivots), a few >= and = rows."""
    rng = np.random.RandomState(seed)
    A = rng.randint(0, 4, size=(m, n)).astype(float)
    b = rng.randint(0, 6, size=m).astype(float)
    c = rng.randint(1, 4, size=n).astype(float)
    rels = rng.choice(["<=", "<=", "<=", "=", ">="], size=m)
    cons = [PyConstraint(A[i].tolist(), str(rels[i]), float(b[i])) for i in range(m)]
    return c.tolist(), cons, True


def unbounded_lp():
    # x2 can grow without bound: its column has no positive entry
    obj = [1.0, 2.0]
    cons = [PyConstraint([1.0, -1.0], "<=", 4.0), PyConstraint([1.0, 0.0], "<=", 3.0)]
    return obj, cons, True


def min_lp():
    obj = [3.0, -2.0, 1.0]
    cons = [PyConstraint([1.0, 1.0, 1.0], "<=", 10.0), PyConstraint([1.0, -1.0, 0.0], "<=", 2.0),
            PyConstraint([0.0, 1.0, 2.0], "<=", 8.0)]
    return obj, cons, False


def klee_minty_bounded(d: int):
    """Klee-Minty-style cube with the growth base kept small (2) and a dimension cap so nothing
    leaves the fp64 range: max sum 2^(d-j) x_j, s.t. 2*sum_{j<i} 2^(i-j) x_j + x_i <= 5^i."""
    obj = [float(2 ** (d - 1 - j)) for j in range(d)]
    cons = []
    for i in range(d):
        row = [0.0] * d
        for j in range(i):
            row[j] = float(2 ** (i - j + 1))
        row[i] = 1.0
        cons.append(PyConstraint(row, "<=", float(5 ** (i + 1))))
    return obj, cons, True


def ragged_lp():
    """Coefficient lists shorter / longer than n (PrimalSimplexSolver.cs:68-72)."""
    obj = [2.0, 1.0, 3.0]
    cons = [PyConstraint([1.0, 1.0], "<=", 4.0),
            PyConstraint([1.0, 0.0, 2.0, 9.0, 9.0], "<=", 6.0),
            PyConstraint([], "<=", 1.0)]
    return obj, cons, True


def all_cases() -> List[Tuple[str, tuple]]:
    cases = [
        ("sample_option1", sample_option1()),
        ("readme_option1", readme_option1()),
        ("unbounded", unbounded_lp()),
        ("min_lp", min_lp()),
        ("ragged", ragged_lp()),
        ("klee_minty_6", klee_minty_bounded(6)),
        ("klee_minty_10", klee_minty_bounded(10)),
    ]
    for (m, n, seed) in [(4, 8, 0), (16, 32, 1), (16, 32, 2), (64, 128, 3), (40, 17, 4)]:
        cases.append((f"dense_{m}x{n}_s{seed}", random_dense(m, n, seed)))
    for (m, n, seed) in [(6, 6, 0), (12, 9, 1), (24, 30, 2), (33, 20, 3), (48, 64, 4)]:
        cases.append((f"ties_{m}x{n}_s{seed}", tie_heavy(m, n, seed)))
    return cases
