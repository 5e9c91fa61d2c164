#!/usr/bin/env python3
"""Time-bounded randomised comparison of the K-pivot paths with the C oracle (status, pivot log,
basis, every byte of the tableau) over more seeds and shapes than the seeded tests hold:
    python tools/fuzz_gpu.py [seconds] [first seed]
Shapes: small dense / tie-heavy / partly negated LPs, wide ones (many head workgroups), tall ones
(many rows per head lane), mid-size synthetic tableaux through the DEFAULT path."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import lp_cases  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
import lpr_381_group_v22_amd as pkg  # noqa: E402
from test_block_gpu import _build, SEQ, OV, OV2, INPLACE  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
oracle = Oracle()
eng = pkg.Engine(0)
t_end = time.time() + budget
cases = legs = 0
seed = seed0
while time.time() < t_end:
    rng = np.random.RandomState(seed)
    shape = rng.randint(0, 5)
    if shape == 4:  # synthetic mid-size tableau, default path, a few dozen pivots
        m, n = int(rng.choice([300, 513, 700, 1100])), int(rng.choice([400, 1024, 1900]))
        T, basis = oracle.gen_dense_tableau(m, n, seed)
        tab = pkg.Tableau.synthetic(eng, m, n, seed)
    else:
        if shape == 0:
            m, n = int(rng.randint(2, 60)), int(rng.randint(2, 80))
        elif shape == 1:
            m, n = int(rng.randint(4, 60)), int(rng.randint(1500, 6000))
        elif shape == 2:
            m, n = int(rng.randint(600, 3000)), int(rng.randint(3, 40))
        else:
            m, n = int(rng.randint(100, 400)), int(rng.randint(100, 600))
        kind = rng.randint(0, 3)
        gen = lp_cases.random_dense if kind != 1 else lp_cases.tie_heavy
        obj, cons, is_max = gen(m, n, int(rng.randint(0, 100000)))
        if kind == 2:
            cons = [type(c)([-v if (k + j) % 3 == 0 else v for j, v in enumerate(c.Coefficients)],
                            c.Relation, c.RHS) for k, c in enumerate(cons)]
        T, basis = _build(oracle, (obj, cons, is_max))
        tab = pkg.Tableau.from_array(eng, T, basis)
    total = 0
    for leg in range(5):
        limit = int(rng.choice([1, 3, 16, 17, 33, 50, 0]))
        variant = int(rng.choice([0, 0, SEQ, OV, OV2, INPLACE, 0x2000]))  # 0x2000: small_kernels.hip
        block = 0 if variant in (0, 0x2000) else int(rng.randint(2, (8 if variant == INPLACE else 16) + 1))
        cap = limit if limit else (400 if shape == 4 else 100000)
        st, piv, log = oracle.primal_solve(T, basis, cap)
        res = tab.solve(max_pivots=cap, block=block, variant=variant)
        total += piv
        tag = (seed, shape, m, n, leg, hex(variant), block, cap)
        assert res.status == st and res.pivots == piv and res.total_pivots == total, (tag, res.status, st)
        assert tab.pivot_log().tolist()[total - piv:] == log.tolist(), tag
        assert tab.basis().tolist() == basis.tolist(), tag
        assert tab.read().tobytes() == T.tobytes(), tag
        legs += 1
        if st != 5:
            break
    tab.destroy()
    cases += 1
    seed += 1
    if cases % 25 == 0:
        print(f"{cases} cases, {legs} legs, seed {seed}", flush=True)
print(f"OK: {cases} cases, {legs} legs, seeds {seed0}..{seed - 1}, all identical to the oracle")
eng.close()
