#!/usr/bin/env python3
"""Time-bounded randomised comparison of the revised simplex on the device with the C oracle
(status, iteration count, log, basis, bits of B^-1 and x_B): python tools/fuzz_revised_gpu.py [s] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import lp_cases  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from test_oracle_revised import flat  # noqa: E402
import lpr_381_group_v22_amd as pkg  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
oracle = Oracle()
eng = pkg.Engine(0)
t_end = time.time() + budget
cases = 0
statuses = {}
while time.time() < t_end:
    rng = np.random.RandomState(seed)
    shape = rng.randint(0, 4)
    if shape == 0:
        m, n = int(rng.randint(2, 40)), int(rng.randint(2, 60))
    elif shape == 1:
        m, n = int(rng.randint(20, 300)), int(rng.randint(20, 500))
    elif shape == 2:
        m, n = int(rng.randint(4, 30)), int(rng.randint(1100, 2600))   # > 1024 fold candidates
    else:
        m, n = int(rng.randint(300, 700)), int(rng.randint(5, 60))
    kind = rng.randint(0, 3)
    gen = lp_cases.random_dense if kind != 1 else lp_cases.tie_heavy
    obj, cons, is_max = gen(m, n, int(rng.randint(0, 100000)))
    if kind == 2:
        cons = [type(c)([-v if (k + j) % 3 == 0 else v for j, v in enumerate(c.Coefficients)],
                        c.Relation, c.RHS) for k, c in enumerate(cons)]
    A, b = flat(cons)
    cap = int(rng.choice([3, 17, 60, 400]))
    ref = oracle.revised_solve(obj, A, b, not is_max, max_iter=cap)
    st = pkg.RevisedState.create(eng, obj, A, b, not is_max)
    res = st.solve(max_pivots=cap, batch=int(rng.choice([0, 1, 5])))
    tag = (seed, m, n, cap)
    assert res.status == ref["status"] and res.iterations == ref["iterations"], (tag, res.status, ref["status"])
    assert st.log().tolist() == ref["log"].tolist(), tag
    assert st.basis().tolist() == ref["basis"].tolist(), tag
    assert st.binv().tobytes() == ref["Binv"].tobytes(), tag
    assert st.xb().tobytes() == ref["xB"].tobytes(), tag
    statuses[res.status] = statuses.get(res.status, 0) + 1
    st.destroy()
    cases += 1
    seed += 1
    if cases % 50 == 0:
        print(f"{cases} cases, seed {seed}, statuses {statuses}", flush=True)
print(f"OK: {cases} cases, seeds {seed0}..{seed - 1}, statuses {statuses}: all identical to the oracle")
eng.close()
