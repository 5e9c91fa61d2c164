"""Writes tests/golden/primal_golden.json: results of the seeded LP families on which the C oracle
and the independent Python restatement agree bit for bit.  (The reference itself cannot be run:
C#, no toolchain; it also ships no golden outputs -- see oracle/lpr_oracle.h.)

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402

import lp_cases  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from ref_py import PyPrimal  # noqa: E402

STATUS = {0: "optimal", 1: "unbounded", 5: "limit"}


def bits(x):
    return struct.pack(">d", float(x)).hex()


def main():
    orc = Oracle()
    out = {}
    for name, (obj, cons, is_max) in lp_cases.all_cases():
        o, A, ncoef, rel, rhs = lp_cases.flatten(obj, cons)
        T, basis = orc.primal_build(o, A, rel, rhs, is_max, ncoef)
        st, piv, log = orc.primal_solve(T, basis, 5000)
        x, z = orc.extract_solution(T, len(obj))
        p = PyPrimal(obj, cons, is_max)
        ps = p.solve(5000)
        assert ps == STATUS[st] and p.log == [tuple(v) for v in log.tolist()], name
        assert np.array(p.t).tobytes() == T.tobytes(), name
        out[name] = dict(status=ps, pivots=int(piv), log=log.tolist(), basis=basis.tolist(),
                         z_bits=bits(T[0, -1]), x_bits=[bits(v) for v in x],
                         tableau_sha256=hashlib.sha256(T.tobytes()).hexdigest())
    with open(os.path.join(HERE, "primal_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
