"""Writes tests/golden/bb_golden.json from runs where the C oracle and the independent Python
restatement of the Branch & Bound path agree (the reference itself cannot be run here).

    python tests/golden/make_golden_bb.py
"""
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import bb_cases  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from ref_py_bb import BranchAndBound  # noqa: E402


def bits(x):
    return struct.pack(">d", float(x)).hex()


def main():
    orc = Oracle()
    out = {}
    for name, (obj, cons) in bb_cases.all_bb_cases():
        st, T, n = bb_cases.primal_final_tableau(orc, obj, cons)
        for cap in (20, 60):
            r = orc.bb_solve(T, n, node_cap=cap)
            bb = BranchAndBound(n, node_cap=cap)
            p = bb.Execute([list(map(float, row)) for row in T.tolist()])
            assert r["records"] == bb.records and r["trace"] == bb.trace, name
            assert bits(r["z"]) == bits(p["z"]), name
            out[f"{name}@{cap}"] = dict(
                status=r["status"], processed=r["processed"], pop_order=r["pop_order"],
                z_bits=bits(r["z"]),
                x_bits=[bits(v) for v in r["x"]] if r["found"] else None,
                records=[[rec["parent"], rec["kind"], rec["var"], rec["status"], bits(rec["z"])]
                         for rec in r["records"]],
                trace=[list(t) for t in r["trace"]])
    with open(os.path.join(HERE, "bb_golden.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
