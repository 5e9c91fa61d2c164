"""CPU tests: the C oracle of the sensitivity re-solve (SensitivityAnalyzer.cs) against the
independent Python restatement, edit by edit.  PARITY UNPINNED by the reference (no tests)."""
import numpy as np

import ref_py_sens as rp
import sens_cases


def test_edit_scripts(oracle):
    codes = set()
    for name, (T, x, z, basis), ops in sens_cases.scripts(oracle):
        o = oracle.sens(T, x, z, basis)
        p = rp.PySens(T.tolist(), list(map(float, x)), float(z), [int(b) for b in basis])
        st = o.state()
        assert np.array(p.t).tobytes() == st["T"].tobytes() and p.basic == st["basic"], name
        for k, (op, args) in enumerate(ops):
            op, args = sens_cases.materialize(op, args, o.state()["T"], k)
            rc = getattr(o, op)(*args)
            prc = rp.run(getattr(p, op), *args)
            prc = 0 if prc is None else prc
            assert prc == rc, (name, k, op, rc, prc)
            st = o.state()
            assert np.array(p.t).tobytes() == st["T"].tobytes(), (name, k, op)
            assert p.basic == st["basic"], (name, k, op)
            assert np.array(p.sol).tobytes() == st["sol"].tobytes(), (name, k, op)
            assert p.z == st["z"], (name, k, op)
            assert p.log == o.log(), (name, k, op)
            codes.add(rc)
    assert {0, -1, 8} <= codes and (1 in codes or 2 in codes), codes


def test_resolve_keeps_an_optimal_tableau_unchanged(oracle):
    T, x, z, basis = sens_cases.solved_lp(oracle, 8, 12, 1)
    o = oracle.sens(T, x, z, basis)
    before = o.state()
    assert o.resolve_all() == 0
    after = o.state()
    assert after["T"].tobytes() == before["T"].tobytes()
    assert abs(after["z"] - z) == 0.0
