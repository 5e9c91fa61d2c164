"""GPU parity tests of the Branch & Bound path (lpr_bb_*) against the CPU oracle: node records,
pop order, every dual / primal pivot of every child LP, incumbent bits -- all identical."""
import json
import os
import struct

import numpy as np
import pytest

import bb_cases

pytestmark = pytest.mark.gpu


def bits(x):
    return struct.pack(">d", float(x)).hex()


def gpu_run(engine, T, n, cap, pruning=False):
    from lpr_381_group_v22_amd import BranchBoundTree
    tree = BranchBoundTree.from_array(engine, T, n, max_depth=max(cap, 20))
    res, x = tree.run(enable_pruning=pruning, node_cap=cap)
    out = dict(status=res.status, found=bool(res.found), z=res.z, x=x,
               processed=res.processed, best_node=res.best_node, records=tree.records(),
               pop_order=tree.pop_order(), trace=tree.trace(), pivots=res.pivots)
    tree.destroy()
    return out


@pytest.mark.parametrize("name,case", bb_cases.all_bb_cases(),
                         ids=[c[0] for c in bb_cases.all_bb_cases()])
@pytest.mark.parametrize("cap", [20, 60])
def test_dfs_run_matches_oracle(engine, oracle, name, case, cap):
    obj, cons = case
    st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
    ref = oracle.bb_solve(T, n, node_cap=cap)
    got = gpu_run(engine, T, n, cap)
    assert got["status"] == ref["status"]
    assert got["processed"] == ref["processed"]
    assert got["pop_order"] == ref["pop_order"]
    assert got["records"] == ref["records"]
    assert got["trace"] == ref["trace"]
    assert got["found"] == ref["found"] and bits(got["z"]) == bits(ref["z"])
    if ref["found"]:
        assert [bits(v) for v in got["x"]] == [bits(v) for v in ref["x"]]
        assert got["best_node"] == ref["best_node"]


def test_pruning_matches_oracle(engine, oracle):
    obj, cons = bb_cases.random_binary_program(8, 3, 3)
    st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
    ref = oracle.bb_solve(T, n, enable_pruning=True, node_cap=80)
    got = gpu_run(engine, T, n, 80, pruning=True)
    assert got["pop_order"] == ref["pop_order"] and got["records"] == ref["records"]
    assert bits(got["z"]) == bits(ref["z"])


def test_adapter_route_option3(engine, oracle):
    """Program.cs option 3: PrimalSimplexSolver.Solve() then BranchAndBoundAdapter.SolveFromPrimal
    on the device-resident FinalTableau (no host round trip of the tableau)."""
    from lpr_381_group_v22_amd import BranchAndBoundAdapter, Constraint, PrimalSimplexSolver
    obj, cons = bb_cases.knapsack_sample()
    primal = PrimalSimplexSolver(obj, [Constraint(list(c.Coefficients), c.Relation, c.RHS)
                                       for c in cons], True, engine=engine, snapshots="none")
    with pytest.raises(RuntimeError, match="has not been solved yet"):
        BranchAndBoundAdapter.SolveFromPrimal(primal)  # BranchAndBoundAdapter.cs:11-14
    primal.Solve()
    x, z = BranchAndBoundAdapter.SolveFromPrimal(primal, enablePruning=False, isMin=False)
    st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
    ref = oracle.bb_solve(T, n, node_cap=20)
    assert bits(z) == bits(ref["z"]) and [bits(v) for v in x] == [bits(v) for v in ref["x"]]
    assert z == 15.0 and x == [0.0, 1.0, 1.0, 1.0, 0.0, 1.0]


def test_expand_building_block_tableaux(engine, oracle):
    """lpr_bb_expand: AddConstraint + DoDualSimplex + RoundAllTableaux on both children of the
    root; the resulting node tableaux are bit-identical to the oracle's."""
    from lpr_381_group_v22_amd import BranchBoundTree, branch_and_bound as bbm
    for name, (obj, cons) in bb_cases.all_bb_cases()[:6]:
        st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
        tree = BranchBoundTree.from_array(engine, T, n, max_depth=8)
        z, vals = tree.node_info([0])
        root = tree.node_read(0)
        want_root = np.vectorize(oracle.round4)(T)
        assert root.tobytes() == want_root.tobytes(), name
        k, val = bbm.choose_branch(vals[0])
        if k < 0:
            tree.destroy()
            continue
        lo, hi = float(np.floor(val)), float(np.ceil(val))
        child, st2, piv = tree.expand([0, 0], [k, k], [lo, hi], [0, 1])
        for side, bound in ((0, lo), (1, hi)):
            con = np.zeros(n + 2)
            con[k] = 1.0
            con[n] = bound
            con[n + 1] = float(side)
            adj = oracle.bb_add_constraint(want_root, con)
            rc, last, npiv, tr = oracle.bb_dual_simplex(adj)
            assert {0: 2, 1: 3, 2: 4}[rc] == st2[side], (name, side)
            if rc == 0:
                assert piv[side] == npiv
                want = np.vectorize(oracle.round4)(last)
                assert tree.node_read(int(child[side])).tobytes() == want.tobytes(), (name, side)
        tree.destroy()


def test_level_synchronous_single_rank_equals_uncapped_dfs(engine, oracle):
    """Cap lifted: the level-synchronous driver explores the same tree as the reference's stack
    and resolves ties the way its pop order does."""
    from lpr_381_group_v22_amd import BranchBoundTree, solve_level_synchronous
    done = 0
    for name, (obj, cons) in bb_cases.all_bb_cases():
        st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
        ref = oracle.bb_solve(T, n, node_cap=300, rec_cap=1 << 12, piv_cap=1 << 18)
        if ref["status"] != 0:
            continue  # the reference never terminates on this instance (only its cap stops it)
        tree = BranchBoundTree.from_array(engine, T, n, max_depth=64)
        got = solve_level_synchronous(tree, n)
        tree.destroy()
        assert got["found"] == ref["found"], name
        assert got["processed"] == ref["processed"], name
        # net pivots (tableaux.Count - 1 per child): a "last tableau dropped" marker (phase 2,
        # :395-400) takes one pivot back
        net = sum(1 for t in ref["trace"] if t[1] < 2) - sum(1 for t in ref["trace"] if t[1] == 2)
        assert got["pivots"] == net, name
        if ref["found"]:
            assert bits(got["z"]) == bits(ref["z"]), name
            assert [bits(v) for v in got["x"]] == [bits(v) for v in ref["x"]], name
        done += 1
    assert done >= 3


def test_golden_fixture_on_gpu(engine, oracle):
    with open(os.path.join(os.path.dirname(__file__), "golden", "bb_golden.json")) as f:
        gold = json.load(f)
    cases = dict(bb_cases.all_bb_cases())
    for key, g in gold.items():
        name, cap = key.rsplit("@", 1)
        obj, cons = cases[name]
        st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
        got = gpu_run(engine, T, n, int(cap))
        assert got["status"] == g["status"] and got["processed"] == g["processed"], key
        assert got["pop_order"] == g["pop_order"], key
        assert bits(got["z"]) == g["z_bits"], key
        assert ([bits(v) for v in got["x"]] if got["found"] else None) == g["x_bits"], key
        assert [[r["parent"], r["kind"], r["var"], r["status"], bits(r["z"])]
                for r in got["records"]] == g["records"], key
        assert [list(t) for t in got["trace"]] == g["trace"], key


def test_larger_root_batched_children(engine, oracle):
    """A wider root (48 variables, 6 constraints + 48 bound rows): children with hundreds of
    columns, 1024-thread select path."""
    obj, cons = bb_cases.random_binary_program(48, 6, 21)
    st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
    assert T.shape[1] > 100
    ref = oracle.bb_solve(T, n, node_cap=12)
    got = gpu_run(engine, T, n, 12)
    assert got["pop_order"] == ref["pop_order"] and got["records"] == ref["records"]
    assert got["trace"] == ref["trace"]
    assert bits(got["z"]) == bits(ref["z"])
