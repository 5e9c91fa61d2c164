"""GPU parity tests of the Branch & Bound path (lpr_bb_*) against the CPU oracle: node records,
pop order, every dual / primal pivot of every child LP, incumbent bits -- all identical."""
import json
import os
import struct

import numpy as np
import pytest

import bb_cases
from oracle_evaluator import OracleEvaluator

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


def _terminating_cases(oracle):
    out = []
    for name, (obj, cons) in bb_cases.all_bb_cases():
        st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
        ref = oracle.bb_solve(T, n, node_cap=300, rec_cap=1 << 12, piv_cap=1 << 18)
        if ref["status"] == 0:
            out.append((name, T, n, ref))
    return out


def test_level_sync_in_the_library_equals_the_python_mirror_and_the_dfs(engine, oracle):
    """lpr_bb_solve_level_sync (the C ABI form; a single rank needs no communicator) against the
    Python mirror of the same driver and against the reference's DFS without its cap."""
    from lpr_381_group_v22_amd import (BranchBoundTree, solve_level_sync_native,
                                       solve_level_synchronous)
    done = 0
    for name, T, n, ref in _terminating_cases(oracle):
        a = BranchBoundTree.from_array(engine, T, n, max_depth=64)
        got = solve_level_sync_native(a)
        a.destroy()
        b = BranchBoundTree.from_array(engine, T, n, max_depth=64)
        py = solve_level_synchronous(b, n)
        b.destroy()
        assert got["status"] == 0 and got["found"] == py["found"] == ref["found"], name
        assert got["processed"] == py["processed"] == ref["processed"], name
        assert got["pivots"] == py["pivots"] and got["levels"] == py["levels"], name
        assert got["path"] == (tuple(py["path"]) if py["path"] is not None else None), name
        if ref["found"]:
            assert bits(got["z"]) == bits(py["z"]) == bits(ref["z"]), name
            assert [bits(v) for v in got["x"]] == [bits(v) for v in ref["x"]], name
        done += 1
    assert done >= 3


def test_level_sync_bound_outside_int_range(engine, oracle):
    """`(int)Math.Floor(value)` (:870-871) of a relaxation value of 1.1e15: C ABI driver, Python
    mirror over the GPU tree and Python mirror over the oracle's evaluator take the same children
    (bound int.MinValue on both sides) and count the same pivots at every depth limit."""
    from oracle_evaluator import OracleEvaluator
    from lpr_381_group_v22_amd import (BranchBoundTree, solve_level_sync_native,
                                       solve_level_synchronous)
    from lpr_381_group_v22_amd.branch_and_bound import dotnet_int32
    assert dotnet_int32(1125965583789882.0) == -2147483648 == dotnet_int32(float("nan"))
    assert dotnet_int32(-2147483648.9) == -2147483648 and dotnet_int32(2147483647.9) == 2147483647
    assert dotnet_int32(-3.0) == -3
    obj, cons = bb_cases.huge_relaxation_value()
    st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
    assert st == 0
    for levels in (2, 3, 4):
        a = BranchBoundTree.from_array(engine, T, n, max_depth=24)
        nat = solve_level_sync_native(a, max_levels=levels)
        a.destroy()
        b = BranchBoundTree.from_array(engine, T, n, max_depth=24)
        py = solve_level_synchronous(b, n, max_levels=levels)
        b.destroy()
        orc = solve_level_synchronous(OracleEvaluator(oracle, T, n), n, max_levels=levels)
        assert nat["pivots"] == py["pivots"] == orc["pivots"], levels
        assert nat["processed"] == py["processed"] == orc["processed"], levels
        assert nat["found"] == py["found"] == orc["found"], levels
        if nat["found"]:
            assert bits(nat["z"]) == bits(py["z"]) == bits(orc["z"]), levels


def test_inplace_second_child_starts_from_a_parent_that_holds_negative_zeros(engine, oracle):
    """The tree drivers let the second child of a parent take the parent's buffer over
    (k_bb_child_inplace): the stored parent keeps the -0.0 that Math.Round(-0.00001, 4) leaves
    (:1124 / :1187), its children start from +0.0 (:307-313), so the rows k_bb_finish flagged are
    cleaned in place.  An instance on which 40 expanded parents hold a -0.0 (checked on the
    oracle's own nodes): the native level-synchronous search and the reference's DFS against the
    oracle."""
    from oracle_evaluator import OracleEvaluator
    from lpr_381_group_v22_amd import (BranchBoundTree, solve_level_sync_native,
                                       solve_level_synchronous)
    obj, cons = bb_cases.fractional_program(9, 4, 1)
    st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
    assert st == 0
    ev = OracleEvaluator(oracle, T, n)
    seen = {"negz": 0}
    inner = ev.expand

    def expand(parents, var, bound, kind):
        for p in set(int(x) for x in parents):
            if p != 0 and np.any((ev.nodes[p] == 0.0) & np.signbit(ev.nodes[p])):
                seen["negz"] += 1
        return inner(parents, var, bound, kind)

    ev.expand = expand
    orc = solve_level_synchronous(ev, n, max_levels=6)
    assert seen["negz"] >= 10
    tree = BranchBoundTree.from_array(engine, T, n, max_depth=16)
    nat = solve_level_sync_native(tree, max_levels=6)
    tree.destroy()
    for key in ("processed", "pivots", "levels", "found", "status"):
        assert nat[key] == orc[key], key
    assert bits(nat["z"]) == bits(orc["z"])
    if orc["found"]:
        assert [bits(v) for v in nat["x"]] == [bits(v) for v in orc["x"]]
    # and the DFS driver (two children per expansion: the second one in place) node by node
    ref = oracle.bb_solve(T, n, node_cap=60)
    got = gpu_run(engine, T, n, 60)
    assert got["processed"] == ref["processed"] and got["records"] == ref["records"]
    assert got["pop_order"] == ref["pop_order"] and got["trace"] == ref["trace"]
    assert bits(got["z"]) == bits(ref["z"])


def test_rccl_communicator_of_one_rank(engine, oracle):
    """lpr_comm_init -> ncclCommInitRank inside the library (world size 1: the pool gives one GPU);
    the levels' all-reduce(MAX) and the final all-gather really go through RCCL: one all-reduce per
    level, one all-gather per solve."""
    from lpr_381_group_v22_amd import BranchBoundTree, Comm, solve_level_sync_native
    comm = Comm.rccl(engine, 0, 1, Comm.unique_id())
    assert comm.info() == dict(rank=0, world=1, allreduce_calls=0, allgather_calls=0)
    assert comm.all_reduce_max([1.5, -2.0, float("-inf")]) == [1.5, -2.0, float("-inf")]
    name, T, n, ref = _terminating_cases(oracle)[0]
    tree = BranchBoundTree.from_array(engine, T, n, max_depth=64)
    got = solve_level_sync_native(tree, comm)
    tree.destroy()
    info = comm.info()
    assert info["allreduce_calls"] == 1 + got["levels"] and info["allgather_calls"] == 1
    assert got["found"] == ref["found"] and got["processed"] == ref["processed"]
    if ref["found"]:
        assert bits(got["z"]) == bits(ref["z"])
    comm.destroy()


def _two_rank_worker(rank, world, port, case_name, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lpr_381_group_v22_amd as pkg
    from oracle_lib import Oracle
    import bb_cases as cases
    orc = Oracle()
    obj, cons = dict(cases.all_bb_cases())[case_name]
    st, T, n = cases.primal_final_tableau(orc, obj, cons)

    def arm(v):
        t = torch.tensor(v, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()

    def gather(b):
        out = [None] * world
        dist.all_gather_object(out, b)
        return out

    eng = pkg.Engine(0)  # both ranks share the one GPU of the box
    comm = pkg.Comm.custom(rank, world, arm, gather)
    tree = pkg.BranchBoundTree.from_array(eng, T, n, max_depth=64)
    res = pkg.solve_level_sync_native(tree, comm)
    res["calls"] = comm.info()
    tree.destroy()
    comm.destroy()
    eng.close()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in res.items()}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_level_sync_two_ranks_through_the_abi(engine, oracle):
    """Two processes (ranks) on the one GPU of the box, lpr_bb_solve_level_sync on each, the
    collectives carried by a caller-supplied transport (gloo): the frontier of depth 1 is dealt,
    sub-trees stay put, one all-reduce per level, same answer as one rank and as the DFS."""
    import socket
    import tempfile
    import torch.multiprocessing as mp
    from lpr_381_group_v22_amd import BranchBoundTree, solve_level_sync_native
    name, T, n, ref = _terminating_cases(oracle)[0]
    tree = BranchBoundTree.from_array(engine, T, n, max_depth=64)
    one = solve_level_sync_native(tree)
    tree.destroy()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_two_rank_worker, args=(2, port, name, d), nprocs=2, join=True)
        res = [json.load(open(os.path.join(d, f"rank{r}.json"))) for r in range(2)]
    for r in res:
        assert r["found"] == one["found"] == ref["found"]
        assert r["processed"] == one["processed"] == ref["processed"]
        assert r["pivots"] == one["pivots"] and r["levels"] == one["levels"]
        assert r["calls"]["allreduce_calls"] == r["levels"] and r["calls"]["allgather_calls"] == 1
        if ref["found"]:
            assert bits(r["z"]) == bits(one["z"]) == bits(ref["z"])
            assert [bits(v) for v in r["x"]] == [bits(v) for v in ref["x"]]
            assert tuple(r["path"]) == tuple(one["path"])


def _fault_worker(rank, world, port, case_name, out_dir, fault):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LPR_BB_INJECT_FAULT"] = fault  # "<rank>:<level>", read by the library
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lpr_381_group_v22_amd as pkg
    from lpr_381_group_v22_amd import _native as N
    from oracle_lib import Oracle
    import bb_cases as cases
    orc = Oracle()
    obj, cons = dict(cases.all_bb_cases())[case_name]
    st, T, n = cases.primal_final_tableau(orc, obj, cons)

    def arm(v):
        t = torch.tensor(v, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()

    def gather(b):
        out = [None] * world
        dist.all_gather_object(out, b)
        return out

    eng = pkg.Engine(0)
    comm = pkg.Comm.custom(rank, world, arm, gather)
    tree = pkg.BranchBoundTree.from_array(eng, T, n, max_depth=64)
    out = {"status": None, "message": ""}
    try:
        pkg.solve_level_sync_native(tree, comm)
        out["status"] = 0
    except N.EngineError as exc:
        out["status"] = exc.status
        out["message"] = str(exc)
    out["calls"] = comm.info()
    tree.destroy()
    comm.destroy()
    eng.close()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_level_sync_failure_on_one_rank_is_returned_by_every_rank(engine, oracle, world):
    """A rank-local failure inside lpr_bb_solve_level_sync (injected: LPR_BB_INJECT_FAULT) must not
    leave the peers in the level's collective for ever: it rides in that all-reduce (4th MAX slot),
    every rank returns LPR_DEVICE_ERROR from the same level and nobody enters the final gather."""
    import socket
    import tempfile
    import time
    import torch.multiprocessing as mp
    case, fault_level = "binary_10v2c_s4", 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as d:
        ctx = mp.spawn(_fault_worker, args=(world, port, case, d, f"{world - 1}:{fault_level}"),
                       nprocs=world, join=False)
        t_end = time.monotonic() + 240
        while not ctx.join(timeout=1.0):
            if time.monotonic() > t_end:
                for p in ctx.processes:
                    if p.is_alive():
                        p.kill()
                pytest.fail("ranks still running: a peer is waiting in a collective")
        res = [json.load(open(os.path.join(d, f"rank{r}.json"))) for r in range(world)]
    for r, out in enumerate(res):
        assert out["status"] == -2, out  # LPR_DEVICE_ERROR on EVERY rank
        assert out["calls"]["allreduce_calls"] == fault_level + 1, out
        assert out["calls"]["allgather_calls"] == 0, out
        assert ("injected fault" if r == world - 1 else "another rank failed") in out["message"], out


def test_level_sync_max_levels_below_the_depth_scores_the_last_frontier(engine, oracle):
    """with max_levels smaller than the tree depth the children solved by the last level
    are still scored (integer check + incumbent) and the status says the search was cut
    (LPR_BB_DEPTH_CAP); the library and the Python mirror agree."""
    from lpr_381_group_v22_amd import (BranchBoundTree, solve_level_sync_native,
                                       solve_level_synchronous)
    from lpr_381_group_v22_amd import _native as N
    for name, T, n, ref in _terminating_cases(oracle)[:4]:
        full = solve_level_synchronous(OracleEvaluator(oracle, T, n), n)
        for L in (1, 2, len(full["path"]) if full["found"] else 3):
            if L < 1:
                continue
            a = BranchBoundTree.from_array(engine, T, n, max_depth=64)
            got = solve_level_sync_native(a, max_levels=L)
            a.destroy()
            py = solve_level_synchronous(OracleEvaluator(oracle, T, n), n, max_levels=L)
            assert got["status"] == py["status"], (name, L)
            assert got["processed"] == py["processed"] and got["levels"] == py["levels"], (name, L)
            assert got["found"] == py["found"] and got["pivots"] == py["pivots"], (name, L)
            if got["found"]:
                assert bits(got["z"]) == bits(py["z"]) and got["path"] == tuple(py["path"])
            if full["found"] and L == len(full["path"]) and full["levels"] > L:
                # the optimum sits exactly at depth max_levels: found, same z as the full search
                assert got["found"] and bits(got["z"]) == bits(full["z"]), name
    assert N.LPR_BB_DEPTH_CAP == 7


def test_comm_outlives_its_engine_safely(oracle):
    """lpr_engine_close with a live RCCL communicator orphans it (no dangling engine
    pointer): collectives are refused, lpr_comm_destroy stays safe."""
    import lpr_381_group_v22_amd as pkg
    from lpr_381_group_v22_amd import _native as N
    eng = pkg.Engine(0)
    comm = pkg.Comm.rccl(eng, 0, 1, pkg.Comm.unique_id())
    assert comm.all_reduce_max([2.0]) == [2.0]
    eng.close()
    with pytest.raises(N.EngineError):
        comm.all_reduce_max([1.0])
    comm.destroy()
