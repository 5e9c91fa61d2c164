"""CPU test of the multi-rank Branch & Bound driver (world_size 2, gloo): the sharding logic, the
single all-reduce per level and the winner resolution of
``lpr_381_group_v22_amd.branch_and_bound.solve_level_synchronous``.

There is no GPU here, so the per-node LP work is done by a stand-in evaluator built on the CPU
oracle (tests may use the oracle; the product never does).  On the MI355X node the same driver runs
with a ``BranchBoundTree`` per rank and backend "nccl" (= RCCL over xGMI) -- tests/test_bb_gpu.py
covers that evaluator on one GPU."""
import json
import os
import socket
import struct
import sys
import tempfile

import numpy as np
import pytest

import bb_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(x):
    return struct.pack(">d", float(x)).hex()


from oracle_evaluator import OracleEvaluator  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case_name, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle_lib import Oracle
    import bb_cases as cases
    from lpr_381_group_v22_amd.branch_and_bound import solve_level_synchronous, torch_collectives
    orc = Oracle()
    obj, cons = dict(cases.all_bb_cases())[case_name]
    st, T, n = cases.primal_final_tableau(orc, obj, cons)
    ev = OracleEvaluator(orc, T, n)
    arm, gather = torch_collectives()
    calls = {"n": 0}

    def counted(v):
        calls["n"] += 1
        return arm(v)

    res = solve_level_synchronous(ev, n, rank=rank, world=world, all_reduce_max=counted,
                                  gather=gather)
    res["collectives"] = calls["n"]
    res["leftover_nodes"] = len(ev.nodes)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in res.items()}, f)
    dist.barrier()
    dist.destroy_process_group()


TERMINATING = ["knapsack_sample", "binary_4v1c_s0", "frac_4v2c_s10"]


@pytest.mark.parametrize("case_name", TERMINATING)
def test_two_ranks_equal_one_rank_and_the_reference_order(oracle, case_name):
    import torch.multiprocessing as mp
    from lpr_381_group_v22_amd.branch_and_bound import solve_level_synchronous
    obj, cons = dict(bb_cases.all_bb_cases())[case_name]
    st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
    ref = oracle.bb_solve(T, n, node_cap=300, rec_cap=1 << 12, piv_cap=1 << 18)
    assert ref["status"] == 0, "pick instances on which the reference's search terminates"
    one = solve_level_synchronous(OracleEvaluator(oracle, T, n), n)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), case_name, d), nprocs=2, join=True)
        res = [json.load(open(os.path.join(d, f"rank{r}.json"))) for r in range(2)]
    for r in res:
        # identical answer on every rank, equal to the single-rank run and to the DFS oracle
        assert r["found"] == one["found"] == ref["found"]
        assert r["processed"] == one["processed"] == ref["processed"]
        assert r["levels"] == one["levels"]
        if ref["found"]:
            assert bits(r["z"]) == bits(one["z"]) == bits(ref["z"])
            assert [bits(v) for v in r["x"]] == [bits(v) for v in ref["x"]]
            assert tuple(r["path"]) == tuple(one["path"])
        assert r["collectives"] == r["levels"], "exactly one all-reduce per level"
        assert r["leftover_nodes"] == 0, "every node buffer is released"


class _FailingEvaluator:
    """An evaluator whose `fail_call`-th expand raises (a device error on one rank's GPU)."""

    def __init__(self, inner, fail_call):
        self.inner, self.fail_call, self.calls = inner, fail_call, 0
        self.nodes = inner.nodes

    def node_info(self, ids):
        return self.inner.node_info(ids)

    def expand(self, *a):
        self.calls += 1
        if self.calls - 1 == self.fail_call:
            raise RuntimeError("injected evaluator failure")
        return self.inner.expand(*a)

    def release(self, ids):
        return self.inner.release(ids)


def _failing_worker(rank, world, port, case_name, out_dir, bad_rank, fail_level):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle_lib import Oracle
    import bb_cases as cases
    from lpr_381_group_v22_amd.branch_and_bound import solve_level_synchronous, torch_collectives
    orc = Oracle()
    obj, cons = dict(cases.all_bb_cases())[case_name]
    st, T, n = cases.primal_final_tableau(orc, obj, cons)
    ev = OracleEvaluator(orc, T, n)
    if rank == bad_rank:
        ev = _FailingEvaluator(ev, fail_level)
    arm, gather = torch_collectives()
    calls = {"n": 0}

    def counted(v):
        calls["n"] += 1
        return arm(v)

    out = {"collectives": None, "error": None}
    try:
        solve_level_synchronous(ev, n, rank=rank, world=world, all_reduce_max=counted,
                                gather=gather)
    except RuntimeError as exc:
        out["error"] = str(exc)
    out["collectives"] = calls["n"]
    out["leftover_nodes"] = len(ev.nodes)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def _spawn_with_deadline(fn, args, nprocs, seconds):
    """mp.spawn that cannot hang the suite: a rank still alive at the deadline is a failure."""
    import time
    import torch.multiprocessing as mp
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    t_end = time.monotonic() + seconds
    while not ctx.join(timeout=1.0):
        if time.monotonic() > t_end:
            for p in ctx.processes:
                if p.is_alive():
                    p.kill()
            pytest.fail(f"ranks still running after {seconds} s: a collective was skipped")


@pytest.mark.parametrize("world", [2, 3])
def test_failure_on_one_rank_reaches_every_rank_on_the_same_level(oracle, world):
    """One rank's evaluator fails at level 2 (after the frontier has been dealt): the failure rides
    in that level's all-reduce, every rank raises after the SAME number of collectives, none waits
    in a collective its peer never enters (the reference's analogue is the
    per-branch catch of BranchBoundSimplexSolver.cs:1145-1148,1205-1208)."""
    case_name, fail_level = "binary_10v2c_s4", 2
    bad = world - 1
    with tempfile.TemporaryDirectory() as d:
        _spawn_with_deadline(_failing_worker, (world, _free_port(), case_name, d, bad, fail_level),
                             world, 120)
        res = [json.load(open(os.path.join(d, f"rank{r}.json"))) for r in range(world)]
    for r, out in enumerate(res):
        assert out["error"] is not None, f"rank {r} did not learn of the failure"
        assert out["collectives"] == fail_level + 1, out
        assert out["leftover_nodes"] == 0
        if r == bad:
            assert "injected evaluator failure" in out["error"]
        else:
            assert "another rank failed" in out["error"]


class _CountingEvaluator(OracleEvaluator):
    def __init__(self, *a):
        super().__init__(*a)
        self.level_sizes = []

    def node_info(self, ids):
        self.level_sizes.append(len(ids))
        return super().node_info(ids)


@pytest.mark.parametrize("case_name", ["knapsack_sample", "binary_10v2c_s4", "frac_9v2c_s12"])
def test_max_levels_below_the_tree_depth_scores_the_last_frontier(oracle, case_name):
    """children solved by the last branched level must still be scored (integer check,
    incumbent) and a truncated search must say so (LPR_BB_DEPTH_CAP), not report a full solve."""
    from lpr_381_group_v22_amd import _native as N
    from lpr_381_group_v22_amd.branch_and_bound import solve_level_synchronous
    obj, cons = dict(bb_cases.all_bb_cases())[case_name]
    st, T, n = bb_cases.primal_final_tableau(oracle, obj, cons)
    ev = _CountingEvaluator(oracle, T, n)
    full = solve_level_synchronous(ev, n, max_levels=40, max_nodes=4000)
    sizes = ev.level_sizes
    assert full["status"] in (0, N.LPR_BB_NODE_CAP) and len(sizes) >= 3
    if full["found"] and 0 < len(full["path"]) < len(sizes) - 1:
        # an integer optimum sitting exactly at depth max_levels is found, with the same z
        L = len(full["path"])
        cut = solve_level_synchronous(OracleEvaluator(oracle, T, n), n, max_levels=L)
        assert cut["found"] and cut["processed"] == sum(sizes[:L + 1])
        assert cut["z"] >= full["z"] or full["status"] != 0
        if full["status"] == 0:
            assert bits(cut["z"]) == bits(full["z"]) and tuple(cut["path"]) == tuple(full["path"])
    for L in (1, 2):
        ev2 = OracleEvaluator(oracle, T, n)
        cut = solve_level_synchronous(ev2, n, max_levels=L)
        assert cut["levels"] == L
        assert cut["processed"] == sum(sizes[:L + 1]), "the depth-L frontier is scored too"
        assert cut["status"] == N.LPR_BB_DEPTH_CAP
        assert not ev2.nodes, "every node buffer is released"


def test_tie_on_z_goes_to_the_dfs_first_node():
    """Two integer nodes with the same z on different ranks: the winner is the one the reference's
    stack would have popped first (lower branch before upper)."""
    from lpr_381_group_v22_amd.branch_and_bound import _dfs_before
    assert _dfs_before((0, 1), (1,)) and _dfs_before((0,), (0, 0)) and not _dfs_before((1,), (0, 1, 1))
