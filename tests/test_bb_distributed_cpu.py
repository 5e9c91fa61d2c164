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


def test_tie_on_z_goes_to_the_dfs_first_node():
    """Two integer nodes with the same z on different ranks: the winner is the one the reference's
    stack would have popped first (lower branch before upper)."""
    from lpr_381_group_v22_amd.branch_and_bound import _dfs_before
    assert _dfs_before((0, 1), (1,)) and _dfs_before((0,), (0, 0)) and not _dfs_before((1,), (0, 1, 1))
