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


class OracleEvaluator:
    """node_info / expand / release with the semantics of lpr_bb_* on top of the C oracle."""

    def __init__(self, oracle, root, nvars):
        self.o = oracle
        self.n = nvars
        self.nodes = {0: np.array(root, dtype=np.float64)}
        self.next_id = 1
        self.r4 = np.vectorize(oracle.round4, otypes=[np.float64])

    def node_info(self, ids):
        zs, vals = [], []
        for i in ids:
            T = self.r4(self.nodes[i])
            self.nodes[i] = T
            zs.append(self.o.round4(T[0, -1]))
            v = []
            for k in range(self.n):
                val = 0.0
                for j in range(T.shape[0]):
                    if abs(self.o.round4(T[j, k]) - 1.0) <= 1e-6:
                        val = self.o.round4(T[j, -1])
                        break
                v.append(val)
            vals.append(v)
        return np.array(zs), np.array(vals).reshape(len(ids), self.n)

    def expand(self, parents, var, bound, kind):
        child, st, piv = [], [], []
        for p, k, b, kd in zip(parents, var, bound, kind):
            con = np.zeros(self.n + 2)
            con[k] = 1.0
            con[self.n] = b
            con[self.n + 1] = float(kd)
            adj = self.o.bb_add_constraint(self.nodes[p], con)
            rc, last, npiv, _ = self.o.bb_dual_simplex(adj)
            if rc == 0:
                self.nodes[self.next_id] = self.r4(last)
                child.append(self.next_id)
                self.next_id += 1
                st.append(2)
                piv.append(npiv)
            else:
                child.append(-1)
                st.append(3 if rc == 1 else 4)
                piv.append(npiv if rc == 1 else 0)
        return np.array(child), np.array(st), np.array(piv)

    def release(self, ids):
        for i in ids:
            self.nodes.pop(int(i), None)


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
