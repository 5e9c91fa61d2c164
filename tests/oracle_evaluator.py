"""Stand-in for a GPU BranchBoundTree built on the CPU oracle (TEST ONLY): node_info / expand /
release with the semantics of lpr_bb_node_info / lpr_bb_expand / lpr_bb_release."""
import numpy as np


class OracleEvaluator:
    """node_info / expand / release with the semantics of lpr_bb_* on top of the C oracle."""

    def __init__(self, oracle, root, nvars):
        self.o = oracle
        self.n = nvars
        self.nodes = {0: np.array(root, dtype=np.float64)}
        self.next_id = 1
        self.r4 = oracle.bb_round_tableau

    def node_info(self, ids):
        """RoundAllTableaux :1047, GetObjective :892-897, decision values :805-857 -- by the C
        oracle (orc_bb_node_info; a Python loop over 512 x 577 entries per node is too slow for
        the bench-sized instance)."""
        zs, vals = [], []
        for i in ids:
            T, z, v = self.o.bb_node_info(self.nodes[i], self.n)
            self.nodes[i] = T
            zs.append(z)
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
            rc, last, npiv, trace = self.o.bb_dual_simplex(adj)
            # pivots as lpr_bb_expand counts them (tableaux.Count - 1 at the exit, also for a child
            # that ends infeasible): pivots performed minus a dropped last tableau (:395-400)
            done = sum(1 for t in trace if t[0] < 2) - sum(1 for t in trace if t[0] == 2)
            if rc == 0:
                assert done == npiv
                self.nodes[self.next_id] = self.r4(last)
                child.append(self.next_id)
                self.next_id += 1
                st.append(2)
                piv.append(npiv)
            else:
                child.append(-1)
                st.append(3 if rc == 1 else 4)
                piv.append(done if rc == 1 else 0)
        return np.array(child), np.array(st), np.array(piv)

    def release(self, ids):
        for i in ids:
            self.nodes.pop(int(i), None)
