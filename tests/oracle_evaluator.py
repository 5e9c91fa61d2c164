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
