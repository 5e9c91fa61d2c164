"""Host-side mirror of the Branch & Bound surface of the reference:

* ``BranchAndBoundAdapter.SolveFromPrimal`` (IntegerProgramming/BranchAndBoundAdapter.cs:9-24) --
  same arguments, same return ``(x, z)``, same InvalidOperationException condition;
* ``BranchBoundTree`` -- handle wrapper over ``lpr_bb_*`` (include/lpr_engine.h);
* ``solve_level_synchronous`` -- the multi-GPU driver: the frontier of each level is dealt over the
  ranks of a ``torch.distributed`` process group (backend "nccl" = RCCL over xGMI on the MI355X
  node, "gloo" in the CPU tests), every rank expands its share on its own GPU, and ONE
  ``all_reduce(MAX)`` of the incumbent bound per level keeps the ranks in step.

All tableau arithmetic runs on the device through the C ABI; this module only holds the tree
bookkeeping the C# keeps in its ``Stack<...>``.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import _native as N
from . import table_iteration_formater as fmt
from .engine import Engine, Tableau, _dptr, _i32ptr

BB_SOLVED, BB_INFEASIBLE, BB_FAILED = 2, 3, 4  # lpr_bb_expand status codes


class BranchBoundTree:
    """Node pool + batched child evaluation on one GPU (lpr_bb_*)."""

    def __init__(self, engine: Engine, handle: C.c_void_p, nvars: int):
        self.engine = engine
        self._h = handle
        self.nvars = nvars

    @classmethod
    def from_array(cls, engine: Engine, final_tableau: np.ndarray, nvars: int,
                   max_depth: int = 0) -> "BranchBoundTree":
        T = np.ascontiguousarray(final_tableau, dtype=np.float64)
        h = C.c_void_p()
        N.check(N.lib.lpr_bb_create(engine._h, _dptr(T), T.shape[0], T.shape[1], nvars, max_depth,
                                    C.byref(h)), "lpr_bb_create")
        return cls(engine, h, nvars)

    @classmethod
    def from_tableau(cls, tab: Tableau, nvars: int, max_depth: int = 0) -> "BranchBoundTree":
        h = C.c_void_p()
        N.check(N.lib.lpr_bb_create_from_tableau(tab._h, nvars, max_depth, C.byref(h)),
                "lpr_bb_create_from_tableau")
        return cls(tab.engine, h, nvars)

    def destroy(self):
        if self._h:
            N.lib.lpr_bb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # ---- the reference's own search order -------------------------------------------------
    def run(self, enable_pruning: bool = False, node_cap: int = 0):
        opts = N.BBOpts(enable_pruning=1 if enable_pruning else 0, node_cap=node_cap)
        res = N.BBResult()
        x = np.zeros(max(self.nvars, 1), dtype=np.float64)
        N.check(N.lib.lpr_bb_run(self._h, C.byref(opts), _dptr(x), C.byref(res)), "lpr_bb_run")
        return res, (x[: self.nvars] if res.found else None)

    def records(self, cap: int = 1 << 16):
        p = np.zeros(cap, dtype=np.int32)
        k = np.zeros(cap, dtype=np.int32)
        d = np.zeros(cap, dtype=np.int32)
        v = np.zeros(cap, dtype=np.int32)
        b = np.zeros(cap)
        s = np.zeros(cap, dtype=np.int32)
        z = np.zeros(cap)
        n = C.c_int64()
        N.check(N.lib.lpr_bb_records_read(self._h, _i32ptr(p), _i32ptr(k), _i32ptr(d), _i32ptr(v),
                                          _dptr(b), _i32ptr(s), _dptr(z), cap, C.byref(n)),
                "lpr_bb_records_read")
        return [dict(parent=int(p[i]), kind=int(k[i]), depth=int(d[i]), var=int(v[i]),
                     bound=float(b[i]), status=int(s[i]), z=float(z[i])) for i in range(n.value)]

    def pop_order(self, cap: int = 1 << 16) -> List[int]:
        ids = np.zeros(cap, dtype=np.int32)
        n = C.c_int64()
        N.check(N.lib.lpr_bb_pop_order_read(self._h, _i32ptr(ids), cap, C.byref(n)),
                "lpr_bb_pop_order_read")
        return ids[: n.value].tolist()

    def trace(self, cap: int = 1 << 20) -> List[Tuple[int, int, int, int]]:
        q = np.zeros(cap * 4, dtype=np.int32)
        n = C.c_int64()
        N.check(N.lib.lpr_bb_trace_read(self._h, _i32ptr(q), cap, C.byref(n)),
                "lpr_bb_trace_read")
        return [tuple(v) for v in q[: 4 * n.value].reshape(-1, 4).tolist()]

    # ---- building blocks --------------------------------------------------------------------
    def node_info(self, ids: Sequence[int]) -> Tuple[np.ndarray, np.ndarray]:
        ids_a = np.ascontiguousarray(ids, dtype=np.int32)
        k = ids_a.shape[0]
        z = np.zeros(max(k, 1))
        vals = np.zeros((max(k, 1), max(self.nvars, 1)))
        N.check(N.lib.lpr_bb_node_info(self._h, _i32ptr(ids_a), k, _dptr(z), _dptr(vals)),
                "lpr_bb_node_info")
        return z[:k], vals[:k, : self.nvars]

    def expand(self, parents: Sequence[int], var: Sequence[int], bound: Sequence[float],
               kind: Sequence[int]):
        p = np.ascontiguousarray(parents, dtype=np.int32)
        v = np.ascontiguousarray(var, dtype=np.int32)
        b = np.ascontiguousarray(bound, dtype=np.float64)
        kd = np.ascontiguousarray(kind, dtype=np.int32)
        k = p.shape[0]
        child = np.zeros(max(k, 1), dtype=np.int32)
        st = np.zeros(max(k, 1), dtype=np.int32)
        piv = np.zeros(max(k, 1), dtype=np.int32)
        N.check(N.lib.lpr_bb_expand(self._h, k, _i32ptr(p), _i32ptr(v), _dptr(b), _i32ptr(kd),
                                    _i32ptr(child), _i32ptr(st), _i32ptr(piv)), "lpr_bb_expand")
        return child[:k], st[:k], piv[:k]

    def expand_traced(self, parents: Sequence[int], var: Sequence[int], bound: Sequence[float],
                      kind: Sequence[int], shape: Tuple[int, int]):
        """lpr_bb_expand_traced: expand() plus, per child, its pivot triples (phase, row, col) and
        every tableau of DoDualSimplex's list (the one AddConstraint hands it, then one per pivot).
        ``shape`` = (rows, cols) of the PARENT (all parents of one call have the same shape)."""
        p = np.ascontiguousarray(parents, dtype=np.int32)
        v = np.ascontiguousarray(var, dtype=np.int32)
        b = np.ascontiguousarray(bound, dtype=np.float64)
        kd = np.ascontiguousarray(kind, dtype=np.int32)
        k = p.shape[0]
        child = np.zeros(max(k, 1), dtype=np.int32)
        st = np.zeros(max(k, 1), dtype=np.int32)
        piv = np.zeros(max(k, 1), dtype=np.int32)
        ntab = np.zeros(max(k, 1), dtype=np.int32)
        rows, cols = shape[0] + 1, shape[1] + 1
        trace_cap, tab_cap = 4 * (rows + cols) * max(k, 1) + 64, (64 * rows * cols) * max(k, 1)
        for _ in range(2):  # (a second try with the sizes the first one reported)
            trace = np.zeros(3 * trace_cap, dtype=np.int32)
            tabs = np.zeros(tab_cap, dtype=np.float64)
            toff = (C.c_int64 * (k + 1))()
            boff = (C.c_int64 * (k + 1))()
            rc = N.lib.lpr_bb_expand_traced(self._h, k, _i32ptr(p), _i32ptr(v), _dptr(b),
                                            _i32ptr(kd), _i32ptr(child), _i32ptr(st), _i32ptr(piv),
                                            _i32ptr(trace), trace_cap, toff, _dptr(tabs), tab_cap,
                                            boff, _i32ptr(ntab))
            if rc == N.LPR_BAD_ARGUMENT and (toff[k] > trace_cap or boff[k] > tab_cap):
                raise N.EngineError(rc, "lpr_bb_expand_traced (a child LP needed more than 64 "
                                        "tableaux: too large for the narrated path)")
            N.check(rc, "lpr_bb_expand_traced")
            break
        traces, tableaux = [], []
        for q in range(k):
            t = trace[3 * toff[q]:3 * toff[q + 1]].reshape(-1, 3)
            traces.append([tuple(int(x) for x in r) for r in t])
            tb = tabs[boff[q]:boff[q + 1]].reshape(int(ntab[q]), rows, cols) if ntab[q] else \
                np.zeros((0, rows, cols))
            tableaux.append([tb[i].copy() for i in range(int(ntab[q]))])
        return child[:k], st[:k], piv[:k], traces, tableaux

    def release(self, ids: Sequence[int]) -> None:
        a = np.ascontiguousarray(ids, dtype=np.int32)
        if a.shape[0]:
            N.check(N.lib.lpr_bb_release(self._h, _i32ptr(a), a.shape[0]), "lpr_bb_release")

    def node_read(self, node_id: int) -> np.ndarray:
        r, c = C.c_int32(), C.c_int32()
        N.check(N.lib.lpr_bb_node_read(self._h, node_id, None, C.byref(r), C.byref(c)),
                "lpr_bb_node_read")
        out = np.empty((r.value, c.value), dtype=np.float64)
        N.check(N.lib.lpr_bb_node_read(self._h, node_id, _dptr(out), C.byref(r), C.byref(c)),
                "lpr_bb_node_read")
        return out


def _g(v: float) -> str:
    return fmt.dotnet_double_to_string(float(v))


def _join(vals) -> str:  # string.Join(", ", List<double>)
    return ", ".join(_g(v) for v in vals)


def execute_branch_and_bound_narrated(tree: "BranchBoundTree", root_shape: Tuple[int, int],
                                      enable_pruning: bool = False, node_cap: int = 0):
    """ExecuteBranchAndBound (BranchBoundSimplexSolver.cs:1006-1233) driven from the host over the
    engine's building blocks, WRITING WHAT THE C# WRITES to the console (captured by Program.cs
    option 3 into the result file): every node header, the branching lines, "pivot @ constraint r,
    column c" of every pivot (:198 / :276) and every tableau of every child through DisplayTableau
    (:623-640).  The numbers -- scores, tableaux, pivots -- come from the device
    (lpr_bb_node_info, lpr_bb_expand_traced); this function only orders and formats them, and only
    for textbook-sized models (the adapter's narrate="auto": <= 4 096 tableau entries).  Same search,
    same answer as lpr_bb_run.  One thing cannot be reproduced: `failed: {e}` (:1147 / :1207) prints
    a .NET exception with its stack trace; the exception's type and message are printed here, the
    trace is not.  Returns (x or None, z, processed)."""
    n = tree.nvars
    cap = node_cap if node_cap > 0 else 20  # :1038
    print("Initiating Branch and Bound Algorithm")                       # :1010
    print("Pruning: Enabled" if enable_pruning else "Pruning: Disabled")  # :1013 / :1017
    print("-" * 50)                                                       # :1019

    def display(tab: np.ndarray, caption: str) -> None:  # DisplayTableau :623-640
        if tab is None or tab.size == 0:
            print(f"{caption} (empty)")
            return
        print(fmt.Format(_round4_np(np.asarray(tab, dtype=np.float64)), n, caption))

    optimal_x, optimal_z, optimal_label, optimal_tab = None, -math.inf, None, None
    branch_count = 0
    counters = {}
    # (node id, its rows, cols, depth, label, constraint path, parent label)
    stack = [(0, root_shape[0], root_shape[1], 0, "0", [], None)]
    iteration = 0
    while stack:
        iteration += 1
        if iteration > cap:
            print("Potential infinite loop detected")  # :1040
            break
        nid, rows, cols, depth, label, path, parent = stack.pop()
        branch_count += 1
        print(f"\n--- Processing branch {label} (Depth {depth}) ---")  # :1049
        if parent is not None:
            print(f"Parent branch: {parent}")
        print(f"Constraint Path: [{', '.join(path)}]")
        zs, vals = tree.node_info([nid])  # RoundAllTableaux :1047 + GetObjective + ExtractSolution
        z, sol = float(zs[0]), [float(v) for v in vals[0]]
        if enable_pruning and optimal_x is not None and z <= optimal_z:  # :1060, :985-1004
            print(f"branch {label} pruned")
            tree.release([nid])
            continue
        if all(_is_integer(v) for v in sol):  # UpdateOptimalSolution :935-983
            if z > optimal_z:
                optimal_z, optimal_x, optimal_label = z, list(sol), label
                optimal_tab = tree.node_read(nid)
                print(f"New optimal integer solution found: [{_join(sol)}] with value {_g(z)}")
            else:
                print(f"Integer solution found: [{_join(sol)}] with value {_g(z)} (not better "
                      f"than current optimal)")
        k, val = choose_branch(sol)  # CreateBranches :859-890
        if k < 0:
            print(f"branch {label}: Integer solution [{_join(sol)}] with value {_g(z)}")  # :1074
            tree.release([nid])
            continue
        print(f"Branching on x{k + 1} = {_g(_round4(val))}")  # :868
        lower = [1.0 if i == k else 0.0 for i in range(n)] + [float(dotnet_int32(math.floor(val))), 0.0]
        upper = [1.0 if i == k else 0.0 for i in range(n)] + [float(dotnet_int32(math.ceil(val))), 1.0]
        counters.setdefault(label, 0)
        child, st, _piv, traces, tabs = tree.expand_traced(
            [nid, nid], [k, k], [lower[n], upper[n]], [0, 1], (rows, cols))
        kids = []
        for side, (name, bnd, star) in enumerate((("Lower", lower, "t"), ("Upper", upper, "x"))):
            counters[label] += 1
            if label == "0":
                child_label = "1" if side == 0 else "2"
            else:
                child_label = f"{label}.{counters[label]}"
            print(f"\n{name} Branch (branch {child_label}): {_join(bnd)} ", end="")  # :1088 / :1155
            for i in range(n):
                if bnd[i] == 0:
                    continue
                print(f"x{i + 1} " if bnd[i] == 1 else f"{_g(bnd[i])}*{star}{i + 1} ", end="")
            print("<= " if bnd[-1] == 0 else ">= ", end="")
            print(f"{_g(bnd[-2])} ", end="")
            for ph, r, c in traces[side]:  # PerformDualPivot :198 / PerformPrimalPivot :276
                if ph < 2:
                    print(f"pivot @ constraint {r}, column {c + 1}")
            shown = [np.array(t) for t in tabs[side]]
            if st[side] == BB_FAILED:  # an exception escaped DoDualSimplex (:396-399)
                print(f"{name} branch (branch {counters[label]}) failed: "
                      f"System.ArgumentOutOfRangeException: Index was out of range. Must be "
                      f"non-negative and less than the size of the collection.")
                continue
            if st[side] == BB_INFEASIBLE:  # optimalSolution == null :1110 / :1177
                display(shown[0] if shown else None, f"branch {child_label}: Infeasible tableau")
                shown = []
            else:
                shown = [_round4_np(t) for t in shown]  # RoundAllTableaux :1124 / :1187
                op = "<=" if side == 0 else ">="
                desc = f"x{k + 1} {op} {_g(bnd[-2])}"
                kids.append((int(child[side]), rows + 1, cols + 1, depth + 1, child_label,
                             path + [desc], label))
                print(f"{name} branch (branch {child_label}) infeasible")  # (sic) :1130 / :1193
            for i, t in enumerate(shown[:-1]):
                display(t, f"branch {child_label} {name} branch Tableau {i + 1}")
            if shown:
                display(shown[-1], f"branch {child_label} {name} branch final tableau")
        for kid in reversed(kids):  # :1210-1213
            stack.append(kid)
        tree.release([nid])
    for item in stack:
        tree.release([item[0]])
    print("\n" + "-" * 50)
    print("BRANCH AND BOUND COMPLETED")
    print("-" * 50)
    if optimal_x is not None:
        display(optimal_tab, f"Optimal solution tableau at branch {optimal_label}")
        print(f"Optimal branch: {optimal_label}")
        print(f"Optimal integer solution: [{_join(optimal_x)}]")
        print(f"Optimal value: {_g(optimal_z)}")
    else:
        print("No integer solution found")
    print(f"Total branchs processed: {branch_count}")
    return optimal_x, optimal_z, branch_count


class BranchAndBoundAdapter:
    """IntegerProgramming/BranchAndBoundAdapter.cs:7-51."""

    @staticmethod
    def SolveFromPrimal(primal, enablePruning: bool = False, isMin: bool = False,
                        node_cap: int = 0, narrate="auto") -> Tuple[List[float], float]:
        if primal.FinalTableau is None:  # :11-14
            raise RuntimeError("Primal simplex has not been solved yet.")
        # :20  SetNumVars(primal.SolutionVector?.Count ?? InferNumVariables(finalTable))
        nvars = len(primal.SolutionVector) if primal.SolutionVector is not None else \
            max(1, primal.FinalTableau.shape[1] - 1)
        # `isMin` is accepted and ignored, as in the reference (never forwarded, :9,:22)
        shape = (primal.tableau.rows, primal.tableau.cols)
        if narrate == "auto":  # the console narration only for models one would read it for
            narrate = shape[0] * shape[1] <= 4096
        tree = BranchBoundTree.from_tableau(primal.tableau, nvars, max_depth=max(node_cap, 20))
        try:
            if narrate:
                x, z, _ = execute_branch_and_bound_narrated(tree, shape, enablePruning, node_cap)
                if x is None:
                    return [], -math.inf
                return [float(v) for v in x], float(z)
            res, x = tree.run(enable_pruning=enablePruning, node_cap=node_cap)
            tree.last_result = res
            BranchAndBoundAdapter.last_tree_records = tree.records()
        finally:
            tree.destroy()
        if x is None:
            return [], -math.inf  # `(x ?? new List<double>(), z)` :23 with optimalValue = -inf
        return [float(v) for v in x], float(res.z)


# ---------------------------------------------------------------------------------------------
# .NET Framework rounding on the host (tree decisions work on n values per node)
def dotnet_int32(x: float) -> int:
    """`(int)d` of the C# (:870-871) as .NET Framework 4.7.2's x64 JIT compiles it (cvttsd2si):
    outside int's range, or NaN, the result is 0x80000000 (LP values of 1e15 do occur)."""
    if not (-2147483649.0 < x < 2147483648.0):
        return -2147483648
    return int(x)


def _round_int(x: float) -> float:
    if x != x or x in (math.inf, -math.inf):
        return x
    if abs(x) < 9.2e18 and x == float(int(x)):
        return x
    t = x + 0.5
    f = math.floor(t)
    if f == t and math.fmod(t, 2.0) != 0:
        f -= 1.0
    return math.copysign(f, x)


def _round4(x: float) -> float:
    if abs(x) < 1e16:
        x = x * 10000.0
        x = _round_int(x)
        x = x / 10000.0
    return x


def _is_integer(v: float) -> bool:  # BranchBoundSimplexSolver.cs:595-599
    r = _round4(v)
    return abs(r - _round_int(r)) <= 1e-6


def _round_int_np(x: np.ndarray) -> np.ndarray:
    """Vectorised Math.Round(double) (same IEEE operations as _round_int, element-wise)."""
    with np.errstate(invalid="ignore", over="ignore"):
        t = x + 0.5
        f = np.floor(t)
        f = np.where((f == t) & (np.fmod(t, 2.0) != 0), f - 1.0, f)
        r = np.copysign(f, x)
        whole = (np.abs(x) < 9.2e18) & (x == np.trunc(x))
        r = np.where(whole | ~np.isfinite(x), x, r)
    return r


def _round4_np(x: np.ndarray) -> np.ndarray:
    with np.errstate(invalid="ignore", over="ignore"):
        return np.where(np.abs(x) < 1e16, _round_int_np(x * 10000.0) / 10000.0, x)


def score_nodes(vals: np.ndarray):
    """Vectorised IsInteger (:595-599) + CheckIntegerBasicVar (:829-847) for a (nodes x nvars)
    array of decision values: returns (all_integer[nodes], branch_var[nodes] (-1: none),
    branch_value[nodes])."""
    vals = np.asarray(vals, dtype=np.float64)
    if vals.ndim == 1:
        vals = vals.reshape(1, -1)
    if vals.shape[1] == 0:
        k = vals.shape[0]
        return np.ones(k, dtype=bool), -np.ones(k, dtype=np.int64), np.zeros(k)
    r = _round4_np(vals)
    with np.errstate(invalid="ignore"):
        is_int = np.abs(r - _round_int_np(r)) <= 1e-6
        dist = np.where(is_int, np.inf, np.abs((vals - np.floor(vals)) - 0.5))
    var = np.argmin(dist, axis=1)  # first occurrence of the minimum == "strict <, first wins"
    none = np.isinf(dist[np.arange(vals.shape[0]), var]) | np.isnan(dist).all(axis=1)
    var = np.where(none, -1, var)
    value = np.where(none, 0.0, vals[np.arange(vals.shape[0]), np.maximum(var, 0)])
    return is_int.all(axis=1), var, value


def choose_branch(vals: Sequence[float]) -> Tuple[int, float]:
    """CheckIntegerBasicVar :829-847: the non-integer value whose fraction is closest to 0.5
    (strict <, first wins).  Returns (-1, 0.0) when every value is integral."""
    best, best_val, min_dist = -1, 0.0, math.inf
    for i, v in enumerate(vals):
        if not _is_integer(v):
            d = abs((v - math.floor(v)) - 0.5)
            if d < min_dist:
                min_dist, best, best_val = d, i, float(v)
    return best, best_val


def solve_level_synchronous(evaluator, nvars: int, *, rank: int = 0, world: int = 1,
                            all_reduce_max: Optional[Callable[[float], float]] = None,
                            gather: Optional[Callable[[object], list]] = None,
                            enable_pruning: bool = False, max_levels: int = 64,
                            max_nodes: int = 1 << 20):
    """Level-synchronous Branch & Bound with the reference's node rules and its cap lifted.

    ``evaluator`` provides ``node_info(ids) -> (z[], vals[][])``, ``expand(parents, var, bound,
    kind) -> (child_ids, status, pivots)`` and ``release(ids)`` -- a ``BranchBoundTree`` on this
    rank's GPU (the CPU tests plug a stand-in).  Every rank holds the same root (node 0).

    Sharding without moving a tableau: the first L0 = ceil(log2(world)) levels are evaluated by
    every rank (identical, deterministic work on tiny frontiers); the depth-L0 frontier, ordered
    by DFS path, is dealt round-robin and from there on a sub-tree stays on the rank that owns its
    root -- sub-problems are independent LPs, nothing but the incumbent crosses xGMI.

    Per level: each rank scores its frontier nodes, updates its local incumbent, branches, and
    evaluates all its children in ONE batched ``expand``; then ONE ``all_reduce(MAX)`` of five
    doubles: the incumbent objective, "some rank still has nodes", "some rank passed max_nodes",
    "some rank failed", "some rank left nodes unbranched at max_levels" -- so every rank leaves the
    loop on the same level, also when one of them raised (the exception is re-raised on that rank
    and a RuntimeError on the others AFTER the collective; nobody hangs).  The frontier of depth
    max_levels is scored, not branched (status LPR_BB_DEPTH_CAP if a fractional node was left).  ``all_reduce_max`` takes and returns a list.
    (This is the Python mirror of ``lpr_bb_solve_level_sync`` -- the C ABI form, which calls RCCL
    itself -- kept for the CPU tests with a stand-in evaluator.)  With
    pruning off (the reference's setting, Program.cs:389) the explored tree does not depend on the
    rank count, and ties on z go to the node the reference's stack pops first (lower child before
    upper child, parent before child), so the answer equals ExecuteBranchAndBound without its
    20-node cap.  Returns dict(x, z, found, processed, pivots, levels, path, status), same on all ranks."""
    if all_reduce_max is None:
        all_reduce_max = lambda v: list(v)  # noqa: E731  (takes and returns a 5-list)
    if gather is None:
        gather = lambda obj: [obj]  # noqa: E731
    split_level = 0
    while (1 << split_level) < world:
        split_level += 1

    # frontier entries: (node id on this rank, DFS path: tuple of 0 lower / 1 upper)
    frontier: List[Tuple[int, Tuple[int, ...]]] = [(0, ())]
    best_local = None  # (z, path, x)
    best_z = -math.inf
    global_bound = -math.inf
    processed = 0
    pivots = 0
    levels = 0
    capped = depth_capped = False
    failure = None  # (rank-local exception, or the fact that a peer failed)
    while True:
        # a level past max_levels is scored but not branched (lpr_bb_solve_level_sync does the same)
        may_expand = levels < max_levels
        replicated = levels < split_level  # every rank is doing the same nodes
        count_here = (not replicated) or rank == 0
        parents, var, bound, kind, paths = [], [], [], [], []
        new_frontier: List[Tuple[int, Tuple[int, ...]]] = []
        local_err = None  # never raised before the collective: the peers would wait for ever
        unbranched = False
        try:
            if frontier:
                ids = [nid for nid, _ in frontier]
                zs, vals = evaluator.node_info(ids)
                all_int, bvar, bval = score_nodes(vals)
                for q, ((nid, path), z) in enumerate(zip(frontier, zs)):
                    if count_here:
                        processed += 1
                    if enable_pruning and global_bound > -math.inf and z <= global_bound:
                        continue  # ShouldPrunebranch :995-1001 against the all-reduced bound
                    if all_int[q]:  # UpdateOptimalSolution :943-981
                        cand = (float(z), path, [float(t) for t in vals[q]])
                        if best_local is None or cand[0] > best_local[0] or \
                                (cand[0] == best_local[0] and _dfs_before(cand[1], best_local[1])):
                            best_local = cand
                            best_z = max(best_z, cand[0])
                    k = int(bvar[q])  # CreateBranches :859-890
                    if k < 0:
                        continue
                    if not may_expand:
                        unbranched = True
                        continue
                    val = float(bval[q])
                    for side, bnd in ((0, math.floor(val)), (1, math.ceil(val))):
                        parents.append(nid)
                        var.append(k)
                        bound.append(float(dotnet_int32(bnd)))
                        kind.append(side)
                        paths.append(path + (side,))
            if parents:
                child, st, piv = evaluator.expand(parents, var, bound, kind)
                if count_here:
                    pivots += int(np.sum(piv))
                for c, s2, pth in zip(child, st, paths):
                    if s2 == BB_SOLVED:
                        new_frontier.append((int(c), pth))
        except Exception as exc:  # noqa: BLE001 -- carried through the all-reduce below
            local_err = exc
            new_frontier = []
        if frontier:
            evaluator.release([nid for nid, _ in frontier])
        if may_expand:
            levels += 1
        if may_expand and levels == split_level and world > 1 and local_err is None:
            # deal the depth-L0 frontier: identical on every rank, so no communication is needed
            new_frontier.sort(key=lambda e: e[1])
            keep = [e for i, e in enumerate(new_frontier) if i % world == rank]
            drop = [e[0] for i, e in enumerate(new_frontier) if i % world != rank]
            if drop:
                evaluator.release(drop)
            new_frontier = keep
        frontier = new_frontier
        # ---- the single collective of the level (RCCL all-reduce over xGMI): the incumbent bound,
        # "someone still has nodes", "someone passed max_nodes", "someone FAILED", "someone left
        # nodes unbranched at max_levels" ride in the same MAX ----
        red = all_reduce_max([best_z, 1.0 if frontier else 0.0,
                              1.0 if processed > max_nodes else 0.0,
                              1.0 if local_err is not None else 0.0,
                              1.0 if unbranched else 0.0])
        global_bound, busy = float(red[0]), red[1] > 0.5
        capped = capped or red[2] > 0.5
        depth_capped = depth_capped or red[4] > 0.5
        if red[3] > 0.5:  # every rank leaves on this level, with an error
            failure = local_err if local_err is not None else RuntimeError(
                f"solve_level_synchronous: another rank failed at level {levels} "
                f"(this rank, {rank} of {world}, was fine)")
            break
        if not busy or capped or not may_expand:  # decided from the same reduced values everywhere
            break
    if frontier:
        evaluator.release([nid for nid, _ in frontier])
    if failure is not None:
        raise failure
    status = N.LPR_BB_NODE_CAP if capped else (N.LPR_BB_DEPTH_CAP if depth_capped else 0)
    # winner identity: one gather at termination, ties by DFS order (:966 "first found wins")
    cands = [c for c in gather(best_local) if c is not None]
    total_processed = int(sum(gather(processed)))
    total_pivots = int(sum(gather(pivots)))
    if not cands:
        return dict(x=None, z=-math.inf, found=False, processed=total_processed,
                    pivots=total_pivots, levels=levels, path=None, status=status)
    best = cands[0]
    for c in cands[1:]:
        if c[0] > best[0] or (c[0] == best[0] and _dfs_before(c[1], best[1])):
            best = c
    return dict(x=best[2], z=best[0], found=True, processed=total_processed,
                pivots=total_pivots, levels=levels, path=best[1], status=status)


def _dfs_before(a: Tuple[int, ...], b: Tuple[int, ...]) -> bool:
    """True if node `a` is popped before node `b` by the reference's stack (pre-order, lower
    child first): lexicographic order on the branch paths, a prefix (ancestor) first."""
    return a < b


def torch_collectives(group=None):
    """(all_reduce_max, gather) over a torch.distributed process group -- backend "nccl" is RCCL
    on the MI355X node (xGMI), "gloo" on CPU.  The all-reduce moves one 16-byte tensor."""
    import torch
    import torch.distributed as dist

    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"

    def all_reduce_max(v):
        t = torch.tensor(list(v), dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return t.cpu().tolist()

    def gather(obj):
        out = [None] * dist.get_world_size(group)
        dist.all_gather_object(out, obj, group=group)
        return out

    return all_reduce_max, gather


class Comm:
    """lpr_comm: the multi-GPU communicator of the C ABI.  ``Comm.rccl`` calls RCCL
    (ncclCommInitRank) inside the library; ``Comm.custom`` plugs two Python callables (tests over
    gloo).  The Branch & Bound levels issue ONE all-reduce(MAX) each through it."""

    def __init__(self, handle, keep=()):
        self._h = handle
        self._keep = keep  # ctypes callbacks must outlive the handle

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * N.LPR_COMM_ID_BYTES)()
        N.check(N.lib.lpr_comm_unique_id(buf), "lpr_comm_unique_id")
        return bytes(buf)

    @classmethod
    def rccl(cls, engine: Engine, rank: int, world: int, unique_id: bytes) -> "Comm":
        buf = (C.c_uint8 * N.LPR_COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        N.check(N.lib.lpr_comm_init(engine._h, rank, world, buf, C.byref(h)), "lpr_comm_init")
        return cls(h)

    @classmethod
    def custom(cls, rank: int, world: int, all_reduce_max: Callable, all_gather: Callable) -> "Comm":
        """all_reduce_max(list[float]) -> list[float]; all_gather(bytes) -> list[bytes] (by rank)."""
        def _ar(_user, ptr, count):
            try:
                out = all_reduce_max([ptr[i] for i in range(count)])
                for i in range(count):
                    ptr[i] = out[i]
                return 0
            except Exception:  # never unwind through the C frames
                return 1

        def _ag(_user, send, recv, nbytes):
            try:
                parts = all_gather(C.string_at(send, nbytes))
                C.memmove(recv, b"".join(parts), nbytes * len(parts))
                return 0
            except Exception:
                return 1

        ar, ag = N.ALLREDUCE_MAX_FN(_ar), N.ALLGATHER_FN(_ag)
        h = C.c_void_p()
        N.check(N.lib.lpr_comm_init_custom(rank, world, ar, ag, None, C.byref(h)),
                "lpr_comm_init_custom")
        return cls(h, keep=(ar, ag))

    def info(self):
        r, w = C.c_int(), C.c_int()
        a, g = C.c_int64(), C.c_int64()
        N.check(N.lib.lpr_comm_info(self._h, C.byref(r), C.byref(w), C.byref(a), C.byref(g)),
                "lpr_comm_info")
        return dict(rank=r.value, world=w.value, allreduce_calls=a.value, allgather_calls=g.value)

    def all_reduce_max(self, values: Sequence[float]) -> List[float]:
        buf = (C.c_double * len(values))(*values)
        N.check(N.lib.lpr_comm_all_reduce_max(self._h, buf, len(values)),
                "lpr_comm_all_reduce_max")
        return list(buf)

    def destroy(self):
        if self._h:
            N.lib.lpr_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def solve_level_sync_native(tree: "BranchBoundTree", comm: Optional[Comm] = None, *,
                            enable_pruning: bool = False, max_levels: int = 0,
                            max_nodes: int = 0):
    """lpr_bb_solve_level_sync: the level-synchronous multi-rank Branch & Bound inside the library
    (its collectives are RCCL calls made by the library itself).  Same dict as
    solve_level_synchronous."""
    opts = N.BBSyncOpts(enable_pruning=1 if enable_pruning else 0, max_levels=max_levels,
                        max_nodes=max_nodes)
    res = N.BBSyncResult()
    x = np.zeros(max(tree.nvars, 1), dtype=np.float64)
    N.check(N.lib.lpr_bb_solve_level_sync(tree._h, comm._h if comm else None, C.byref(opts),
                                          x.ctypes.data_as(C.POINTER(C.c_double)),
                                          C.byref(res)), "lpr_bb_solve_level_sync")
    path = tuple((res.path_bits >> k) & 1 for k in range(res.path_len)) if res.found else None
    return dict(x=[float(v) for v in x[:tree.nvars]] if res.found else None,
                z=res.z if res.found else -math.inf, found=bool(res.found),
                processed=int(res.processed), pivots=int(res.pivots), levels=int(res.levels),
                path=path, status=int(res.status))
