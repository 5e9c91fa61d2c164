"""TEST ONLY -- what ExecuteBranchAndBound (IntegerProgramming/BranchBoundSimplexSolver.cs:1006-1233)
writes to the console, restated on top of the independent Python restatement of the search itself
(tests/ref_py_bb.py) and of the number / table formatting (tests/ref_py_text.py).  Nothing of the
product is used.  Program.cs option 3 captures this text (TeeTextWriter) into the result file.

What cannot be restated: `$"... failed: {e}"` (:1147 / :1207) prints a .NET exception with its stack
trace (source paths and line numbers of the original build); the exception's type and message are
written here, the trace is not -- a run that hits it is not byte-comparable with the C#.
PARITY UNPINNED by the reference (it commits no output)."""
from __future__ import annotations

import math
from typing import List, Optional

from ref_py_bb import INF, BranchAndBound, to_int32
from ref_py_text import CRLF, py_double_to_string, py_format_table


def _g(v) -> str:
    return py_double_to_string(float(v))


def _join(vals) -> str:
    return ", ".join(_g(v) for v in vals)


class NarratedBranchAndBound(BranchAndBound):
    def __init__(self, nvars: int, node_cap: int = 20):
        super().__init__(nvars, node_cap)
        self.console: List[str] = []

    def wl(self, s: str = "") -> None:   # Console.WriteLine
        self.console.append(s + CRLF)

    def w(self, s: str) -> None:         # Console.Write
        self.console.append(s)

    def DisplayTableau(self, tab, caption: str) -> None:  # :623-640
        if tab is None or len(tab) == 0:
            self.wl(f"{caption} (empty)")
            return
        self.wl(py_format_table(self.RoundTableau(tab), self.nvars, caption))

    def ExecuteNarrated(self, initial, enable_pruning=False):
        self.wl("Initiating Branch and Bound Algorithm")                        # :1010
        self.wl("Pruning: Enabled" if enable_pruning else "Pruning: Disabled")   # :1013-1017
        self.wl("-" * 50)                                                        # :1019
        root = self.RoundTableau(initial)                                        # :1021
        optimalSolution: Optional[List[float]] = None
        optimalValue = -INF
        optimalLabel = None
        optimalTableau = None
        branchCount = 0
        childCounters = {}
        stack = [(root, 0, "0", [], None)]                                       # :1029
        iteration = 0
        while stack:
            iteration += 1
            if iteration > self.node_cap:                                        # :1038-1042
                self.wl("Potential infinite loop detected")
                break
            tab, depth, label, path, parent = stack.pop()
            branchCount += 1
            tab = self.RoundTableau(tab)                                         # :1047
            self.wl(f"\n--- Processing branch {label} (Depth {depth}) ---")
            if parent is not None:
                self.wl(f"Parent branch: {parent}")
            self.wl(f"Constraint Path: [{', '.join(path)}]")
            objVal = self.round4(tab[0][-1]) if hasattr(self, "round4") else _r4(tab[0][-1])
            if enable_pruning and optimalSolution is not None and objVal <= optimalValue:
                self.wl(f"branch {label} pruned")
                continue
            vals = self._decision(tab)
            if all(self.IsInteger(v) for v in vals):                             # :943-981
                if objVal > optimalValue:
                    optimalValue, optimalSolution = objVal, vals
                    optimalTableau, optimalLabel = tab, label
                    self.wl(f"New optimal integer solution found: [{_join(vals)}] with value "
                            f"{_g(objVal)}")
                else:
                    self.wl(f"Integer solution found: [{_join(vals)}] with value {_g(objVal)} "
                            f"(not better than current optimal)")
            best, bestValue, minDist = -1, None, INF                             # :829-847
            for i, v in enumerate(vals):
                if not self.IsInteger(v):
                    d = abs((v - math.floor(v)) - 0.5)
                    if d < minDist:
                        minDist, best, bestValue = d, i, v
            if best == -1:                                                       # :1070-1076
                self.wl(f"branch {label}: Integer solution [{_join(vals)}] with value {_g(objVal)}")
                continue
            self.wl(f"Branching on x{best + 1} = {_g(_r4(bestValue))}")           # :868
            n = self.nvars
            lowerBound = [1.0 if i == best else 0.0 for i in range(n)] + \
                [float(to_int32(math.floor(bestValue))), 0.0]
            upperBound = [1.0 if i == best else 0.0 for i in range(n)] + \
                [float(to_int32(math.ceil(bestValue))), 1.0]
            childCounters.setdefault(label, 0)
            kids = []
            for side, (name, bnd, star) in enumerate((("Lower", lowerBound, "t"),
                                                      ("Upper", upperBound, "x"))):
                try:
                    childCounters[label] += 1
                    if label == "0":
                        childLabel = "1" if side == 0 else "2"
                    else:
                        childLabel = f"{label}.{childCounters[label]}"
                    self.w(f"\n{name} Branch (branch {childLabel}): {_join(bnd)} ")
                    for i in range(len(bnd) - 2):
                        if bnd[i] == 0:
                            continue
                        if bnd[i] == 1:
                            self.w(f"x{i + 1} ")
                        else:
                            self.w(f"{_g(bnd[i])}*{star}{i + 1} ")
                    self.w("<= " if bnd[-1] == 0 else ">= ")
                    self.w(f"{_g(bnd[-2])} ")
                    adj = self.AddConstraint(bnd, tab)
                    try:
                        tabs, opt = self.solver.DoDualSimplex(adj)
                    finally:
                        for ph, r, c in self.solver.trace:   # printed as the pivots happen
                            if ph < 2:
                                self.wl(f"pivot @ constraint {r}, column {c + 1}")
                    if opt is None:
                        # (headerRow is empty in override mode: :300 removes its only entry)
                        self.DisplayTableau(tabs[0], f"branch {childLabel}: Infeasible tableau")
                        tabs = []
                    else:
                        if tabs:
                            tabs = [self.RoundTableau(t) for t in tabs]          # :1124 / :1187
                            op = "<=" if side == 0 else ">="
                            desc = f"x{bnd[:-2].index(1.0) + 1} {op} {_g(bnd[-2])}"
                            kids.append((tabs[-1], depth + 1, childLabel, path + [desc], label))
                        self.wl(f"{name} branch (branch {childLabel}) infeasible")   # (sic)
                    if tabs:
                        for i in range(len(tabs) - 1):
                            self.DisplayTableau(tabs[i],
                                                f"branch {childLabel} {name} branch Tableau {i + 1}")
                        self.DisplayTableau(tabs[-1], f"branch {childLabel} {name} branch final tableau")
                except IndexError:
                    self.wl(f"{name} branch (branch {childCounters[label]}) failed: "
                            f"System.ArgumentOutOfRangeException: Index was out of range. Must be "
                            f"non-negative and less than the size of the collection.")
            for k in reversed(kids):                                             # :1210-1213
                stack.append(k)
        self.wl("\n" + "-" * 50)
        self.wl("BRANCH AND BOUND COMPLETED")
        self.wl("-" * 50)
        if optimalSolution is not None:
            self.DisplayTableau(optimalTableau, f"Optimal solution tableau at branch {optimalLabel}")
            self.wl(f"Optimal branch: {optimalLabel}")
            self.wl(f"Optimal integer solution: [{_join(optimalSolution)}]")
            self.wl(f"Optimal value: {_g(optimalValue)}")
        else:
            self.wl("No integer solution found")
        self.wl(f"Total branchs processed: {branchCount}")
        return dict(x=optimalSolution, z=optimalValue, processed=branchCount,
                    text="".join(self.console))


def _r4(x: float) -> float:
    from ref_py_bb import round4
    return round4(x)
