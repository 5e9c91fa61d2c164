"""Host-side mirror of the solver-selection surface of the reference: ``Program.cs`` main menu
options 1-3 (``Program.cs:89-415``), ``CanonicalFormConverter`` (``Utilities/
CanonicalFormConverter.cs:15-98``) and ``OutputFileWrite`` (``IO/OutputFileWrite.cs:16-136``), so that
the same text model file drives the MI355X engine and produces the same ``output_results.txt``
layout.  Everything numeric goes through the C ABI; this file is text plumbing.

Not mirrored (not on the pivot path): the sensitivity sub-menu
(``Program.cs:158-294``), option 5 (does not compile in the reference, ``Program.cs:444,468``) and
option 6 (golden-section demo).  PARITY UNPINNED for the exact text: the reference commits no
output file to compare with (``data/output_results.txt`` is empty).

    python -m lpr_381_group_v22_amd.program data/model.txt 1 [output_results.txt]
"""
from __future__ import annotations

import datetime
import io
import os
import sys
from contextlib import redirect_stdout
from typing import List, Optional, Sequence

from . import table_iteration_formater as fmt
from .branch_and_bound import BranchAndBoundAdapter
from .input_file_parser import Constraint, InputFileParser
from .primal_simplex_solver import PrimalSimplexSolver
from .revised_primal_simplex_solver import RevisedPrimalSimplexSolver

NL = fmt.NEWLINE


dotnet_double_to_string = fmt.dotnet_double_to_string  # double.ToString() ("G")


def _format_coeff(c: float) -> str:  # CanonicalFormConverter.cs:95-98
    return f"+ {dotnet_double_to_string(c)}" if c >= 0 else dotnet_double_to_string(c)


def canonical_form_for_file(problemType: str, objective: Sequence[float],
                            constraints: Sequence[Constraint], signs: Sequence[str]) -> str:
    """CanonicalFormConverter.CanonicalFormForFile, :55-93 (always "+ S_i", relation ignored)."""
    out = ["\n=== Canonical Form ===" + NL, "Z "]
    for i, c in enumerate(objective):
        out.append(f"{_format_coeff(c * -1)}x{i + 1} ")
    out.append("= 0\n")
    for i, con in enumerate(constraints):
        for j, a in enumerate(con.Coefficients):
            out.append(f"{_format_coeff(a)}x{j + 1} ")
        out.append(f"+ S{i + 1} ")
        out.append(f"= {dotnet_double_to_string(con.RHS)}\n")
    out.append("\nSign Restrictions: ")
    for i, s in enumerate(signs):
        out.append(f"x{i + 1}: {s} ")
    out.append("\n======================\n" + NL)
    return "".join(out)


def display_canonical_form(problemType: str, objective, constraints, signs) -> None:
    """CanonicalFormConverter.DisplayCanonicalForm, :15-52."""
    print("\n=== Canonical Form ===")
    print(f"{problemType.upper()} Z ", end="")
    for i, c in enumerate(objective):
        print(f"{_format_coeff(c * -1)}x{i + 1} ", end="")
    print("= 0\n")
    for i, con in enumerate(constraints):
        for j, a in enumerate(con.Coefficients):
            print(f"{_format_coeff(a)}x{j + 1} ", end="")
        print(f"+ S{i + 1} ", end="")
        print(f"= {dotnet_double_to_string(con.RHS)}")
    print()
    print("Sign Restrictions: ", end="")
    for i, s in enumerate(signs):
        print(f"x{i + 1}: {s} ", end="")
    print("\n======================\n")


def write_full_results(filePath: str, solverUsed: str, problemType: str, objective, constraints,
                       signs, snapshots: Optional[List[str]], finalZ: float,
                       solution: Optional[Sequence[float]], append: bool = False) -> None:
    """OutputFileWrite.WriteFullResults, IO/OutputFileWrite.cs:16-78."""
    sb = ["=" * 60 + NL, f"Solver: {solverUsed}" + NL, f"Problem type: {problemType}" + NL,
          f"Timestamp: {datetime.datetime.now():%Y-%m-%d %H:%M:%S}" + NL, "=" * 60 + NL]
    try:
        sb.append(canonical_form_for_file(problemType, objective, constraints, signs))
    except Exception:
        sb.append("[Canonical form unavailable]" + NL)
    if snapshots:
        sb.append("=== Iteration Snapshots ===" + NL)
        for i, s in enumerate(snapshots):
            sb.append(f"--- Iteration {i + 1} ---" + NL)
            sb.append(s + NL)
        sb.append(NL)
    sb.append("=== Final Results ===" + NL)
    sb.append(f"Z* = {fmt.N3(finalZ)}" + NL)
    if solution:
        for i, v in enumerate(solution):
            sb.append(f"x{i + 1} = {fmt.N3(v)}" + NL)
    _write(filePath, "".join(sb), append)


def write_snapshots_only(filePath: str, solverUsed: str, snapshots: Optional[List[str]],
                         finalZ: float, solution: Optional[Sequence[float]],
                         append: bool = True) -> None:
    """OutputFileWrite.WriteSnapshotsOnly, :83-119."""
    sb = ["=" * 60 + NL, f"Solver: {solverUsed}" + NL,
          f"Timestamp: {datetime.datetime.now():%Y-%m-%d %H:%M:%S}" + NL, "=" * 60 + NL]
    if snapshots:
        sb.append("=== Solver Log ===" + NL)
        for s in snapshots:
            sb.append(s + NL)
            if not s.endswith("\n"):
                sb.append(NL)
    sb.append("=== Final Results ===" + NL)
    sb.append(f"Z* = {fmt.N3(finalZ)}" + NL)
    if solution:
        for i, v in enumerate(solution):
            sb.append(f"x{i + 1} = {fmt.N3(v)}" + NL)
    _write(filePath, "".join(sb), append)


def _write(path: str, content: str, append: bool) -> None:
    d = os.path.dirname(path)
    if d and not os.path.isdir(d):
        os.makedirs(d)
    mode = "a" if (append and os.path.exists(path)) else "w"
    with open(path, mode, encoding="utf-8-sig" if mode == "w" else "utf-8", newline="") as f:
        f.write(content)  # File.WriteAllText(..., Encoding.UTF8) writes a BOM


def add_upper_bound_constraints(n: int, signs: Sequence[str],
                                constraints: List[Constraint]) -> None:
    """Program.AddUpperBoundConstraints, Program.cs:511-535 (option 2)."""
    if not signs:
        return
    for j in range(n):
        sr = signs[min(j, len(signs) - 1)] or ""
        s = sr.replace(" ", "")
        if "bin" in s.lower() or "≤1" in s or "<=1" in s:
            co = [0.0] * n
            co[j] = 1.0
            constraints.append(Constraint(co, "<=", 1.0))


def _append_unit_bound_rows(parser: InputFileParser) -> None:
    """Program.cs:114-124 / :372-382: n rows "x_i <= 1" with n + 3 coefficients, appended to the
    PARSER's own list (so choosing option 1 or 3 twice appends twice -- reference behaviour)."""
    n = len(parser.ObjectiveCoefficients)
    for i in range(n):
        co = [0.0] * (n + 3)
        co[i] = 1.0
        co[n + 1] = 1.0
        parser.Constraints.append(Constraint(co, "<=", 1.0))


def run_option(parser: InputFileParser, choice: str, out_path: str = "data/output_results.txt",
               engine=None) -> dict:
    """One main-menu choice of Program.cs (:89-502) without the console pauses.  Returns the
    numbers the menu prints / writes."""
    is_min = (parser.ProblemType or "").lower() == "min"
    if choice == "1":  # Program.cs:91-151
        if is_min:
            print("\n⚠️ WARNING: This is a Minimization Problem.")
            print("Please use Option 2 (Revised Simplex Method) instead for better results!")
            return {"skipped": True}
        print("Solving with Primal Simplex Algorithm...")
        display_canonical_form(parser.ProblemType, parser.ObjectiveCoefficients,
                               parser.Constraints, parser.SignRestrictions)
        _append_unit_bound_rows(parser)
        s = PrimalSimplexSolver(parser.ObjectiveCoefficients, parser.Constraints, engine=engine,
                                verbose=True)
        s.Solve()
        write_full_results(out_path, "Primal Simplex Algorithm", parser.ProblemType,
                           parser.ObjectiveCoefficients, parser.Constraints,
                           parser.SignRestrictions, s.IterationSnapshots, s.FinalZ,
                           s.SolutionVector)
        print("\nAll results have been saved to 'output_results.txt'.")
        return {"z": s.FinalZ, "x": s.SolutionVector, "solver": s}
    if choice == "2":  # Program.cs:306-354
        print("Solving with Revised Primal Simplex Algorithm...")
        objective2 = list(parser.ObjectiveCoefficients)
        constraints2 = [Constraint(list(c.Coefficients), c.Relation, c.RHS)
                        for c in parser.Constraints]
        add_upper_bound_constraints(len(objective2), parser.SignRestrictions, constraints2)
        display_canonical_form(parser.ProblemType, objective2, constraints2,
                               parser.SignRestrictions)
        s = RevisedPrimalSimplexSolver(objective2, constraints2, is_min, engine=engine)
        s.Solve()  # the C# lets the solver's exceptions escape (uncaught in Program.cs)
        write_full_results(out_path, "Revised Primal Simplex Algorithm (T-*)",
                           parser.ProblemType, objective2, constraints2, parser.SignRestrictions,
                           s.IterationSnapshots, s.FinalZ, s.SolutionVector)
        print("\nAll results have been saved to 'output_results.txt'.")
        return {"z": s.FinalZ, "x": s.SolutionVector, "solver": s}
    if choice == "3":  # Program.cs:356-415 (console output captured like TeeTextWriter does)
        buf = io.StringIO()

        class _Tee(io.TextIOBase):
            def write(self, t):
                sys.__stdout__.write(t)
                # print() sends its line terminator as a write of its own: that one is the
                # Console.WriteLine terminator (Environment.NewLine in the C#'s StringWriter);
                # a "\n" INSIDE a string stays what the C# literal holds
                buf.write(NL if t == "\n" else t)
                return len(t)

        with redirect_stdout(_Tee()):
            print("Solving with Branch and Bound Simplex Algorithm...")
            display_canonical_form(parser.ProblemType, parser.ObjectiveCoefficients,
                                   parser.Constraints, parser.SignRestrictions)
            _append_unit_bound_rows(parser)
            primal = PrimalSimplexSolver(parser.ObjectiveCoefficients, parser.Constraints,
                                         engine=engine, verbose=True)
            primal.Solve()
            x, z = BranchAndBoundAdapter.SolveFromPrimal(primal, enablePruning=False, isMin=False)
            print("\n=== Branch & Bound Result ===")
            print(f"Z* = {fmt._custom_0_hashes(z) if z == z and abs(z) != float('inf') else z}")
            for i, v in enumerate(x):
                print(f"x{i + 1} = {fmt._custom_0_hashes(v)}")
        write_snapshots_only(out_path, "Branch and Bound Simplex Algorithm", [buf.getvalue()], z,
                             x, append=False)
        print("----------------------------------------------")
        print("\nAll results have been saved to 'output_results.txt'.")
        return {"z": z, "x": x}
    if choice == "4":  # Program.cs:417-428: prints the canonical form and nothing else
        print("Solving with Cutting Plane Algorithm...")
        display_canonical_form(parser.ProblemType, parser.ObjectiveCoefficients,
                               parser.Constraints, parser.SignRestrictions)
        return {}
    print("Invalid choice. Please select a valid option (1-6).")
    return {}


def main(argv: Optional[Sequence[str]] = None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 2:
        print(__doc__)
        return 2
    parser = InputFileParser()
    parser.ReadInputFile(argv[0])
    if parser.ProblemType is None:
        print("Error reading file. Please ensure the file is formatted correctly and try again.")
        return 1
    run_option(parser, argv[1], argv[2] if len(argv) > 2 else "data/output_results.txt")
    return 0


if __name__ == "__main__":
    sys.exit(main())
