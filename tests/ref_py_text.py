"""TEST ONLY -- an independent restatement of the reference's text output for row f2:
  Utilities/TableIterationFormater.cs:22-48        Format (+ the {v:F3} specifier)
  Utilities/CanonicalFormConverter.cs:55-98        CanonicalFormForFile, FormatCoeff
  IO/OutputFileWrite.cs:16-119                     WriteFullResults, WriteSnapshotsOnly
  Simplex/PrimalSimplexSolver.cs:84-150,253-267    IterationSnapshots, console text, SolutionSummary
  Program.cs:356-415                               what option 3 captures around the solvers

Written without looking at lpr_381_group_v22_amd/{table_iteration_formater,program}.py's helpers:
numbers are formatted by digit-string arithmetic on the 15-significant-digit decimal image of the
double (the product uses the decimal module's quantize), the files are assembled from the C#'s
StringBuilder calls one by one.  PARITY UNPINNED by the reference (its output_results.txt is
empty): two independent readings of the C# agreeing byte for byte is what these tests can give.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

from ref_py import PyConstraint, PyPrimal, py_n3

CRLF = "\r\n"  # StringBuilder.AppendLine / Console.WriteLine on the reference's platform


# ---------------------------------------------------------------------------------------------
# number formatting of .NET Framework 4.7.2
# ---------------------------------------------------------------------------------------------
def _digits15(v: float):
    """(negative, digits, point): |v| = 0.d1 d2 ... d15 x 10^point, the 15-significant-digit decimal
    image the Framework's number formatter starts from (correctly rounded: C printf %.14e)."""
    s = "%.14e" % abs(v)
    mant, exp = s.split("e")
    digits = mant.replace(".", "")
    assert len(digits) == 15
    return (math.copysign(1.0, v) < 0), digits, int(exp) + 1


def py_fixed(v: float, decimals: int) -> str:
    """double.ToString("F<decimals>"): the 15-digit image rounded half AWAY at `decimals` places
    (NumberToString -> RoundNumber), no sign on a result that is all zeros."""
    if v != v:
        return "NaN"
    if math.isinf(v):
        return "Infinity" if v > 0 else "-Infinity"
    neg, digits, point = _digits15(v)
    # integer part = digits[:point] (padded), fraction = the rest
    if point > 0:
        ip = digits[:point] if point <= 15 else digits + "0" * (point - 15)
        fp = digits[point:] if point < 15 else ""
    else:
        ip = "0"
        fp = "0" * (-point) + digits
    keep, rest = fp[:decimals], fp[decimals:]
    keep = keep + "0" * (decimals - len(keep))
    number = list(ip + keep)
    if rest and rest[0] >= "5":  # RoundNumber: digit >= 5 rounds up, whatever follows
        k = len(number) - 1
        while k >= 0:
            if number[k] == "9":
                number[k] = "0"
                k -= 1
            else:
                number[k] = chr(ord(number[k]) + 1)
                break
        if k < 0:
            number.insert(0, "1")
    text = "".join(number)
    ip2, fp2 = (text[:len(text) - decimals], text[len(text) - decimals:]) if decimals else (text, "")
    ip2 = ip2.lstrip("0") or "0"
    nonzero = any(ch != "0" for ch in ip2 + fp2)
    out = ip2 + ("." + fp2 if decimals else "")
    return ("-" if neg and nonzero else "") + out


def py_double_to_string(v: float) -> str:
    """double.ToString() = "G": 15 significant digits, trailing zeros dropped, scientific when the
    decimal exponent is <= -5 or >= 15 (fixed only while -5 < exponent < 15: 0.0001 but 1E-05;
    "E+16": at least two exponent digits)."""
    if v != v:
        return "NaN"
    if math.isinf(v):
        return "Infinity" if v > 0 else "-Infinity"
    if v == 0:
        return "0"
    neg, digits, point = _digits15(v)
    digits = digits.rstrip("0") or "0"
    exp10 = point - 1
    if exp10 <= -5 or exp10 >= 15:
        mant = digits[0] + ("." + digits[1:] if len(digits) > 1 else "")
        body = mant + "E" + ("+" if exp10 >= 0 else "-") + "%02d" % abs(exp10)
    elif point <= 0:
        body = "0." + "0" * (-point) + digits
    elif len(digits) <= point:
        body = digits + "0" * (point - len(digits))
    else:
        body = digits[:point] + "." + digits[point:]
    return ("-" if neg else "") + body


# ---------------------------------------------------------------------------------------------
# TableIterationFormater.Format (:22-48)
# ---------------------------------------------------------------------------------------------
def py_format_table(tab: Sequence[Sequence[float]], num_original_vars: int, title: str,
                    row_labels: Optional[Sequence[str]] = None) -> str:
    rows, cols = len(tab), len(tab[0])
    sb: List[str] = []
    sb.append("\n" + title + ":" + CRLF)                       # :25 AppendLine($"\n{title}:")
    sb.append("-" * 80 + CRLF)                                  # :26
    sb.append("Table\t")                                        # :31
    for j in range(num_original_vars):                          # :32
        sb.append("x%d\t" % (j + 1))
    for j in range(num_original_vars, cols - 1):                # :33
        sb.append("t%d\t" % (j - num_original_vars + 1))
    sb.append("RHS" + CRLF)                                     # :34
    sb.append("Z\t")                                            # :36
    for j in range(cols):                                       # :37
        sb.append(py_fixed(tab[0][j], 3) + "\t")
    sb.append(CRLF)                                             # :38
    for i in range(1, rows):                                    # :40-46
        label = row_labels[i - 1] if (row_labels is not None and len(row_labels) >= i) else str(i)
        sb.append(label + "\t")
        for j in range(cols):
            sb.append(py_fixed(tab[i][j], 3) + "\t")
        sb.append(CRLF)
    return "".join(sb)


# ---------------------------------------------------------------------------------------------
# CanonicalFormConverter (:15-98)
# ---------------------------------------------------------------------------------------------
def _coeff(c: float) -> str:  # FormatCoeff :95-98
    return "+ " + py_double_to_string(c) if c >= 0 else py_double_to_string(c)


def py_canonical_form_for_file(problem_type: str, objective: Sequence[float],
                               constraints: Sequence[PyConstraint], signs: Sequence[str]) -> str:
    sb = ["\n=== Canonical Form ===" + CRLF, "Z "]              # :60-61
    for i, c in enumerate(objective):                           # :62-66
        sb.append(_coeff(c * -1) + "x%d " % (i + 1))
    sb.append("= 0\n")                                          # :67
    for i, con in enumerate(constraints):                       # :70-82
        for j, a in enumerate(con.Coefficients):
            sb.append(_coeff(a) + "x%d " % (j + 1))
        sb.append("+ S%d " % (i + 1))
        sb.append("= " + py_double_to_string(con.RHS) + "\n")
    sb.append("\nSign Restrictions: ")                          # :85
    for i, s in enumerate(signs):                               # :86-89
        sb.append("x%d: %s " % (i + 1, s))
    sb.append("\n======================\n" + CRLF)              # :90 AppendLine
    return "".join(sb)


def py_canonical_form_console(problem_type: str, objective: Sequence[float],
                              constraints: Sequence[PyConstraint], signs: Sequence[str]) -> str:
    """DisplayCanonicalForm :15-52 as the text it sends to Console.Out."""
    sb = ["\n=== Canonical Form ===" + CRLF, problem_type.upper() + " Z "]
    for i, c in enumerate(objective):
        sb.append(_coeff(c * -1) + "x%d " % (i + 1))
    sb.append("= 0\n" + CRLF)
    for i, con in enumerate(constraints):
        for j, a in enumerate(con.Coefficients):
            sb.append(_coeff(a) + "x%d " % (j + 1))
        sb.append("+ S%d " % (i + 1))
        sb.append("= " + py_double_to_string(con.RHS) + CRLF)
    sb.append(CRLF)
    sb.append("Sign Restrictions: ")
    for i, s in enumerate(signs):
        sb.append("x%d: %s " % (i + 1, s))
    sb.append("\n======================\n" + CRLF)
    return "".join(sb)


# ---------------------------------------------------------------------------------------------
# PrimalSimplexSolver: IterationSnapshots and console text (:84-150, :253-267)
# ---------------------------------------------------------------------------------------------
class PyPrimalText(PyPrimal):
    """PyPrimal + what Solve() appends to IterationSnapshots and writes to the console."""

    def col_label(self, col: int) -> str:  # :253-254
        return "x%d" % (col + 1) if col < self.n else "t%d" % (col - self.n + 1)

    def solution_summary(self, title: str = "Optimal solution") -> str:  # :256-267
        sb = [title + ":" + CRLF, "Z = " + py_fixed(self.FinalZ, 6) + CRLF]
        if self.SolutionVector is not None:
            for i in range(self.n):
                sb.append("x%d = %s%s" % (i + 1, py_fixed(self.SolutionVector[i], 6), CRLF))
        return "".join(sb)

    def solve_text(self):
        """Returns (IterationSnapshots, console text)."""
        snaps = [py_format_table(self.t, self.n, "Initial Tableau")]  # ctor :86
        con: List[str] = []
        it = 0
        while True:
            e = self.find_entering()
            if e == -1:  # :110-126
                self.status = "optimal"
                self.FinalZ = self.t[0][-1]
                self.SolutionVector = self.extract_solution()
                con.append("Optimal Solution Found!" + CRLF)
                block = py_format_table(self.t, self.n, "Final Tableau (Optimal)") + CRLF \
                    + self.solution_summary() + CRLF
                snaps.append(block)
                con.append(self.solution_summary() + CRLF)
                con.append("-" * 100 + CRLF)
                break
            r = self.find_leaving(e)
            if r == -1:  # :129-135
                self.status = "unbounded"
                con.append("Unbounded Solution!" + CRLF)
                snaps.append(py_format_table(self.t, self.n, "Unbounded Tableau"))
                break
            it += 1
            con.append("\nIteration %d: pivot @ constraint %d, column %s%s"
                       % (it, r, self.col_label(e), CRLF))                       # :138
            con.append(py_format_table(self.t, self.n, "Before pivot") + CRLF)   # :139
            self.log.append((r, e))
            self.pivot(r, e)
            self.basic[r - 1] = e
            con.append("After pivot (constraint %d, column %s):%s" % (r, self.col_label(e), CRLF))
            con.append(py_format_table(self.t, self.n, "After pivot") + CRLF)    # :146
            snaps.append(py_format_table(self.t, self.n, "Iteration %d - After pivot" % it))
        return snaps, "".join(con)


# ---------------------------------------------------------------------------------------------
# OutputFileWrite (:16-119); the files start with a UTF-8 BOM (File.WriteAllText(.., Encoding.UTF8))
# ---------------------------------------------------------------------------------------------
BOM = b"\xef\xbb\xbf"
TIMESTAMP = "Timestamp: <masked>"


def py_write_full_results(solver_used: str, problem_type: str, objective, constraints, signs,
                          snapshots: Optional[Sequence[str]], final_z: float,
                          solution: Optional[Sequence[float]]) -> bytes:
    sb = ["=" * 60 + CRLF, "Solver: " + solver_used + CRLF, "Problem type: " + problem_type + CRLF,
          TIMESTAMP + CRLF, "=" * 60 + CRLF]                                      # :31-37
    sb.append(py_canonical_form_for_file(problem_type, objective, constraints, signs))  # :40-52
    if snapshots is not None and len(snapshots) > 0:                             # :55-64
        sb.append("=== Iteration Snapshots ===" + CRLF)
        for i, s in enumerate(snapshots):
            sb.append("--- Iteration %d ---%s" % (i + 1, CRLF))
            sb.append(s + CRLF)
        sb.append(CRLF)
    sb.append("=== Final Results ===" + CRLF)                                     # :67-68
    sb.append("Z* = " + py_n3(final_z) + CRLF)
    if solution is not None and len(solution) > 0:                                # :70-74
        for i, v in enumerate(solution):
            sb.append("x%d = %s%s" % (i + 1, py_n3(v), CRLF))
    return BOM + "".join(sb).encode("utf-8")


def py_write_snapshots_only(solver_used: str, snapshots: Optional[Sequence[str]], final_z: float,
                            solution: Optional[Sequence[float]]) -> bytes:
    sb = ["=" * 60 + CRLF, "Solver: " + solver_used + CRLF, TIMESTAMP + CRLF, "=" * 60 + CRLF]
    if snapshots is not None and len(snapshots) > 0:                              # :98-106
        sb.append("=== Solver Log ===" + CRLF)
        for s in snapshots:
            sb.append(s + CRLF)
            if not s.endswith("\n"):
                sb.append(CRLF)
    sb.append("=== Final Results ===" + CRLF)
    sb.append("Z* = " + py_n3(final_z) + CRLF)
    if solution is not None and len(solution) > 0:
        for i, v in enumerate(solution):
            sb.append("x%d = %s%s" % (i + 1, py_n3(v), CRLF))
    return BOM + "".join(sb).encode("utf-8")


def mask_timestamp(data: bytes) -> bytes:
    import re
    return re.sub(rb"Timestamp: [^\r\n]*", TIMESTAMP.encode(), data)
