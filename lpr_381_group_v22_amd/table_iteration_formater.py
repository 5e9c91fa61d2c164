"""Host-side mirror of Utilities/TableIterationFormater.cs:19-48 and of the .NET Framework
number formatting it relies on (``{v:F3}``), plus ``NumFormat.N3``
(Simplex/RevisedPrimalSimplexSolver.cs:451-466).

.NET Framework 4.7.2 formats a double with a fixed-point specifier in two steps: the value is
first converted to 15 significant decimal digits (correctly rounded), then that decimal string is
rounded half-away-from-zero to the requested number of decimals; a result that rounds to zero is
printed without a sign.  ``'%.3f'`` differs (it rounds the exact binary value half-even: 2.0005 ->
'2.000' where .NET prints '2.001'), so the two steps are restated with ``decimal``.
PARITY UNPINNED: the reference holds no formatted output to check these strings against.
"""
from __future__ import annotations

import math
from decimal import ROUND_HALF_UP, Context, Decimal
from typing import Optional, Sequence

import numpy as np

NEWLINE = "\r\n"  # Environment.NewLine on the reference's platform (Windows / .NET Framework)


def _dec15(v: float) -> Decimal:
    return Decimal("%.14e" % v)  # 15 significant digits, correctly rounded


def format_fixed(v: float, decimals: int) -> str:
    """double.ToString("F<decimals>", InvariantCulture) on .NET Framework."""
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "Infinity" if v > 0 else "-Infinity"
    q = Decimal(1).scaleb(-decimals)
    # (a context of its own: the default 28 digits cannot hold 1e300 at three decimals)
    d = _dec15(v).quantize(q, rounding=ROUND_HALF_UP, context=Context(prec=700))
    if d == 0:
        d = abs(d)  # "-0.000" never appears on .NET Framework
    return format(d, "f")


def F3(v: float) -> str:
    return format_fixed(v, 3)


def F6(v: float) -> str:
    return format_fixed(v, 6)


def dotnet_double_to_string(v: float) -> str:
    """double.ToString() on .NET Framework ("G", 15 significant digits)."""
    if v != v:
        return "NaN"
    if v in (float("inf"), float("-inf")):
        return "Infinity" if v > 0 else "-Infinity"
    s = format(v, ".15g")
    if "e" in s:
        mant, exp = s.split("e")
        sign = "+" if int(exp) >= 0 else "-"
        s = f"{mant}E{sign}{abs(int(exp)):02d}"
    return "0" if s in ("-0", "0") else s


def dotnet_round_half_even(x: float) -> float:
    """Math.Round(double) on .NET Framework (classlibnative COMDouble::Round)."""
    if math.isnan(x) or math.isinf(x):
        return x
    if abs(x) < 9.2e18 and x == float(int(x)):
        return x
    t = x + 0.5
    f = math.floor(t)
    if f == t and math.fmod(t, 2.0) != 0:
        f -= 1.0
    return math.copysign(f, x)


def dotnet_round_digits(x: float, digits: int, away_from_zero: bool = False) -> float:
    """Math.Round(double, int[, MidpointRounding]) on .NET Framework (Math.InternalRound)."""
    if math.isnan(x) or math.isinf(x):
        return x
    if abs(x) < 1e16:
        p = float(10 ** digits)
        x = x * p
        if away_from_zero:
            frac, whole = math.modf(x)
            x = whole
            if abs(frac) >= 0.5:
                x += math.copysign(1.0, frac)
        else:
            x = dotnet_round_half_even(x)
        x = x / p
    return x


def _custom_0_hashes(r: float) -> str:
    """double.ToString("0.###", InvariantCulture): up to 3 decimals, trailing zeros dropped."""
    d = _dec15(r).quantize(Decimal("0.001"), rounding=ROUND_HALF_UP)
    neg = d < 0
    s = format(abs(d), "f")
    if "." in s:
        s = s.rstrip("0").rstrip(".")
    if s == "":
        s = "0"
    if s == "0":
        neg = False
    return ("-" if neg else "") + s


def N3(x: float) -> str:
    """NumFormat.N3, Simplex/RevisedPrimalSimplexSolver.cs:455-465."""
    if math.isnan(x):
        return "NaN"
    if abs(x) < 1e-12:
        x = 0.0
    if math.isinf(x):
        return "Infinity" if x > 0 else "-Infinity"
    r = dotnet_round_digits(x, 3, away_from_zero=True)
    rr = dotnet_round_half_even(r)
    if abs(r - rr) < 1e-12:
        # double.ToString() of an integral value ("R"-less general format, 15 digits)
        if rr == 0:
            return "0"
        if abs(rr) < 1e15:
            return str(int(rr))
        return _dotnet_g15_scientific(rr)
    return _custom_0_hashes(r)


def _dotnet_g15_scientific(v: float) -> str:
    """double.ToString() on .NET Framework for |v| >= 1e15: "G" with 15 significant digits falls
    back to scientific notation, mantissa without trailing zeros, exponent sign and at least two
    digits ("1E+15", "1.23456789012346E+17")."""
    mant, exp = f"{abs(v):.14e}".split("e")
    mant = mant.rstrip("0").rstrip(".")
    e = int(exp)
    return ("-" if v < 0 else "") + f"{mant}E{'+' if e >= 0 else '-'}{abs(e):02d}"


def Format(tab: np.ndarray, numOriginalVars: int, title: str,
           rowLabels: Optional[Sequence[str]] = None) -> str:
    """TableIterationFormater.Format, Utilities/TableIterationFormater.cs:22-48."""
    rows, cols = tab.shape
    nl = NEWLINE
    out = ["\n" + title + ":" + nl, "-" * 80 + nl, "Table\t"]
    for j in range(numOriginalVars):
        out.append(f"x{j + 1}\t")
    for j in range(numOriginalVars, cols - 1):
        out.append(f"t{j - numOriginalVars + 1}\t")
    out.append("RHS" + nl)
    out.append("Z\t")
    for j in range(cols):
        out.append(F3(float(tab[0, j])) + "\t")
    out.append(nl)
    for i in range(1, rows):
        label = rowLabels[i - 1] if (rowLabels is not None and len(rowLabels) >= i) else f"{i}"
        out.append(label + "\t")
        for j in range(cols):
            out.append(F3(float(tab[i, j])) + "\t")
        out.append(nl)
    return "".join(out)
