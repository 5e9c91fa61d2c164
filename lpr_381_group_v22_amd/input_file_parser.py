"""The reference's text-file model format (host-side mirror of IO/InputFileParser.cs:10-84).

    line 0          ``max|min c1 c2 ...``            (:36-43, split on single spaces)
    lines 1..L-2    ``a1 a2 ... rel rhs``            (:45-62, empty entries removed)
    last line       sign restrictions, one token per variable (:64-65)

Member names follow the C# (ProblemType, ObjectiveCoefficients, Constraints, SignRestrictions,
ReadInputFile) so call sites read like Program.cs.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional


def _parse_double(tok: str) -> float:
    # double.Parse(..., InvariantCulture) accepts a leading '+', surrounding white space,
    # thousands separators and exponents; the model files only use signed decimals.
    return float(tok.replace(",", ""))


@dataclass
class Constraint:
    """IO/InputFileParser.cs:70-82."""
    Coefficients: List[float]
    Relation: str
    RHS: float


@dataclass
class InputFileParser:
    ProblemType: Optional[str] = None
    ObjectiveCoefficients: List[float] = field(default_factory=list)
    Constraints: List[Constraint] = field(default_factory=list)
    SignRestrictions: List[str] = field(default_factory=list)

    def ReadInputFile(self, filePath: str) -> None:
        if not os.path.exists(filePath):  # :21-25
            print("Sorry, we can't find your file, please check it's in the right folser")
            return
        with open(filePath, "r", encoding="utf-8-sig") as f:
            linesInFile = f.read().splitlines()  # File.ReadAllLines
        if len(linesInFile) < 3:  # :30-34
            print("The input file is not formatted correctly.")
            return

        objectiveLine = linesInFile[0].strip().split(" ")  # :36 (no RemoveEmptyEntries here)
        self.ProblemType = objectiveLine[0].lower()
        for tok in objectiveLine[1:]:
            self.ObjectiveCoefficients.append(_parse_double(tok))  # :39-43 (throws on "")

        n = len(self.ObjectiveCoefficients)
        for i in range(1, len(linesInFile) - 1):  # :45-62
            parts = [p for p in linesInFile[i].strip().split(" ") if p != ""]
            coeffs = [_parse_double(parts[j]) for j in range(n)]
            relation = parts[n]
            rhs = _parse_double(parts[n + 1])
            self.Constraints.append(Constraint(coeffs, relation, rhs))

        self.SignRestrictions.extend(linesInFile[-1].strip().split(" "))  # :64-65
        print("Your file was read and is in the correct format!")
