"""GPU test of the solver-selection surface (Program.cs options 1-3) on the reference's sample
model: same text file in, same numbers and result-file layout out."""
import os
import shutil

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def load(pkg):
    p = pkg.InputFileParser()
    p.ReadInputFile(os.path.join(HERE, "golden", "TextFile.txt"))
    return p


def test_option1_primal(engine, tmp_path):
    import lpr_381_group_v22_amd as pkg
    from lpr_381_group_v22_amd.program import run_option
    p = load(pkg)
    out = str(tmp_path / "data" / "output_results.txt")
    r = run_option(p, "1", out, engine=engine)
    assert r["z"] == 15.4 and len(p.Constraints) == 7  # the parser's own list grew (Program.cs:123)
    text = open(out, encoding="utf-8-sig", newline="").read()
    assert "Solver: Primal Simplex Algorithm" in text and "Problem type: max" in text
    assert "=== Canonical Form ===" in text and "=== Iteration Snapshots ===" in text
    assert "--- Iteration 8 ---" in text  # initial + 6 pivots + final block
    assert "Z* = 15.4\r\n" in text and "x5 = 0.2\r\n" in text and "x1 = 0\r\n" in text
    assert "Final Tableau (Optimal):" in text and "Z = 15.400000" in text
    # choosing the option again appends the bound rows again (reference behaviour)
    run_option(p, "1", out, engine=engine)
    assert len(p.Constraints) == 13


def test_option2_revised(engine, tmp_path):
    import lpr_381_group_v22_amd as pkg
    from lpr_381_group_v22_amd.program import run_option
    p = load(pkg)
    out = str(tmp_path / "output_results.txt")
    r = run_option(p, "2", out, engine=engine)
    assert r["z"] == 15.399999999999999 and len(p.Constraints) == 1  # option 2 works on copies
    text = open(out, encoding="utf-8-sig", newline="").read()
    assert "Solver: Revised Primal Simplex Algorithm (T-*)" in text
    assert "Z* = 15.4\r\n" in text and "x5 = 0.2\r\n" in text


def test_option3_branch_and_bound(engine, tmp_path):
    import lpr_381_group_v22_amd as pkg
    from lpr_381_group_v22_amd.program import run_option
    p = load(pkg)
    out = str(tmp_path / "output_results.txt")
    r = run_option(p, "3", out, engine=engine)
    assert r["z"] == 15.0 and r["x"] == [0.0, 1.0, 1.0, 1.0, 0.0, 1.0]
    text = open(out, encoding="utf-8-sig", newline="").read()
    assert "Solver: Branch and Bound Simplex Algorithm" in text and "=== Solver Log ===" in text
    assert "=== Branch & Bound Result ===" in text
    assert "Z* = 15\r\n" in text and "x2 = 1\r\n" in text and "x1 = 0\r\n" in text


def test_min_problem_redirect(engine, tmp_path, capsys):
    import lpr_381_group_v22_amd as pkg
    from lpr_381_group_v22_amd.program import run_option
    f = tmp_path / "min.txt"
    f.write_text("min +3 +2\n+1 +1 <= 4\n+ +\n")
    p = pkg.InputFileParser()
    p.ReadInputFile(str(f))
    r = run_option(p, "1", str(tmp_path / "o.txt"), engine=engine)  # Program.cs:92-99
    assert r == {"skipped": True}
    assert "Please use Option 2" in capsys.readouterr().out
    r = run_option(p, "2", str(tmp_path / "o.txt"), engine=engine)
    assert r["z"] == 0.0 and r["x"] == [0.0, 0.0]


MODELS = {
    "sample": None,  # tests/golden/TextFile.txt, the reference's data/TextFile.txt
    "readme_ge_row": "max +2 +3 +4\n+1 +2 +3 <= 10\n+3 +2 +1 >= 15\n+ + +\n",
    "fractions": "max +2.5 +3.125 +0.75 +4.0005\n+1.5 +2.25 +3.3 +0.7 <= 10.45\n"
                 "+0.3 +0.2 +1.7 +2.9 <= 7.0005\n+ + + +\n",
    "unit_rows_bound_it": "max +1 +1\n+1 -1 <= 2\n+ +\n",  # unbounded without Program.cs:114-124
}


def _model_path(name, tmp_path):
    if MODELS[name] is None:
        return os.path.join(HERE, "golden", "TextFile.txt")
    f = tmp_path / f"{name}.txt"
    f.write_text(MODELS[name])
    return str(f)


@pytest.mark.parametrize("name", list(MODELS))
def test_option1_result_file_byte_for_byte(engine, tmp_path, name):
    """Row f2, option 1 (Program.cs:91-140): the whole data/output_results.txt -- header, canonical
    form (CanonicalFormConverter.cs:55-93), every IterationSnapshot through
    TableIterationFormater.Format / {v:F3} (:22-48), the final block with SolutionSummary's {v:F6},
    Z* / x_i through NumFormat.N3 (OutputFileWrite.cs:16-78) -- against the file assembled by the
    INDEPENDENT restatement tests/ref_py_text.py (its own primal loop, its own number formatting;
    nothing of the product is used to build the expected bytes).  Timestamp masked.  The reference
    commits no output file, so text parity stays unpinned by the reference itself."""
    import lpr_381_group_v22_amd as pkg
    from lpr_381_group_v22_amd.program import run_option
    from ref_py import parse_model_text, program_option1_constraints
    from ref_py_text import PyPrimalText, mask_timestamp, py_write_full_results
    path = _model_path(name, tmp_path)
    p = pkg.InputFileParser()
    p.ReadInputFile(path)
    out = str(tmp_path / "data" / "output_results.txt")
    run_option(p, "1", out, engine=engine)
    got = open(out, "rb").read()
    ptype, obj, cons, signs = parse_model_text(open(path).read())
    cons1 = program_option1_constraints(len(obj), cons)  # Program.cs:114-124 (parser's own list)
    ref = PyPrimalText(obj, cons1, True)
    snaps, _console = ref.solve_text()
    want = py_write_full_results("Primal Simplex Algorithm", ptype, obj, cons1, signs, snaps,
                                 ref.FinalZ, ref.SolutionVector)
    assert mask_timestamp(got) == want
    if name == "sample":
        assert len(snaps) == 8 and b"Iteration 6 - After pivot" in got
    if name == "unit_rows_bound_it":
        assert ref.status == "optimal" and b"Z* = 2\r\n" in got


def test_option3_result_file_byte_for_byte(engine, tmp_path):
    """Option 3 (Program.cs:356-415) writes ONE captured console text through WriteSnapshotsOnly
    (OutputFileWrite.cs:83-119): the banner, DisplayCanonicalForm, the primal solve's Before / After
    pivot tables, then everything ExecuteBranchAndBound (BranchBoundSimplexSolver.cs:1006-1233)
    writes -- node headers, branching lines, "pivot @ constraint r, column c" of every pivot, every
    tableau of every child through DisplayTableau -- and the result block.  The whole file, byte for
    byte (timestamp masked), against the independent restatement: tests/ref_py_bb.py (the search),
    tests/ref_py_bb_text.py (what it prints), tests/ref_py_text.py (number / table formatting, the
    file).  The numbers of the product side come from the device (lpr_bb_node_info,
    lpr_bb_expand_traced).  No branch of this model fails, so the one thing that cannot be
    reproduced (`failed: {e}`, a .NET stack trace) does not occur."""
    import lpr_381_group_v22_amd as pkg
    from lpr_381_group_v22_amd.program import run_option
    from ref_py import parse_model_text, program_option1_constraints
    from ref_py_bb_text import NarratedBranchAndBound
    from ref_py_text import (CRLF, PyPrimalText, mask_timestamp, py_canonical_form_console,
                             py_double_to_string, py_write_snapshots_only)
    from ref_py import py_n3
    path = os.path.join(HERE, "golden", "TextFile.txt")
    p = pkg.InputFileParser()
    p.ReadInputFile(path)
    out = str(tmp_path / "output_results.txt")
    r = run_option(p, "3", out, engine=engine)
    got = mask_timestamp(open(out, "rb").read())
    ptype, obj, cons, signs = parse_model_text(open(path).read())
    ref = PyPrimalText(obj, program_option1_constraints(len(obj), cons), True)
    _snaps, console = ref.solve_text()
    bb = NarratedBranchAndBound(len(obj))
    res = bb.ExecuteNarrated(ref.t)
    assert "failed:" not in res["text"] and res["processed"] == 20
    x_bb = res["x"] if res["x"] is not None else []

    def f0(v):  # {v:0.###}
        return py_n3(v) if abs(v) >= 1e-12 or v == 0 else py_n3(v)

    text = ("Solving with Branch and Bound Simplex Algorithm..." + CRLF
            + py_canonical_form_console(ptype, obj, cons, signs) + console + res["text"]
            + "\n=== Branch & Bound Result ===" + CRLF + "Z* = " + f0(res["z"]) + CRLF
            + "".join(f"x{i + 1} = {f0(v)}" + CRLF for i, v in enumerate(x_bb)))
    want = py_write_snapshots_only("Branch and Bound Simplex Algorithm", [text], res["z"], x_bb)
    assert r["z"] == res["z"] and r["x"] == x_bb
    if got != want:  # show where the two texts part
        k = next(i for i in range(min(len(got), len(want))) if got[i] != want[i])
        raise AssertionError(f"first difference at byte {k}: got {got[max(0, k - 120):k + 80]!r} "
                             f"want {want[max(0, k - 120):k + 80]!r}")
    assert b"Total branchs processed: 20" in got and b"Optimal branch: 1.1" in got


@pytest.mark.parametrize("name", ["binary_4v1c_s0", "frac_4v2c_s10", "binary_8v3c_s3",
                                  "binary_6v2c_s2"])
def test_narrated_branch_and_bound_equals_the_batched_one_and_the_restatement(engine, oracle,
                                                                            name, capsys):
    """The host-driven, narrating form of ExecuteBranchAndBound (lpr_bb_node_info +
    lpr_bb_expand_traced) against lpr_bb_run (same x, z) and against the restated console text,
    on instances with infeasible children, primal-phase pivots and a dropped last tableau."""
    import bb_cases
    import lp_cases
    import lpr_381_group_v22_amd as pkg
    from ref_py import PyPrimal
    from ref_py_bb_text import NarratedBranchAndBound
    from ref_py_text import CRLF
    obj, cons = dict(bb_cases.all_bb_cases())[name]
    s = pkg.PrimalSimplexSolver(obj, [pkg.Constraint(list(c.Coefficients), c.Relation, c.RHS)
                                      for c in cons], True, engine=engine, snapshots="none")
    s.Solve()
    x0, z0 = pkg.BranchAndBoundAdapter.SolveFromPrimal(s, narrate=False)
    capsys.readouterr()
    x1, z1 = pkg.BranchAndBoundAdapter.SolveFromPrimal(s, narrate=True)
    text = capsys.readouterr().out
    assert x1 == x0 and (z1 == z0 or (z1 != z1 and z0 != z0))
    p = PyPrimal(obj, cons, True)
    p.solve()
    bb = NarratedBranchAndBound(len(obj))
    res = bb.ExecuteNarrated(p.t)
    assert (res["x"] or []) == x1
    # print() ends lines with "\n" where Console.WriteLine writes Environment.NewLine
    assert text.replace(CRLF, "\n") == res["text"].replace(CRLF, "\n")


def test_option2_result_file_byte_for_byte(engine, tmp_path):
    """The whole data/output_results.txt of option 2 on the reference's sample model against the
    bytes assembled by the independent restatement (tests/ref_py.py: PyRevised captures +
    CaptureSnapshot's text + NumFormat.N3; tests/ref_py_text.py: WriteFullResults + the canonical
    form -- nothing of the product builds the expected file), timestamp line masked.  The reference commits no output
    file (data/output_results.txt is empty), so this pins the device path and the host mirror
    against the second restatement, not against the C# itself: text parity unpinned."""
    import re
    import lpr_381_group_v22_amd as pkg
    from lpr_381_group_v22_amd.program import run_option
    from ref_py import (PyRevised, parse_model_text, program_option2_constraints,
                        py_snapshot_text)
    import lp_cases
    p = load(pkg)
    out = str(tmp_path / "output_results.txt")
    run_option(p, "2", out, engine=engine)
    got = open(out, "rb").read()
    ptype, obj, cons, signs = parse_model_text(lp_cases.SAMPLE_MODEL)
    cons2 = program_option2_constraints(len(obj), signs, cons)
    ref = PyRevised(obj, cons2, ptype == "min")
    assert ref.solve(capture=True) == "optimal"
    snaps = [py_snapshot_text(s, ref.n, ref.m, ptype == "min") for s in ref.snapshots]
    from ref_py_text import py_write_full_results
    want = py_write_full_results("Revised Primal Simplex Algorithm (T-*)", ptype, obj, cons2, signs,
                                 snaps, ref.FinalZ, ref.SolutionVector)
    from ref_py_text import mask_timestamp
    assert len(ref.snapshots) == 7 and b"--- Iteration 7 ---" in got
    assert mask_timestamp(got) == want


def test_unbounded_snapshots_and_console_text(engine, capsys):
    """PrimalSimplexSolver.Solve's unbounded exit (:129-135): "Unbounded Solution!", FinalTableau
    kept, an "Unbounded Tableau" snapshot -- IterationSnapshots and the console text of the mirror
    class against the independent restatement (tests/ref_py_text.py)."""
    import lpr_381_group_v22_amd as pkg
    from ref_py import PyConstraint
    from ref_py_text import CRLF, PyPrimalText
    obj = [1.0, 2.0, 0.5]
    cons = [([1.0, -1.0, 0.25], "<=", 2.0), ([-2.0, 0.0, 1.0], "<=", 3.5)]
    ref = PyPrimalText(obj, [PyConstraint(list(a), r, b) for a, r, b in cons], True)
    snaps, console = ref.solve_text()
    assert ref.status == "unbounded" and "Unbounded Tableau" in snaps[-1]
    s = pkg.PrimalSimplexSolver(obj, [pkg.Constraint(list(a), r, b) for a, r, b in cons], True,
                                engine=engine, verbose=True, snapshots="all")
    capsys.readouterr()
    s.Solve()
    out = capsys.readouterr().out
    assert s.IterationSnapshots == snaps
    assert s.SolutionVector is None and s.FinalZ == 0.0 and s.FinalTableau is not None
    # print() terminates lines with "\n" where Console.WriteLine writes Environment.NewLine
    assert out.replace(CRLF, "\n") == console.replace(CRLF, "\n")
