"""CPU tests of the host-side text plumbing: model-file parser (IO/InputFileParser.cs), .NET number
formatting used by the snapshots / result file.  PARITY UNPINNED: the reference commits no output
text; the expected strings below are .NET Framework's documented behaviour for these specifiers."""
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def pkg():
    import lpr_381_group_v22_amd as p
    return p


def test_parser_reads_the_sample_model(pkg):
    p = pkg.InputFileParser()
    p.ReadInputFile(os.path.join(HERE, "golden", "TextFile.txt"))
    assert p.ProblemType == "max"
    assert p.ObjectiveCoefficients == [2, 3, 3, 5, 2, 4]
    assert len(p.Constraints) == 1
    assert p.Constraints[0].Coefficients == [11, 8, 6, 14, 10, 10]
    assert p.Constraints[0].Relation == "<=" and p.Constraints[0].RHS == 40
    assert p.SignRestrictions == ["bin"] * 6


def test_parser_rejects_short_and_missing_files(pkg, tmp_path, capsys):
    p = pkg.InputFileParser()
    p.ReadInputFile(str(tmp_path / "nope.txt"))
    assert p.ProblemType is None and "can't find your file" in capsys.readouterr().out
    f = tmp_path / "short.txt"
    f.write_text("max 1 2\n1 1 <= 3\n")
    p.ReadInputFile(str(f))
    assert p.ProblemType is None and "not formatted correctly" in capsys.readouterr().out


def test_parser_relations_and_min(pkg, tmp_path):
    f = tmp_path / "m.txt"
    f.write_text("MIN +2 -3.5 +4\n+1 +2   +3 <= 10\n+3 +2 +1 >= 15\n1 0 0 = 2\n+ - urs\n")
    p = pkg.InputFileParser()
    p.ReadInputFile(str(f))
    assert p.ProblemType == "min" and p.ObjectiveCoefficients == [2.0, -3.5, 4.0]
    assert [c.Relation for c in p.Constraints] == ["<=", ">=", "="]
    assert p.Constraints[0].Coefficients == [1.0, 2.0, 3.0]
    assert p.SignRestrictions == ["+", "-", "urs"]


def test_dotnet_fixed_point_formatting():
    from lpr_381_group_v22_amd import table_iteration_formater as f
    assert f.F3(15.4) == "15.400" and f.F3(-0.0) == "0.000" and f.F3(-0.0004) == "0.000"
    assert f.F3(2.0005) == "2.001"      # 15-digit decimal first, then half away from zero
    assert f.F3(0.0005) == "0.001" and f.F3(-1.2345) == "-1.235" and f.F3(1e6) == "1000000.000"
    assert f.F6(15.4) == "15.400000"
    assert f.N3(15.399999999999999) == "15.4" and f.N3(0.19999999999999973) == "0.2"
    assert f.N3(1e-13) == "0" and f.N3(-2.0) == "-2" and f.N3(0.0005) == "0.001"
    assert f.N3(1234.5678) == "1234.568" and f.N3(-0.1234) == "-0.123"


def test_dotnet_general_formatting():
    from lpr_381_group_v22_amd.program import dotnet_double_to_string as g
    assert g(40.0) == "40" and g(0.5) == "0.5" and g(-11.0) == "-11" and g(1e-5) == "1E-05"
    assert g(1e20) == "1E+20" and g(0.1 + 0.2) == "0.3" and g(-0.0) == "0"


def test_canonical_form_text(pkg):
    from lpr_381_group_v22_amd.program import canonical_form_for_file
    p = pkg.InputFileParser()
    p.ReadInputFile(os.path.join(HERE, "golden", "TextFile.txt"))
    s = canonical_form_for_file(p.ProblemType, p.ObjectiveCoefficients, p.Constraints,
                                p.SignRestrictions)
    assert "Z -2x1 -3x2 -3x3 -5x4 -2x5 -4x6 = 0\n" in s
    assert "+ 11x1 + 8x2 + 6x3 + 14x4 + 10x5 + 10x6 + S1 = 40\n" in s
    assert "Sign Restrictions: x1: bin x2: bin" in s


def test_table_format_layout():
    import numpy as np
    from lpr_381_group_v22_amd import table_iteration_formater as f
    t = np.array([[-2.0, -3.0, 0.0, 0.0], [1.0, 2.0, 1.0, 10.0]])
    s = f.Format(t, 2, "Initial Tableau")
    lines = s.split("\r\n")
    assert lines[0] == "\nInitial Tableau:" and lines[1] == "-" * 80
    assert lines[2] == "Table\tx1\tx2\tt1\tRHS"
    assert lines[3] == "Z\t-2.000\t-3.000\t0.000\t0.000\t"
    assert lines[4] == "1\t1.000\t2.000\t1.000\t10.000\t"


def test_vectorised_node_scoring_equals_scalar_rules():
    """score_nodes (numpy) == IsInteger / CheckIntegerBasicVar applied value by value."""
    import numpy as np
    from lpr_381_group_v22_amd import branch_and_bound as bb
    rng = np.random.RandomState(0)
    vals = np.round(rng.uniform(-3, 9, size=(200, 7)), 4)
    vals[rng.rand(200, 7) < 0.5] = rng.randint(0, 3, size=(200, 7))[rng.rand(200, 7) < 0.5].mean()
    vals[5] = [0, 1, 2, 1, 0, 3, 1]
    vals[6] = [0.5, 1.5, 2.5, 0.25, 0.75, 0.50005, 0.49995]
    vals[7, 0] = 1e17
    all_int, var, value = bb.score_nodes(vals)
    for q in range(vals.shape[0]):
        k, v = bb.choose_branch(list(vals[q]))
        assert var[q] == k and (k < 0 or value[q] == v), q
        assert all_int[q] == all(bb._is_integer(t) for t in vals[q]), q
    xs = np.concatenate([rng.uniform(-50, 50, 500), [0.00005, 2.5, 3.5, -2.5, 1e16, 0.49999999999999994]])
    assert [bb._round4(float(x)) for x in xs] == bb._round4_np(xs).tolist()


def test_two_independent_number_formatters_agree():
    """Row f2: the product's {v:F3} / {v:F6} / double.ToString() mirrors (decimal-module based)
    against the test-side restatement (digit-string arithmetic, tests/ref_py_text.py) on ties,
    carries, zeros, huge / tiny magnitudes and 40 000 random doubles."""
    import numpy as np
    from lpr_381_group_v22_amd import table_iteration_formater as f
    from lpr_381_group_v22_amd.program import dotnet_double_to_string
    from ref_py_text import py_double_to_string, py_fixed
    special = [0.0, -0.0, 2.0005, 0.0005, -0.0004, -0.0005, 0.9995, 9.9995, 99.9995, 999.9995,
               -999.9995, 1e15, 1.5e15, 1e16, 123456789012345678.0, 1e-5, 9.99999e-6, 1e-4,
               1.23456789012345678, 15.4, 15.399999999999999, 0.1, 0.2, 0.30000000000000004,
               1e300, -1e300, 5e-324, 0.49999999999999994, 0.0049999999999999999, 0.0015,
               0.0025, 1.0005, 1.0015, 1.0025, float("inf"), float("-inf"), float("nan"),
               4.35, 4.345, 2.675, 1.005, 100.0, -100.0, 1234567.0004999, 0.00049999999999999]
    rng = np.random.RandomState(11)
    rnd = np.concatenate([rng.uniform(-1000, 1000, 20000), rng.uniform(-1, 1, 10000) ** 5,
                          np.round(rng.uniform(-50, 50, 10000), 4)])
    for v in special + rnd.tolist():
        assert f.F3(v) == py_fixed(v, 3), v
        assert f.F6(v) == py_fixed(v, 6), v
        assert dotnet_double_to_string(v) == py_double_to_string(v), v


def test_two_independent_table_formatters_agree():
    """TableIterationFormater.Format (:22-48): product mirror vs test-side restatement, with and
    without row labels, on a tableau holding ties and negative zeros."""
    import numpy as np
    from lpr_381_group_v22_amd import table_iteration_formater as f
    from ref_py_text import py_format_table
    rng = np.random.RandomState(3)
    T = np.round(rng.uniform(-20, 20, size=(5, 9)), 4)
    T[0, 0], T[1, 1], T[2, 2], T[3, 3] = -0.0, 2.0005, -0.0004, 1e7
    for nv in (2, 4, 8):
        assert f.Format(T, nv, "Before pivot") == py_format_table(T.tolist(), nv, "Before pivot")
    assert f.Format(T, 3, "x", ["a", "b"]) == py_format_table(T.tolist(), 3, "x", ["a", "b"])
