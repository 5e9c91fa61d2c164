// GpuSolvers.cs -- drop-in bodies for the three solver classes of LPR_381_Group_V22, calling the
// MI355X engine through NativeMethods.  UNVERIFIED (never compiled: no C# toolchain in the build
// image).  Public members, argument meaning and error behaviour are those of the reference classes
// so that Program.cs (cases "1", "2", "3") compiles unchanged when these replace
// Simplex/PrimalSimplexSolver.cs, Simplex/RevisedPrimalSimplexSolver.cs and
// IntegerProgramming/BranchAndBoundAdapter.cs.
using System;
using System.Collections.Generic;
using System.Linq;
using LPR_381_Group_V22.Native;
using LPR_381_Group_V22.Utilities;
using IOConstraint = LPR_381_Group_V22.IO.InputFileParser.Constraint;

namespace LPR_381_Group_V22.Simplex
{
    public class PrimalSimplexSolver : IDisposable
    {
        private readonly int numVariables, numConstraints;
        internal IntPtr Tableau;                       // lpr_tableau*
        public List<string> IterationSnapshots = new List<string>();
        public double FinalZ { get; private set; }
        public List<double> SolutionVector { get; private set; }
        public double[,] FinalTableau { get; private set; }
        /// <summary>Snapshots are O(R*C) text each; off above this many elements.</summary>
        public static int SnapshotElementLimit = 4096;

        public PrimalSimplexSolver(List<double> objective, List<IOConstraint> constraints, bool isMaximization = true)
        {
            numVariables = objective.Count;
            numConstraints = constraints.Count;
            int n = numVariables, m = numConstraints;
            var A = new double[Math.Max(1, m * n)];
            var ncoef = new int[Math.Max(1, m)];
            var rel = new sbyte[Math.Max(1, m)];
            var rhs = new double[Math.Max(1, m)];
            for (int i = 0; i < m; i++)
            {
                int k = Math.Min(n, constraints[i].Coefficients.Count);   // PrimalSimplexSolver.cs:68-72
                for (int j = 0; j < k; j++) A[i * n + j] = constraints[i].Coefficients[j];
                ncoef[i] = k;
                rel[i] = (sbyte)(constraints[i].Relation == ">=" ? 1 : constraints[i].Relation == "=" ? 2 : 0);
                rhs[i] = constraints[i].RHS;
            }
            NativeMethods.ThrowIfError(NativeMethods.lpr_tableau_from_lp(Engine.Handle, n, m, objective.ToArray(), A, n,
                ncoef, rel, rhs, isMaximization ? 1 : 0, out Tableau), "lpr_tableau_from_lp");
            if ((long)(m + 1) * (n + m + 1) <= SnapshotElementLimit) CaptureSnapshot("Initial Tableau");
        }

        public void Solve()
        {
            var opts = new LprSolveOpts();
            NativeMethods.ThrowIfError(NativeMethods.lpr_primal_solve(Tableau, ref opts, out var res), "lpr_primal_solve");
            if (res.status == (int)LprStatus.Optimal)
            {
                var x = new double[Math.Max(1, numVariables)];
                NativeMethods.lpr_extract_solution(Tableau, numVariables, x, out double z);
                FinalZ = z;                                            // :113
                SolutionVector = x.Take(numVariables).ToList();        // :114
                FinalTableau = GetFinalTableau();                      // :116
                Console.WriteLine("Optimal Solution Found!");
            }
            else if (res.status == (int)LprStatus.Unbounded)
            {
                Console.WriteLine("Unbounded Solution!");             // :131, FinalZ stays 0, SolutionVector null
                FinalTableau = GetFinalTableau();
            }
        }

        public double[,] GetFinalTableau()
        {
            NativeMethods.lpr_tableau_shape(Tableau, out int r, out int c, out _);
            var t = new double[r, c];
            NativeMethods.ThrowIfError(NativeMethods.lpr_tableau_read(Tableau, t), "lpr_tableau_read");
            return t;
        }

        public List<int> BasicVariables
        {
            get { var b = new int[Math.Max(1, numConstraints)]; NativeMethods.lpr_basis_read(Tableau, b); return b.Take(numConstraints).ToList(); }
        }

        private void CaptureSnapshot(string title) =>
            IterationSnapshots.Add(TableIterationFormater.Format(GetFinalTableau(), numVariables, title));

        public void Dispose() { if (Tableau != IntPtr.Zero) { NativeMethods.lpr_tableau_destroy(Tableau); Tableau = IntPtr.Zero; } }
    }

    public class RevisedPrimalSimplexSolver : IDisposable
    {
        private readonly int n, m;
        private readonly bool isMin;   // the reference's `isMin` field (:29), only printed
        private IntPtr solver;
        public List<string> IterationSnapshots { get; private set; } = new List<string>();
        public double FinalZ { get; private set; }
        public List<double> SolutionVector { get; private set; } = new List<double>();

        public RevisedPrimalSimplexSolver(List<double> objective, List<IOConstraint> constraints, bool isMinimization)
        {
            if (objective == null || objective.Count == 0) throw new ArgumentException("Objective cannot be null or empty.");
            if (constraints == null || constraints.Count == 0) throw new ArgumentException("Constraints cannot be null or empty.");
            n = objective.Count; m = constraints.Count; isMin = isMinimization;
            var A = new double[m, n];
            var b = new double[m];
            for (int i = 0; i < m; i++)
            {
                if (constraints[i].Coefficients.Count != n)
                    throw new ArgumentException($"Constraint {i + 1} has incorrect number of coefficients.");
                for (int j = 0; j < n; j++) A[i, j] = constraints[i].Coefficients[j];
                b[i] = constraints[i].RHS;
            }
            NativeMethods.ThrowIfError(NativeMethods.lpr_revised_create(Engine.Handle, n, m, objective.ToArray(), A, n, b,
                isMinimization ? 1 : 0, out solver), "lpr_revised_create");
        }

        /// <summary>Above this many table entries per snapshot none are kept (the reference would
        /// write megabytes of text per pivot); set before Solve() to force either way.</summary>
        public bool KeepSnapshots { get; set; }

        public void Solve()
        {
            int status;
            if (KeepSnapshots || (long)m * (n + m + 1) <= 4096)
            {
                // one pass of the reference's while-loop per call; after every pivot and at the
                // optimum the numbers of CaptureSnapshot (:294-387) are read back and formatted by
                // FormatSnapshot below
                int iteration = 0;
                while (true)
                {
                    NativeMethods.ThrowIfError(NativeMethods.lpr_revised_step(solver, out var info), "lpr_revised_step");
                    status = info.status;
                    if (status != (int)LprStatus.PivotLimit && status != (int)LprStatus.Optimal) break;
                    double[] y = new double[m], rc = new double[n + m], u = new double[m], ratios = new double[m], xB = new double[m];
                    int[] basisPre = new int[m];
                    NativeMethods.ThrowIfError(NativeMethods.lpr_revised_snapshot_read(solver, y, rc, u, ratios, basisPre, xB), "lpr_revised_snapshot_read");
                    var binvA = new double[m, n]; var binv = new double[m, m];
                    NativeMethods.ThrowIfError(NativeMethods.lpr_revised_binv_a_exact(solver, binvA), "lpr_revised_binv_a_exact");
                    NativeMethods.ThrowIfError(NativeMethods.lpr_revised_binv_read(solver, binv), "lpr_revised_binv_read");
                    bool optimal = status == (int)LprStatus.Optimal;
                    if (optimal) { u = new double[m]; for (int i = 0; i < m; i++) ratios[i] = double.PositiveInfinity; basisPre = BasicVariables.ToArray(); }
                    IterationSnapshots.Add(FormatSnapshot(optimal ? "Optimal" : $"Iteration {++iteration}", xB, y,
                        rc.Take(n).ToArray(), rc.Skip(n).ToArray(), info.entering, info.entering_rc_pre, u, ratios,
                        basisPre.ToList(), info.leaving_row, info.leaving_var, info.z_working, info.z_original, binvA, binv));
                    if (optimal) break;
                }
            }
            else
            {
                var opts = new LprSolveOpts();
                NativeMethods.ThrowIfError(NativeMethods.lpr_revised_solve(solver, ref opts, out var res), "lpr_revised_solve");
                status = res.status;
            }
            switch ((LprStatus)status)
            {
                case LprStatus.Optimal:
                    var x = new double[n];
                    NativeMethods.lpr_revised_solution(solver, x, out double z);
                    SolutionVector = x.ToList(); FinalZ = z; break;
                // the reference's `throw new Exception(...)` texts (RevisedPrimalSimplexSolver.cs:91,179,183,267)
                case LprStatus.InfeasibleBasis: throw new Exception("Infeasible basis (negative basic value).");
                case LprStatus.Unbounded: throw new Exception("Unbounded problem (no positive component in direction).");
                case LprStatus.EnteringAlreadyBasic: throw new Exception("Internal error: entering variable is already basic.");
                case LprStatus.PivotTooSmall: throw new Exception("Pivot too small.");
            }
        }

        public List<int> BasicVariables { get { var b = new int[m]; NativeMethods.lpr_revised_basis_read(solver, b); return b.ToList(); } }

        // The text block of the reference's CaptureSnapshot (RevisedPrimalSimplexSolver.cs:294-387), built from the
        // numbers the engine hands back (its two matrix products arrive as binvA / binv).  Same layout as the Python
        // mirror (lpr_381_group_v22_amd/revised_primal_simplex_solver.py), which tests/test_revised_gpu.py compares
        // byte for byte with an independent restatement.  NumFormat.N3 stays the reference's own (:451-466).
        private static string Label(int idx, int nVars) => idx < nVars ? $"x{idx + 1}" : $"S{idx - nVars + 1}";   // VarLabel :289-292

        private string FormatSnapshot(string title, double[] xB, double[] y, double[] rcX_post, double[] rcS_post,
            int enteringIdx, double enteringRC_pre, double[] u_pre, double[] ratios_pre, List<int> basisForRatios_Pre,
            int leavingRow, int leavingVarIndex_Pre, double zWorking, double zOriginal, double[,] BInvA, double[,] BInv)
        {
            Func<IEnumerable<double>, string> tabbed = v => string.Join("\t", v.Select(NumFormat.N3));
            var sb = new System.Text.StringBuilder();
            sb.AppendLine(title);
            sb.AppendLine("Current Tableau (Revised Simplex)");
            sb.AppendLine("Problem type: " + (isMin ? "MIN (solving by MAX of -c)" : "MAX"));
            sb.AppendLine();
            sb.AppendLine("Dual prices (y = c_B^T B^{-1}):");
            sb.AppendLine(tabbed(y));
            sb.AppendLine();
            sb.AppendLine("Reduced costs:");
            sb.AppendLine("  x: " + tabbed(rcX_post));
            sb.AppendLine("  s: " + tabbed(rcS_post));
            sb.AppendLine();
            if (enteringIdx >= 0)
            {
                string entering = Label(enteringIdx, n);
                sb.AppendLine($"Entering variable (chosen pre-pivot): {entering}  (reduced cost pre = {NumFormat.N3(enteringRC_pre)})");
                sb.AppendLine("Direction u = B^{-1} a_enter (pre-pivot):");
                sb.AppendLine(tabbed(u_pre));
                sb.AppendLine();
                sb.AppendLine("Ratio test (xB_i / u_i; \u221E if u_i \u2264 0)  [labels = pre-pivot basis]:");
                for (int i = 0; i < m; i++)
                    sb.AppendLine(Label(basisForRatios_Pre[i], n) + ": " +
                                  (double.IsPositiveInfinity(ratios_pre[i]) ? "\u221E" : NumFormat.N3(ratios_pre[i])));
                if (leavingRow >= 0 && leavingVarIndex_Pre >= 0)
                {
                    sb.AppendLine($"Pivot (pre\u2192post): {Label(leavingVarIndex_Pre, n)}  \u2192  {entering}    (pivot = {NumFormat.N3(u_pre[leavingRow])})");
                    sb.AppendLine();
                }
            }
            sb.AppendLine("Working objective Z_working (maxified): " + NumFormat.N3(zWorking));
            sb.AppendLine($"Original objective Z_original ({(isMin ? "MIN" : "MAX")}): {NumFormat.N3(zOriginal)}");
            sb.AppendLine();
            sb.Append("Table\t");
            for (int j = 0; j < n; j++) sb.Append($"x{j + 1}\t");
            for (int j = 0; j < m; j++) sb.Append($"S{j + 1}\t");
            sb.AppendLine("RHS");
            sb.Append("Z~\t");
            foreach (double v in rcX_post) sb.Append(NumFormat.N3(v) + "\t");
            foreach (double v in rcS_post) sb.Append(NumFormat.N3(v) + "\t");
            sb.AppendLine(NumFormat.N3(zWorking));
            var post = BasicVariables;
            for (int i = 0; i < m; i++)
            {
                sb.Append(Label(post[i], n) + "\t");
                for (int j = 0; j < n; j++) sb.Append(NumFormat.N3(BInvA[i, j]) + "\t");
                for (int j = 0; j < m; j++) sb.Append(NumFormat.N3(BInv[i, j]) + "\t");
                sb.AppendLine(NumFormat.N3(xB[i]));
            }
            sb.AppendLine("Basic Variables: " + string.Join(", ", post.Select(v => Label(v, n))));
            return sb.ToString();
        }

        /// <summary>B^-1 * A of CaptureSnapshot (:360) on the fp64 matrix cores.</summary>
        public double[,] BInverseTimesA() { var p = new double[m, n]; NativeMethods.lpr_revised_binv_a(solver, p, out _); return p; }

        public void Dispose() { if (solver != IntPtr.Zero) { NativeMethods.lpr_revised_destroy(solver); solver = IntPtr.Zero; } }
    }
}

namespace LPR_381_Group_V22.IntegerProgramming
{
    using LPR_381_Group_V22.Simplex;

    public static class BranchAndBoundAdapter
    {
        public static (List<double> x, double z) SolveFromPrimal(PrimalSimplexSolver primal, bool enablePruning = false, bool isMin = false)
        {
            if (primal.FinalTableau == null)
                throw new InvalidOperationException("Primal simplex has not been solved yet.");   // BranchAndBoundAdapter.cs:11-14
            int nvars = primal.SolutionVector?.Count ?? Math.Max(1, primal.FinalTableau.GetLength(1) - 1);   // :20
            // the FinalTableau is still resident on the device: no double[,] -> List<List<double>> conversion
            NativeMethods.ThrowIfError(NativeMethods.lpr_bb_create_from_tableau(primal.Tableau, nvars, 20, out IntPtr bb), "lpr_bb_create_from_tableau");
            try
            {
                var opts = new LprBbOpts { enable_pruning = enablePruning ? 1 : 0, node_cap = 20 };   // :1038
                var x = new double[Math.Max(1, nvars)];
                NativeMethods.ThrowIfError(NativeMethods.lpr_bb_run(bb, ref opts, x, out var res), "lpr_bb_run");
                if (res.status == (int)LprStatus.BbNodeCap) Console.WriteLine("Potential infinite loop detected");
                return res.found != 0 ? (x.Take(nvars).ToList(), res.z) : (new List<double>(), double.NegativeInfinity);   // :23
            }
            finally { NativeMethods.lpr_bb_destroy(bb); }
        }
    }
}

namespace LPR_381_Group_V22.SensitivityAnalysis
{
    using LPR_381_Group_V22.Simplex;

    /// <summary>
    /// The numeric half of SensitivityAnalyzer (SensitivityAnalyzer.cs) on the device.  The reference class keeps its
    /// prompts / Console output and replaces its double[,] tableau + Pivot / ReOptimize / DualSimplexIfNeeded /
    /// RebuildBasicsFromTableau by calls on this handle.  Uncompiled here (no .NET toolchain in the build image).
    /// </summary>
    internal sealed class GpuSensitivity : IDisposable
    {
        private IntPtr h;

        // Program.cs:147-151 -- the solved tableau is copied device to device
        internal GpuSensitivity(PrimalSimplexSolver primal, int numDecisionVariables)
        {
            NativeMethods.ThrowIfError(NativeMethods.lpr_sens_create_from_tableau(primal.Tableau, numDecisionVariables, out h), "lpr_sens_create_from_tableau");
        }

        // SensitivityAnalyzer(double[,], List<double>, double, List<int>) :22-39 (basicVariables is rebuilt by :35)
        internal GpuSensitivity(double[,] finalTableau, List<double> solution, double zValue)
        {
            NativeMethods.ThrowIfError(NativeMethods.lpr_sens_create(Engine.Handle, finalTableau, finalTableau.GetLength(0), finalTableau.GetLength(1),
                solution.ToArray(), solution.Count, zValue, out h), "lpr_sens_create");
        }

        private static void Throw(int outcome)
        {
            switch (outcome)
            {
                case 1: throw new InvalidOperationException("Unbounded during re-optimization.");            // :151
                case 2: throw new InvalidOperationException("Infeasible after RHS change (dual simplex).");  // :197
                case 3: throw new InvalidOperationException("Zero pivot encountered.");                      // :101
                case 5: throw new InvalidOperationException("Re-optimization exceeded iteration limit.");    // :126 / :183
                case 9: throw new IndexOutOfRangeException();                                                // tech[basicVars[pos]] with -1, :642
            }
        }

        /// <returns>false when the C# would have printed "Invalid ..." and returned</returns>
        internal bool ChangeNonBasicReducedCost(int index, double newCbar)
        { NativeMethods.ThrowIfError(NativeMethods.lpr_sens_change_nonbasic_cbar(h, index, newCbar, out int oc), "lpr_sens_change_nonbasic_cbar"); Throw(oc); return oc == 0; }
        internal bool ChangeBasic(int col, double delta)
        { NativeMethods.ThrowIfError(NativeMethods.lpr_sens_change_basic(h, col, delta, out int oc), "lpr_sens_change_basic"); Throw(oc); return oc == 0; }
        /// <returns>0 re-solved, 8 rolled back (the caller prints the C#'s message :467-468), -1 invalid index</returns>
        internal int ChangeRHS(int k, double newB)
        { NativeMethods.ThrowIfError(NativeMethods.lpr_sens_change_rhs(h, k, newB, out int oc), "lpr_sens_change_rhs"); return oc; }
        internal bool ChangeNonBasicColumn(int row, int col, double newVal)
        { NativeMethods.ThrowIfError(NativeMethods.lpr_sens_change_nonbasic_column(h, row, col, newVal, out int oc), "lpr_sens_change_nonbasic_column"); Throw(oc); return oc == 0; }
        internal void AddNewActivity(double cNew, double[] aNew)
        { NativeMethods.ThrowIfError(NativeMethods.lpr_sens_add_activity(h, cNew, aNew, aNew.Length, out int oc), "lpr_sens_add_activity"); Throw(oc); }
        internal void AddNewConstraint(double[] tech, double rhs)
        { NativeMethods.ThrowIfError(NativeMethods.lpr_sens_add_constraint(h, tech, tech.Length, rhs, out int oc), "lpr_sens_add_constraint"); Throw(oc); }

        internal int GetBasicRow(int col) { NativeMethods.ThrowIfError(NativeMethods.lpr_sens_basic_row(h, col, out int r), "lpr_sens_basic_row"); return r; }   // :69-77

        internal double[,] CurrentTableau   // :727
        {
            get
            {
                NativeMethods.lpr_sens_shape(h, out int R, out int C, out _, out _, out _, out _);
                var t = new double[R, C];
                NativeMethods.ThrowIfError(NativeMethods.lpr_sens_read(h, t, null, null), "lpr_sens_read");
                return t;
            }
        }
        internal double CurrentZ { get { NativeMethods.lpr_sens_shape(h, out _, out _, out _, out _, out double z, out _); return z; } }   // :728
        internal List<double> CurrentSolutionVector   // :729
        {
            get
            {
                NativeMethods.lpr_sens_shape(h, out _, out _, out int ns, out _, out _, out _);
                var x = new double[Math.Max(1, ns)];
                NativeMethods.ThrowIfError(NativeMethods.lpr_sens_read(h, null, null, x), "lpr_sens_read");
                return x.Take(ns).ToList();
            }
        }
        internal double[] Row(int i) { NativeMethods.lpr_sens_shape(h, out _, out int C, out _, out _, out _, out _); var r = new double[C]; NativeMethods.ThrowIfError(NativeMethods.lpr_sens_read_block(h, i, 1, 0, C, r), "lpr_sens_read_block"); return r; }
        internal double[] Column(int j) { NativeMethods.lpr_sens_shape(h, out int R, out _, out _, out _, out _, out _); var c = new double[R]; NativeMethods.ThrowIfError(NativeMethods.lpr_sens_read_block(h, 0, R, j, 1, c), "lpr_sens_read_block"); return c; }
        // A-tilde^T y of RecoverObjectiveC / PerformDuality (:236-245, :690-694), summed in the C#'s order on the device
        internal double[] ATy(double[] y, int n) { var o = new double[Math.Max(1, n)]; NativeMethods.ThrowIfError(NativeMethods.lpr_sens_column_fold(h, y, y.Length, null, n, o), "lpr_sens_column_fold"); return o; }

        public void Dispose() { if (h != IntPtr.Zero) { NativeMethods.lpr_sens_destroy(h); h = IntPtr.Zero; } }
    }
}
