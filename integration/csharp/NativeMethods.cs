// NativeMethods.cs -- P/Invoke binding of include/lpr_engine.h for LPR_381_Group_V22.
// UNVERIFIED: no C# toolchain exists in the build image, this file has never been compiled.
// Add it (and the three Gpu*.cs wrappers) to LPR_381_Group_V22.csproj, build x64
// (<PlatformTarget>x64</PlatformTarget>: the engine is a 64-bit library) and ship liblpr_engine.so
// (Linux/.NET on ROCm) next to the executable.
using System;
using System.Runtime.InteropServices;

namespace LPR_381_Group_V22.Native
{
    internal enum LprStatus
    {
        Optimal = 0, Unbounded = 1, InfeasibleBasis = 2, PivotTooSmall = 3,
        EnteringAlreadyBasic = 4, PivotLimit = 5, BbNodeCap = 6, BbDepthCap = 7,
        BadArgument = -1, DeviceError = -2, OutOfMemory = -3
    }

    [StructLayout(LayoutKind.Sequential)]
    internal struct LprSolveOpts
    {
        public long max_pivots; public int time_kernels; public int batch; public int variant; public int block;
    }

    [StructLayout(LayoutKind.Sequential)]
    internal struct LprSolveResult
    {
        public int status; public int block; public long pivots; public long total_pivots; public double z;
    }

    [StructLayout(LayoutKind.Sequential)]
    internal struct LprRevisedSnapshotInfo
    {
        public int status, entering, leaving_row, leaving_var;
        public double entering_rc_pre, z_working, z_original;
    }

    [StructLayout(LayoutKind.Sequential)]
    internal struct LprRevisedResult
    {
        public int status; public int reserved; public long iterations; public long total_iterations; public double z;
    }

    [StructLayout(LayoutKind.Sequential)]
    internal struct LprBbOpts
    {
        public int enable_pruning; public int node_cap; public int reserved0; public int reserved1;
    }

    [StructLayout(LayoutKind.Sequential)]
    internal struct LprBbResult
    {
        public int status; public int found; public long processed; public int best_node; public int reserved;
        public double z; public long pivots; public long nodes_created;
    }

    [StructLayout(LayoutKind.Sequential)]
    internal struct LprBbSyncOpts
    {
        public int enable_pruning; public int max_levels; public long max_nodes;
    }

    [StructLayout(LayoutKind.Sequential)]
    internal struct LprBbSyncResult
    {
        public int status; public int found; public long processed; public long pivots;
        public int levels; public int path_len; public ulong path_bits; public double z;
    }

    // transport callbacks of lpr_comm_init_custom (a host that brings its own fabric); called on the calling thread only
    [UnmanagedFunctionPointer(CallingConvention.Cdecl)] internal delegate int LprAllReduceMaxFn(IntPtr user, IntPtr inout, int count);
    [UnmanagedFunctionPointer(CallingConvention.Cdecl)] internal delegate int LprAllGatherFn(IntPtr user, IntPtr send, IntPtr recv, int bytes);

    internal static class NativeMethods
    {
        internal const int LPR_COMM_ID_BYTES = 128;
        private const string Lib = "lpr_engine"; // liblpr_engine.so
        private const CallingConvention CC = CallingConvention.Cdecl;

        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_abi_version();
        [DllImport(Lib, CallingConvention = CC)] internal static extern IntPtr lpr_last_error();
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_engine_open(int device, out IntPtr engine);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_engine_close(IntPtr engine);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_engine_sync(IntPtr engine);
        [DllImport(Lib, CallingConvention = CC)] internal static extern ulong lpr_engine_stream(IntPtr engine);

        // ---- PrimalSimplexSolver (Simplex/PrimalSimplexSolver.cs) ----
        [DllImport(Lib, CallingConvention = CC)]
        internal static extern int lpr_tableau_from_lp(IntPtr engine, int n, int m, double[] objective, double[] A, int lda,
            int[] ncoef, sbyte[] relation, double[] rhs, int is_max, out IntPtr tableau);
        [DllImport(Lib, CallingConvention = CC)]
        internal static extern int lpr_tableau_create(IntPtr engine, int rows, int cols, double[,] rowmajor, int[] basis, out IntPtr tableau);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_tableau_destroy(IntPtr tableau);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_tableau_shape(IntPtr tableau, out int rows, out int cols, out int ld);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_primal_solve(IntPtr tableau, ref LprSolveOpts opts, out LprSolveResult res);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_select_entering(IntPtr tableau, out int col);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_select_leaving(IntPtr tableau, int col, out int row);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_pivot(IntPtr tableau, int row, int col);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_extract_solution(IntPtr tableau, int n, double[] x, out double z);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_tableau_read(IntPtr tableau, double[,] rowmajorOut);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_basis_read(IntPtr tableau, int[] basisOut);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_pivot_log_read(IntPtr tableau, int[] rows, int[] cols, long cap, out long count);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_tableau_read_block(IntPtr tableau, int row0, int nrows, int col0, int ncols, double[] block);
        // benchmark input + kernel timing (bench.py's counterparts; no reference member)
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_tableau_synthetic(IntPtr engine, int m, int n, ulong seed, out IntPtr tableau);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_tableau_kernel_stats(IntPtr tableau, out long launches, out double totalMs, out double avgMs);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_tableau_step_stats(IntPtr tableau, out long steps, out double totalMs);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_debug_head_stamps(IntPtr tableau, ulong[] stamps, long cap, out long count);

        // ---- DualSimplexSolver (Simplex/DualSimplex.cs:14-114), PrimalSimplexSolver2 (Simplex/PrimalSimplexSolver2.cs:46-97),
        //      CuttingPlaneSolver.CuttingPlaneSolution (IntegerProgramming/CuttingPlaneSolver.cs:64-229) on a tableau handle ----
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_dual_solve(IntPtr tableau, int maxIters, int printSteps, long hardCap, out LprSolveResult res);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_primal2_solve(IntPtr tableau, int maxIters, int printSteps, long hardCap, out LprSolveResult res);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_cutting_plane(IntPtr tableau, int maxCuts, long hardCap, out int exitCode, out int cuts);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_cut_log_read(IntPtr tableau, int[] triples, long cap, out long count);

        // ---- RevisedPrimalSimplexSolver (Simplex/RevisedPrimalSimplexSolver.cs) ----
        [DllImport(Lib, CallingConvention = CC)]
        internal static extern int lpr_revised_create(IntPtr engine, int n, int m, double[] objective, double[,] A, int lda, double[] b, int is_min, out IntPtr solver);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_revised_destroy(IntPtr solver);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_revised_solve(IntPtr solver, ref LprSolveOpts opts, out LprRevisedResult res);
        // IterationSnapshots (RevisedPrimalSimplexSolver.cs:36, CaptureSnapshot :294-387)
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_revised_step(IntPtr solver, out LprRevisedSnapshotInfo info);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_revised_snapshot_read(IntPtr solver, double[] y, double[] rc, double[] u, double[] ratios, int[] basisPre, double[] xB);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_revised_binv_a_exact(IntPtr solver, double[,] product);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_revised_binv_read(IntPtr solver, double[,] binv);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_revised_solution(IntPtr solver, double[] x, out double z);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_revised_basis_read(IntPtr solver, int[] basis);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_revised_xb_read(IntPtr solver, double[] xb);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_revised_binv_a(IntPtr solver, double[,] product, out double ms);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_revised_log_read(IntPtr solver, int[] rows, int[] entering, int[] leaving, long cap, out long count);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_revised_synthetic(IntPtr engine, int m, int n, ulong seed, out IntPtr solver);

        // ---- BranchAndBoundAdapter / BranchBoundSimplexSolver ----
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_bb_create_from_tableau(IntPtr tableau, int nvars, int max_depth, out IntPtr bb);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_bb_create(IntPtr engine, double[,] finalTableau, int rows, int cols, int nvars, int max_depth, out IntPtr bb);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_bb_run(IntPtr bb, ref LprBbOpts opts, double[] x, out LprBbResult res);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_bb_destroy(IntPtr bb);
        // node records / pop order / pivot trace of the last lpr_bb_run (what ExecuteBranchAndBound prints per branch, :1049-1230)
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_bb_records_read(IntPtr bb, int[] parent, int[] kind, int[] depth, int[] var, double[] bound, int[] status, double[] z, long cap, out long count);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_bb_pop_order_read(IntPtr bb, int[] ids, long cap, out long count);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_bb_trace_read(IntPtr bb, int[] quads, long cap, out long count);
        // building blocks: batched AddConstraint (:694-803) + DoDualSimplex (:289-468) of many children, node scoring, buffers
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_bb_expand(IntPtr bb, int count, int[] parentIds, int[] var, double[] bound, int[] kind, int[] childIds, int[] status, int[] pivots);
        // the same + what ExecuteBranchAndBound prints about each child: pivot triples and every tableau of DoDualSimplex's list
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_bb_expand_traced(IntPtr bb, int count, int[] parentIds, int[] var, double[] bound, int[] kind, int[] childIds, int[] status, int[] pivots, int[] trace, long traceCap, long[] traceOff, double[] tableaux, long tabCap, long[] tabOff, int[] ntab);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_bb_node_info(IntPtr bb, int[] ids, int count, double[] z, double[] vals);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_bb_node_read(IntPtr bb, int id, double[] tableau, out int rows, out int cols);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_bb_release(IntPtr bb, int[] ids, int count);
        // level-synchronous, multi-GPU form (cap lifted): ONE all-reduce(MAX) per level issued by the library on RCCL
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_bb_solve_level_sync(IntPtr bb, IntPtr comm, ref LprBbSyncOpts opts, double[] x, out LprBbSyncResult res);

        // ---- multi-GPU communicator (one process per GPU; RCCL over xGMI inside the library) ----
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_comm_unique_id([Out] byte[] id128);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_comm_init(IntPtr engine, int rank, int world, byte[] id128, out IntPtr comm);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_comm_init_custom(int rank, int world, LprAllReduceMaxFn allReduceMax, LprAllGatherFn allGather, IntPtr user, out IntPtr comm);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_comm_destroy(IntPtr comm);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_comm_info(IntPtr comm, out int rank, out int world, out long allReduceCalls, out long allGatherCalls);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_comm_all_reduce_max(IntPtr comm, double[] inout, int count);

        // ---- SensitivityAnalyzer (SensitivityAnalysis/SensitivityAnalyzer.cs); outcome = lpr_sens_outcome ----
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_create(IntPtr engine, double[,] finalTableau, int rows, int cols, double[] solution, int nsol, double z, out IntPtr sens);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_create_from_tableau(IntPtr tableau, int nDecision, out IntPtr sens);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_destroy(IntPtr sens);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_shape(IntPtr sens, out int rows, out int cols, out int nsol, out int nbasic, out double z, out long lastPivots);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_read(IntPtr sens, double[,] tableau, int[] basic, double[] solution);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_read_block(IntPtr sens, int row0, int nrows, int col0, int ncols, double[] block);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_basic_row(IntPtr sens, int col, out int row);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_log_read(IntPtr sens, int[] triples, long cap, out long count);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_column_fold(IntPtr sens, double[] w, int nw, double[] init, int ncols, double[] result);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_resolve_all(IntPtr sens, out int outcome);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_change_nonbasic_cbar(IntPtr sens, int index, double newCbar, out int outcome);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_change_basic(IntPtr sens, int col, double delta, out int outcome);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_change_rhs(IntPtr sens, int k, double newB, out int outcome);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_change_nonbasic_column(IntPtr sens, int row, int col, double newVal, out int outcome);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_add_activity(IntPtr sens, double cNew, double[] aNew, int na, out int outcome);
        [DllImport(Lib, CallingConvention = CC)] internal static extern int lpr_sens_add_constraint(IntPtr sens, double[] tech, int ntech, double rhs, out int outcome);

        internal static string LastError() => Marshal.PtrToStringAnsi(lpr_last_error()) ?? "";

        internal static void ThrowIfError(int status, string where)
        {
            if (status < 0) throw new InvalidOperationException($"{where}: {(LprStatus)status} -- {LastError()}");
        }
    }

    /// <summary>One engine (HIP device 0 + stream) for the process, released at exit.</summary>
    internal static class Engine
    {
        private static IntPtr _handle = IntPtr.Zero;
        internal static IntPtr Handle
        {
            get
            {
                if (_handle == IntPtr.Zero)
                {
                    NativeMethods.ThrowIfError(NativeMethods.lpr_engine_open(0, out _handle), "lpr_engine_open");
                    AppDomain.CurrentDomain.ProcessExit += (s, e) => NativeMethods.lpr_engine_close(_handle);
                }
                return _handle;
            }
        }
    }
}
