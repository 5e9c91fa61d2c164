/*
 * lpr_engine.h -- C ABI of the MI355X (gfx950) simplex pivot engine.
 *
 * This is the drop-in boundary for the dense-tableau hot path of
 * Storm-Tarran/LPR_381_Group_V22.  The reference has no FFI layer of its own;
 * its de-facto boundary is the public surface of three C# classes consumed by
 * Program.cs.  Every entry point below names the reference member it replaces
 * (paths relative to LPR_381_Group_V22/ in the reference tree).  The C# side
 * binds these with [DllImport("lpr_engine")] -- see INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no exceptions cross this boundary.
 *   - every function returns an lpr_status (>= 0: solver outcome, < 0: error);
 *     the text of the last error on the calling thread is lpr_last_error().
 *   - input buffers are borrowed for the duration of the call only (P/Invoke
 *     pins blittable arrays); outputs go to caller-allocated buffers.
 *   - handles are opaque, owned by the library, not thread-safe; one engine
 *     == one HIP device + one stream.
 *   - matrices are row-major fp64 (C# double[,] is contiguous row-major);
 *     column / variable indices are 0-based; tableau row 0 is the Z row, so
 *     constraint rows (and pivot-log rows of the tableau solver) are 1-based
 *     exactly as in PrimalSimplexSolver.cs:138,142.
 *   - all arithmetic on the device is IEEE binary64, no FMA contraction,
 *     true division -- bit-for-bit the C# expressions it replaces.
 */
#ifndef LPR_ENGINE_H
#define LPR_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPR_ABI_VERSION 1

/* Solver outcomes mirror the reference's behaviours one-to-one:
 *   OPTIMAL              PrimalSimplexSolver.cs:110-126 / RevisedPrimalSimplexSolver.cs:124-146
 *   UNBOUNDED            PrimalSimplexSolver.cs:129-135 (print + break) /
 *                        RevisedPrimalSimplexSolver.cs:178-179 (throw "Unbounded problem ...")
 *   INFEASIBLE_BASIS     RevisedPrimalSimplexSolver.cs:90-91
 *   PIVOT_TOO_SMALL      RevisedPrimalSimplexSolver.cs:267
 *   ENTERING_ALREADY_BASIC RevisedPrimalSimplexSolver.cs:182-183
 *   PIVOT_LIMIT          (no reference equivalent: the C# loops are uncapped; returned only when
 *                         the caller sets max_pivots)
 *   BB_NODE_CAP          BranchBoundSimplexSolver.cs:1038-1042 ("Potential infinite loop detected")
 *   BB_DEPTH_CAP         (no reference equivalent: lpr_bb_solve_level_sync stopped at max_levels
 *                         with fractional nodes it scored but did not branch)
 */
typedef enum lpr_status {
    LPR_OK_OPTIMAL = 0,
    LPR_UNBOUNDED = 1,
    LPR_INFEASIBLE_BASIS = 2,
    LPR_PIVOT_TOO_SMALL = 3,
    LPR_ENTERING_ALREADY_BASIC = 4,
    LPR_PIVOT_LIMIT = 5,
    LPR_BB_NODE_CAP = 6,
    LPR_BB_DEPTH_CAP = 7,
    LPR_BAD_ARGUMENT = -1,
    LPR_DEVICE_ERROR = -2,
    LPR_OUT_OF_MEMORY = -3
} lpr_status;

/* Constraint relation codes for lpr_tableau_from_lp (InputFileParser.Constraint.Relation,
 * IO/InputFileParser.cs:70-82).  Anything that is not ">=" is treated as "<=" by the reference
 * (PrimalSimplexSolver.cs:36-50). */
enum { LPR_REL_LE = 0, LPR_REL_GE = 1, LPR_REL_EQ = 2 };

typedef struct lpr_engine lpr_engine;
typedef struct lpr_tableau lpr_tableau;

/* ------------------------------------------------------------------ engine */

int lpr_abi_version(void);
/* Thread-local text of the last failure ("" if none). */
const char* lpr_last_error(void);
/* Binds HIP device `device`, creates the engine stream.  LPR_DEVICE_ERROR if no gfx950 device. */
int lpr_engine_open(int device, lpr_engine** out);
int lpr_engine_close(lpr_engine* e);
/* Blocks until all work queued on the engine stream has finished. */
int lpr_engine_sync(lpr_engine* e);
/* The engine's hipStream_t as an integer (for callers that time with their own HIP events). */
uint64_t lpr_engine_stream(lpr_engine* e);

/* ------------------------------------------------------- tableau (primal) */

/* Replaces `new PrimalSimplexSolver(objective, constraints, isMaximization)`
 * (Simplex/PrimalSimplexSolver.cs:27-87), built on the device:
 *   row 0      = -c (max) or +c (min)                         :61-62
 *   ">=" rows  negated incl. RHS; "=" and others kept as "<=" :36-50
 *   row i+1    = [first min(n, ncoef[i]) coeffs | e_i | rhs]  :68-82
 *   basis[i]   = n + i                                         :78
 * A is m x lda row-major (lda >= n); ncoef may be NULL (== n for every row). */
int lpr_tableau_from_lp(lpr_engine* e, int n, int m, const double* objective,
                        const double* A, int lda, const int32_t* ncoef,
                        const int8_t* relation, const double* rhs, int is_max,
                        lpr_tableau** out);

/* Adopts a ready (rows x cols) row-major tableau + basis (rows-1 column indices); the entry used
 * by BranchAndBoundAdapter-style callers that already hold a double[,] (BranchAndBoundAdapter.cs:31-46). */
int lpr_tableau_create(lpr_engine* e, int rows, int cols, const double* rowmajor,
                       const int32_t* basis, lpr_tableau** out);

/* Benchmark input: dense random LP generated ON the device by the counter-based
 * generator documented in DESIGN.md (SplitMix64 keyed by (seed, stream, i, j)), so that the
 * 402.8 MB tableau of the m=4096,n=8192 configuration never crosses PCIe:
 *   A[i][j] ~ U(0,1), b_i = (n/4)*(1 + 0.1*U(0,1)), c_j ~ U(0,1), all "<=", maximise. */
int lpr_tableau_synthetic(lpr_engine* e, int m, int n, uint64_t seed, lpr_tableau** out);

int lpr_tableau_destroy(lpr_tableau* t);
int lpr_tableau_shape(const lpr_tableau* t, int* rows, int* cols, int* ld);

/* Solver options.  Zero-initialise, then set fields; defaults equal the reference's literals. */
typedef struct lpr_solve_opts {
    int64_t max_pivots;    /* <= 0: uncapped like PrimalSimplexSolver.cs:107 */
    int32_t time_kernels;  /* != 0: launch eagerly and bracket rank-1 update / sweep launches with
                              HIP events on the engine stream (every sweep on the K-pivot paths,
                              one update in four on the one-pivot path); read back with
                              lpr_tableau_kernel_stats / lpr_tableau_step_stats  On the K-pivot paths a value n > 1 brackets every
                              n-th step only (the events sit on the sweep's stream: two per step
                              cost the two-stream pipeline ~3 %). */
    int32_t batch;         /* pivots queued between host polls of the device status word (0: auto) */
    int32_t variant;       /* kernel variant (0: auto); for tuning and tests only, same bits.  Low 16
                              bits: path + sweep tile (0x30tr two-stream overlap, 0x40tr heads then
                              in-place sweep, 0x50tr one-launch overlap, 0x60tr one launch per head;
                              tr = 0x04/0x08/0x10 rows per chunk, 0x24/0x28 two chunks in flight).
                              Bits 16..22, K-pivot paths: 0x10000 diagnostic time stamps of the loop
                              heads, 0x20000 loop heads not confined to one XCD, 0x40000 confined
                              but hand-offs through the memory side, 0x80000 the sweep does not
                              leave the heads' XCD to them, 0x100000 it does so only once this
                              launch's heads have said where they are (no hint from the previous
                              launch), 0x200000 the heads of a step follow their predecessor
                              at once and wait for the previous sweep on a device flag instead of
                              a cross-stream event, 0x400000 the sweep of a step follows its
                              predecessor at once and asks the heads' completion word as it starts.
                              Both need the two kernels of a step to run at the same time (tools
                              that serialise kernels -- rocprofv3 --pmc -- prevent that: the
                              bounded waits then end the solve with LPR_DEVICE_ERROR) and neither
                              is faster than the events any more: opt-in only. */
    int32_t block;         /* pivots decided ahead and applied per sweep of the tableau on large
                              tableaux: 0 auto (16), 1 one pivot per sweep, 2..16 that many.  The bits
                              stored are the same for every value (each element goes through the
                              same sequence of rounded operations, in registers). */
} lpr_solve_opts;

typedef struct lpr_solve_result {
    int32_t status;        /* lpr_status */
    int32_t block;         /* pivots per sweep this call used (1 on the small-tableau paths) */
    int64_t pivots;        /* pivots performed by THIS call */
    int64_t total_pivots;  /* pivots performed on this tableau so far (== C# `iteration`) */
    double z;              /* T[0, cols-1] at exit (FinalZ, PrimalSimplexSolver.cs:113); the C#
                              leaves FinalZ = 0 on the unbounded exit -- the host mirror does that */
} lpr_solve_result;

/* Replaces PrimalSimplexSolver.Solve() (Simplex/PrimalSimplexSolver.cs:102-150): repeats
 * FindEnteringVariable -> FindLeavingVariable -> Pivot -> basicVariables[r-1] = e on the device
 * until optimal / unbounded / max_pivots.  No tableau data crosses the host boundary. */
int lpr_primal_solve(lpr_tableau* t, const lpr_solve_opts* opts, lpr_solve_result* res);

/* Single-step forms of the same three private methods, for callers (and tests) that drive the
 * loop themselves:
 *   lpr_select_entering  FindEnteringVariable  PrimalSimplexSolver.cs:152-167   (-1: optimal)
 *   lpr_select_leaving   FindLeavingVariable   PrimalSimplexSolver.cs:169-191   (-1: unbounded)
 *   lpr_pivot            Pivot                 PrimalSimplexSolver.cs:193-211   (+ basis update :142)
 */
int lpr_select_entering(lpr_tableau* t, int32_t* col);
int lpr_select_leaving(lpr_tableau* t, int32_t col, int32_t* row);
int lpr_pivot(lpr_tableau* t, int32_t row, int32_t col);

/* Replaces ExtractSolution() + FinalZ (PrimalSimplexSolver.cs:213-252, :113): x has n entries. */
int lpr_extract_solution(lpr_tableau* t, int n, double* x, double* z);

/* Result accessors (PrimalSimplexSolver.cs:18-24, 269-278). */
int lpr_tableau_read(lpr_tableau* t, double* rowmajor_out);            /* rows*cols doubles   */
int lpr_tableau_read_block(lpr_tableau* t, int row0, int nrows, int col0, int ncols,
                           double* out);                               /* nrows*ncols doubles */
int lpr_basis_read(lpr_tableau* t, int32_t* basis_out);                /* rows-1 ints         */
/* Pivot log: (row, col) pairs in order, row 1-based like the C# console line (:138).  Returns the
 * number of pairs written through *count (at most cap). */
int lpr_pivot_log_read(lpr_tableau* t, int32_t* rows_out, int32_t* cols_out, int64_t cap,
                       int64_t* count);

/* Timing of the rank-1 update launches recorded while opts.time_kernels was set:
 * launches, summed and average duration in milliseconds (HIP events on the engine stream). */
int lpr_tableau_kernel_stats(lpr_tableau* t, int64_t* launches, double* total_ms, double* avg_ms);
/* K-pivot paths: the steps timed with the sweeps above.  A step is one sweep of K pivots together
 * with the loop heads of the next K (start of one sweep to the start of the next). */
int lpr_tableau_step_stats(lpr_tableau* t, int64_t* steps, double* total_ms);
/* Diagnostic (opts.variant bit 16): s_memrealtime stamps (10 ns ticks) the lead loop-head workgroup
 * left per pivot and phase -- 64 pivots x 12 stamps, then its XCC id and whether the launch handed
 * off through the XCD's L2.  Not part of the solver surface of the reference. */
int lpr_debug_head_stamps(lpr_tableau* t, uint64_t* out, int64_t cap, int64_t* count);

/* ------------------------------------- cutting-plane side path ("next" row f3)
 * The reference's dual simplex / second primal simplex / Gomory step (dead code in its menu,
 * Program.cs:417-428) on the same device tableau: row 0 = objectiveRow, rows 1.. = constraintRows.
 * They differ from PrimalSimplexSolver by EPS-band selection rules and by leaving rows whose
 * factor is within 1e-9 of zero untouched.  `print_steps` reproduces the C# quirk that `iter`
 * (and with it max_iters) only advances when printSteps is set; hard_cap (<= 0: none) is an
 * extra pivot limit with no C# counterpart.
 *   lpr_dual_solve     DualSimplexSolver.Solve     Simplex/DualSimplex.cs:14-114
 *                      status: OPTIMAL (true) / INFEASIBLE_BASIS (false, :72-76) / PIVOT_LIMIT
 *   lpr_primal2_solve  PrimalSimplexSolver2.Solve  Simplex/PrimalSimplexSolver2.cs:46-97
 *                      status: OPTIMAL (true) / UNBOUNDED (false, :63-68) / PIVOT_LIMIT
 *   lpr_cutting_plane  CuttingPlaneSolver.CuttingPlaneSolution
 *                      IntegerProgramming/CuttingPlaneSolver.cs:64-229 (recursion unrolled, at most
 *                      max_cuts cuts; every cut appends one ROW, no slack column, :104-110).
 *                      *exit_code: 0 optimal tableau displayed (:224), 1 all RHS integral (:87-91),
 *                      2 no pivot column on the cut (:134-138), 3 pivot too small (:146-150),
 *                      4 dual simplex failed (:191), 5 "step finished" (:228), 6 max_cuts reached,
 *                      7 an InvalidOperationException escaped a solver.
 *   lpr_cut_log_read   pivots of this path in order: triples (kind 0 dual / 1 primal2 / 2 cut,
 *                      row in the C#'s own numbering, column). */
int lpr_dual_solve(lpr_tableau* t, int max_iters, int print_steps, int64_t hard_cap,
                   lpr_solve_result* res);
int lpr_primal2_solve(lpr_tableau* t, int max_iters, int print_steps, int64_t hard_cap,
                      lpr_solve_result* res);
int lpr_cutting_plane(lpr_tableau* t, int max_cuts, int64_t hard_cap, int32_t* exit_code,
                      int32_t* cuts);
int lpr_cut_log_read(lpr_tableau* t, int32_t* triples, int64_t cap, int64_t* count);

/* -------------------------------------------------- revised primal simplex */

typedef struct lpr_revised lpr_revised;

/* Replaces `new RevisedPrimalSimplexSolver(objective, constraints, isMinimization)`
 * (Simplex/RevisedPrimalSimplexSolver.cs:41-80): c = -objective if is_min (:51), dense A[m, n] and
 * b[m] copied to HBM, B^-1 = I, basis = slacks.  Every row is treated as "<=": the reference never
 * reads Constraint.Relation here.  The C# ArgumentExceptions (:43-44, :57-58: empty objective /
 * constraints, wrong coefficient count) are LPR_BAD_ARGUMENT; the row-length check is the
 * caller's because A arrives already flattened (lda >= n). */
int lpr_revised_create(lpr_engine* e, int n, int m, const double* objective, const double* A,
                       int lda, const double* b, int is_min, lpr_revised** out);
/* The synthetic dense LP of lpr_tableau_synthetic in (c, A, b) form, generated on the device. */
int lpr_revised_synthetic(lpr_engine* e, int m, int n, uint64_t seed, lpr_revised** out);
int lpr_revised_destroy(lpr_revised* s);

typedef struct lpr_revised_result {
    int32_t status;           /* lpr_status; the C# throws for 1..4 with the messages of
                                 :91 / :179 / :183 / :267 -- the host mirror re-raises them */
    int32_t reserved;
    int64_t iterations;       /* pivots performed by THIS call */
    int64_t total_iterations; /* pivots so far (the C# `iteration`, :249) */
    double z;                 /* FinalZ = Dot(cOrig, x) (:286); valid when status == OPTIMAL */
} lpr_revised_result;

/* Replaces RevisedPrimalSimplexSolver.Solve() (:82-251) + ExtractSolution (:277-287).  Per
 * iteration: x_B = B^-1 b, feasibility, y = c_B B^-1, reduced costs, entering fold (:105-121),
 * u = B^-1 a_e, ratio fold (:154-176), bookkeeping, B^-1 <- E B^-1 (:264-275).  All sums in the
 * C#'s sequential order, all on the device.  Only opts->max_pivots and opts->batch are used. */
int lpr_revised_solve(lpr_revised* s, const lpr_solve_opts* opts, lpr_revised_result* res);

/* SolutionVector / FinalZ (:36-38): x has n entries.  Valid after an OPTIMAL solve. */
int lpr_revised_solution(lpr_revised* s, double* x, double* z);
/* BasicVariables (:39): m entries, by basis row. */
int lpr_revised_basis_read(lpr_revised* s, int32_t* basis_out);
/* Pivot log: (leavingRow 0-based, entering variable, leaving variable) per iteration. */
int lpr_revised_log_read(lpr_revised* s, int32_t* row_out, int32_t* enter_out,
                         int32_t* leave_out, int64_t cap, int64_t* count);
/* BInverse (m x m row-major) and the last x_B (m), for snapshots and tests. */
int lpr_revised_binv_read(lpr_revised* s, double* out);
int lpr_revised_xb_read(lpr_revised* s, double* out);

/* Replaces `MultiplyMatrices(BInverse, A)` of CaptureSnapshot (:360, helper :426-441, zero-skip
 * |b_ik| < 1e-9): the m x n product B^-1 * A, computed on the fp64 matrix cores
 * (v_mfma_f64_16x16x4_f64).  out (m x n row-major) may be NULL to keep the product on the device
 * (benchmarking); *ms, if not NULL, receives the kernel time in milliseconds (HIP events).
 * This product only feeds the printed tableau, so its contract is a tolerance
 * (|err| <= 1e-9 * sum_k |b_ik a_kj|), not bit equality -- DESIGN.md. */
int lpr_revised_binv_a(lpr_revised* s, double* out, double* ms);

/* ---- IterationSnapshots of the revised solver (RevisedPrimalSimplexSolver.cs:36, consumed by
 * Program.cs:329-347).  The C# appends one text block per iteration (CaptureSnapshot :294-387); the
 * numbers in it come from the device through the three calls below, the host only formats them
 * (NumFormat.N3 :451-466).  Meant for the models a person reads (snapshot policy "all"); a solve
 * without snapshots is lpr_revised_solve. */
typedef struct lpr_revised_snapshot_info {
    int32_t status;        /* LPR_PIVOT_LIMIT: one pivot done, the loop goes on ("Iteration k"
                              snapshot, :232-247); LPR_OK_OPTIMAL: the "Optimal" snapshot
                              (:124-146); any other status: the C# threw, no snapshot */
    int32_t entering;      /* enteringIdx (-1 on the Optimal snapshot) */
    int32_t leaving_row;   /* leavingRow, 0-based basis row */
    int32_t leaving_var;   /* leavingVarIndex_Pre */
    double entering_rc_pre;  /* :189-191 */
    double z_working;      /* Dot(cB, xB) :245 / :141 */
    double z_original;     /* ComputeOriginalZFromCurrentBasis(xB) :246 / finalZ :142 */
} lpr_revised_snapshot_info;

/* ONE pass of the while-loop of Solve() (:86-249), including the post-pivot recomputation of
 * x_B, y and the reduced costs (:217-227) that the snapshot prints.  Returns info->status.
 * Can be mixed freely with lpr_revised_solve on the same handle. */
int lpr_revised_step(lpr_revised* s, lpr_revised_snapshot_info* info);
/* What the last lpr_revised_step left: y (m), rc (n + m: rcX then rcS = -y), u_pre (m), ratios_pre
 * (m, +inf where u_i <= EPS), basisForRatios_Pre (m), x_B (m).  Any pointer may be NULL.  After
 * an "Optimal" step u / ratios / basis_pre are those of the previous pivot -- the C# prints zeros
 * and infinities there (:133-134). */
int lpr_revised_snapshot_read(lpr_revised* s, double* y, double* rc, double* u, double* ratios,
                              int32_t* basis_pre, double* xB);
/* MultiplyMatrices(BInverse, A) (:360) in the C#'s own summation order (k ascending, |a_ik| < EPS
 * skipped, product rounded then added) -- bit-exact, for the printed table of small models, where
 * a 3-decimal tie must round as the C# rounds it.  out: m x n row-major. */
int lpr_revised_binv_a_exact(lpr_revised* s, double* out);

/* ------------------------------------------------------- branch and bound */

typedef struct lpr_bb lpr_bb;

/* Replaces BranchAndBoundAdapter.SolveFromPrimal's set-up (IntegerProgramming/
 * BranchAndBoundAdapter.cs:9-24): Convert(primal.FinalTableau) (:31-46) becomes the root node
 * (node id 0) in HBM, SetNumVars(nvars) (:20).  max_depth bounds the branching depth (every level
 * adds one row and one column; node buffers are sized for it; <= 0: 64).
 *   lpr_bb_create               from a host double[,] (rows x cols row-major)
 *   lpr_bb_create_from_tableau  from a solved device tableau -- no host round trip */
int lpr_bb_create(lpr_engine* e, const double* final_tableau, int rows, int cols, int nvars,
                  int max_depth, lpr_bb** out);
int lpr_bb_create_from_tableau(lpr_tableau* t, int nvars, int max_depth, lpr_bb** out);
int lpr_bb_destroy(lpr_bb* b);

typedef struct lpr_bb_opts {
    int32_t enable_pruning; /* ShouldPrunebranch (:985-1004); Program.cs:389 passes false */
    int32_t node_cap;       /* <= 0: 20, the reference's hard stop (:1038-1042) */
    int32_t reserved[2];
} lpr_bb_opts;

typedef struct lpr_bb_result {
    int32_t status;        /* LPR_OK_OPTIMAL (stack emptied) or LPR_BB_NODE_CAP */
    int32_t found;         /* 0: "No integer solution found" -> the C# returns (null, -inf) */
    int64_t processed;     /* branchCount (:1045) */
    int32_t best_node;     /* record id of the optimal branch, -1 if none */
    int32_t reserved;
    double z;              /* optimalValue (rounded to 4 decimals like every B&B value) */
    int64_t pivots;        /* dual + primal pivots over all child LPs */
    int64_t nodes_created; /* node records (root + every child attempted) */
} lpr_bb_result;

/* Replaces BranchAndBound.ExecuteBranchAndBound (BranchBoundSimplexSolver.cs:1006-1233) in the
 * reference's own order: DFS stack, lower child first, incumbent replaced on strict improvement,
 * 4-decimal Math.Round between all stages.  Per popped node the two children (AddConstraint
 * :694-803, DoDualSimplex :289-468 = PerformDualPivot :115-201 then PerformPrimalPivot :203-279)
 * are evaluated as one batch on the device.  x (nvars entries) is written when found. */
int lpr_bb_run(lpr_bb* b, const lpr_bb_opts* opts, double* x, lpr_bb_result* res);

/* Node records of the last lpr_bb_run, in creation order (record 0 = root): parent record, kind
 * (0 root / 1 lower / 2 upper), depth, branching variable, bound, status (0 solved, 1 infeasible
 * = DoDualSimplex returned a null optimum, 2 failed = an exception escaped it), rounded z. */
int lpr_bb_records_read(lpr_bb* b, int32_t* parent, int32_t* kind, int32_t* depth, int32_t* var,
                        double* bound, int32_t* status, double* z, int64_t cap, int64_t* count);
/* Record ids in the order the nodes were popped (processed). */
int lpr_bb_pop_order_read(lpr_bb* b, int32_t* ids, int64_t cap, int64_t* count);
/* Pivot trace of the last run: quads (record id, phase 0 dual / 1 primal / 2 "last tableau
 * dropped" :395-400, row, col). */
int lpr_bb_trace_read(lpr_bb* b, int32_t* quads, int64_t cap, int64_t* count);

/* Building blocks of the same path for callers that drive the tree themselves (the
 * level-synchronous multi-GPU driver shards the frontier over ranks and calls these per level):
 *   lpr_bb_node_info  RoundAllTableaux on pop (:1047) + GetObjective (:892-897) + the decision
 *                     values of CheckIntegerBasicVar / ExtractSolution (:807-827, :899-921);
 *                     z_out[count], vals_out[count * nvars]
 *   lpr_bb_expand     for each (parent, var, bound, kind 0 "<=" / 1 ">="): AddConstraint +
 *                     DoDualSimplex + RoundAllTableaux, all children in one batch;
 *                     status_out: 2 solved (child_ids_out = new node id), 3 infeasible, 4 failed
 *   lpr_bb_release    frees node buffers
 *   lpr_bb_node_read  copies a node tableau to the host (tests / snapshots) */
int lpr_bb_node_info(lpr_bb* b, const int32_t* ids, int count, double* z_out, double* vals_out);
int lpr_bb_expand(lpr_bb* b, int count, const int32_t* parent_ids, const int32_t* var,
                  const double* bound, const int32_t* kind, int32_t* child_ids_out,
                  int32_t* status_out, int32_t* pivots_out);
/* lpr_bb_expand for a caller that also reproduces what the reference PRINTS about each child
 * (ExecuteBranchAndBound :1086-1208): the pivots of DoDualSimplex ("pivot @ constraint r, column c",
 * :198 / :276) and every tableau of its list (:292, :341, :388) for DisplayTableau (:623-640) --
 * the one AddConstraint hands it, then one per pivot; a last tableau dropped by :395-400 is dropped
 * here too.  Small models only: the tableaux are copied off the device after every pivot step.
 *   trace_out: (phase 0 dual / 1 primal / 2 "last tableau dropped", row, col) triples of all
 *     children, child k's are [trace_off_out[k], trace_off_out[k + 1]) (count + 1 offsets);
 *   tab_out: child k's ntab_out[k] tableaux, each (parent rows + 1) x (parent cols + 1) row-major,
 *     from tab_off_out[k] (doubles; count + 1 offsets).  Buffers too small: LPR_BAD_ARGUMENT, the
 *     offsets still say how much is needed. */
int lpr_bb_expand_traced(lpr_bb* b, int count, const int32_t* parent_ids, const int32_t* var,
                         const double* bound, const int32_t* kind, int32_t* child_ids_out,
                         int32_t* status_out, int32_t* pivots_out, int32_t* trace_out,
                         int64_t trace_cap, int64_t* trace_off_out, double* tab_out, int64_t tab_cap,
                         int64_t* tab_off_out, int32_t* ntab_out);
int lpr_bb_release(lpr_bb* b, const int32_t* ids, int count);
int lpr_bb_node_read(lpr_bb* b, int32_t id, double* out, int32_t* rows, int32_t* cols);

/* ---- multi-GPU Branch & Bound: one process per GPU, sub-trees sharded over the ranks, ONE
 * all-reduce(MAX) of the incumbent bound per level over RCCL / xGMI.  The reference
 * loop being sharded is ExecuteBranchAndBound (BranchBoundSimplexSolver.cs:1006-1233) with its
 * 20-node stop lifted; with pruning off (Program.cs:389) the explored tree, and so the answer,
 * does not depend on the number of ranks. */
typedef struct lpr_comm lpr_comm;
#define LPR_COMM_ID_BYTES 128
/* Rank 0 makes the id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by whatever
 * channel the host has (a file, a socket, its launcher's environment); then every rank calls
 * lpr_comm_init with its engine (= its GPU): ncclCommInitRank. */
int lpr_comm_unique_id(uint8_t id[LPR_COMM_ID_BYTES]);
int lpr_comm_init(lpr_engine* e, int rank, int world, const uint8_t id[LPR_COMM_ID_BYTES],
                  lpr_comm** out);
/* The same with the caller's own transport instead of RCCL.  Both callbacks return 0 on success
 * and are invoked on the calling thread only: all_reduce_max(user, v, n) leaves in v[0..n) the
 * element-wise maximum over all ranks; all_gather(user, send, recv, bytes) leaves in
 * recv[world * bytes] the `bytes` of every rank in rank order. */
typedef int (*lpr_allreduce_max_fn)(void* user, double* inout, int count);
typedef int (*lpr_allgather_fn)(void* user, const void* send, void* recv, int bytes);
int lpr_comm_init_custom(int rank, int world, lpr_allreduce_max_fn all_reduce_max,
                         lpr_allgather_fn all_gather, void* user, lpr_comm** out);
int lpr_comm_destroy(lpr_comm* c);
/* rank, world and how many collectives this communicator has issued so far */
int lpr_comm_info(const lpr_comm* c, int* rank, int* world, int64_t* allreduce_calls,
                  int64_t* allgather_calls);
/* inout[0..count) <- maximum over all ranks (count <= 64); the collective the B&B levels use */
int lpr_comm_all_reduce_max(lpr_comm* c, double* inout, int count);

typedef struct lpr_bb_sync_opts {
    int32_t enable_pruning;  /* ShouldPrunebranch (:985-1004) against the all-reduced bound */
    int32_t max_levels;      /* <= 0: max_depth of the handle */
    int64_t max_nodes;       /* stop once a rank has processed more than this (<= 0: 2^20) */
} lpr_bb_sync_opts;

typedef struct lpr_bb_sync_result {
    int32_t status;      /* LPR_OK_OPTIMAL; LPR_BB_NODE_CAP when max_nodes stopped it; LPR_BB_DEPTH_CAP
                          * when max_levels did (the nodes of that depth are still scored); < 0: a
                          * rank failed and EVERY rank returns that status from the same level */
    int32_t found;       /* 0: no integer solution */
    int64_t processed;   /* nodes scored, all ranks */
    int64_t pivots;      /* dual + primal pivots of all child LPs, all ranks */
    int32_t levels;      /* levels branched (depth reached).  All-reduces issued = levels, + 1 when a
                          * scoring-only pass over the depth-max_levels frontier followed */
    int32_t path_len;    /* the winner's branch path: path_len sides from the root ... */
    uint64_t path_bits;  /* ... bit k = side taken at depth k (0 lower "<= floor", 1 upper) */
    double z;            /* incumbent objective (rounded to 4 decimals like every B&B value) */
} lpr_bb_sync_result;

/* Level-synchronous form of ExecuteBranchAndBound (:1006-1233): every level scores this rank's
 * frontier (IsInteger :595-599, CheckIntegerBasicVar :829-847), branches (CreateBranches
 * :859-890), evaluates ALL children in one batch (AddConstraint + DoDualSimplex on the device)
 * and issues ONE all-reduce(MAX) of {incumbent z, "someone has nodes left", "someone hit
 * max_nodes", "someone failed" (its lpr_status), "someone left nodes unbranched at max_levels"}.
 * A rank-local failure (device error, out of memory, a cycling child's pivot limit) is never
 * returned before that collective: it rides in it, so all ranks return it together instead of the
 * healthy ones waiting for ever (the per-branch catch of :1145-1148,1205-1208 is the reference's
 * analogue).  Every rank must pass the same opts and handles of the same max_depth.  The first ceil(log2(world)) levels are done by every rank alike; the frontier at
 * that depth is dealt round robin in DFS order and a sub-tree then stays on its rank -- tableaux
 * never move.  With pruning off (the reference's setting, Program.cs:389) ties on z go to the node
 * the reference's stack pops first; with enable_pruning a tied integer node that the DFS would
 * have met BEFORE the incumbent may be pruned here (same z, possibly another x).  One all-gather
 * at the end names the winner.  Test hook: the environment variable LPR_BB_INJECT_FAULT=
 * "<rank>:<level>" makes that rank fail at that level.  comm == NULL: a single rank.  Every rank passes a handle holding the
 * same root; x (nvars) is written on every rank when found. */
int lpr_bb_solve_level_sync(lpr_bb* b, lpr_comm* comm, const lpr_bb_sync_opts* opts, double* x,
                            lpr_bb_sync_result* res);

/* ------------------------------------------------------------------------------------------
 * Sensitivity re-solve ("next" row f4): SensitivityAnalysis/SensitivityAnalyzer.cs.
 * The analyzer owns a device copy of the final tableau (the C# clones it, :24).  The console
 * prompts of the C# become arguments; every edit reports an lpr_sens_outcome through *outcome
 * (the function's own return value is an lpr_status: < 0 only for API / device failures).
 * Range displays, shadow prices and the duality print-out are read-only arithmetic on a few rows
 * and columns: the host mirror does them from lpr_sens_read_block / lpr_sens_column_fold. */
typedef struct lpr_sens lpr_sens;

enum lpr_sens_outcome {
    LPR_SENS_OK = 0,                 /* re-solved; z / solution vector refreshed (:159-165)      */
    LPR_SENS_UNBOUNDED = 1,          /* "Unbounded during re-optimization." :151                 */
    LPR_SENS_INFEASIBLE = 2,         /* "Infeasible after RHS change (dual simplex)." :197       */
    LPR_SENS_ZERO_PIVOT = 3,         /* "Zero pivot encountered." :101                           */
    LPR_SENS_ITER_LIMIT = 5,         /* "Iteration limit in ReOptimize / dual simplex" :126 :183 */
    LPR_SENS_ROLLED_BACK = 8,        /* ChangeRHS caught an exception and restored (:462-469)    */
    LPR_SENS_INDEX_OUT_OF_RANGE = 9, /* AddNewConstraint: a row without a basic column (tech[-1]) */
    LPR_SENS_INVALID_INDEX = -1      /* the C# prints "Invalid ..." and returns; nothing changed */
};

/* ctor :22-39 (basicVariables is rebuilt from the tableau by :35, so it is not an argument).
 * final_tableau is rows x cols row-major with row 0 = Z row; cols >= rows. */
int lpr_sens_create(lpr_engine* e, const double* final_tableau, int32_t rows, int32_t cols,
                    const double* solution, int32_t nsol, double z, lpr_sens** out);
/* Program.cs:147-151: primalSolver.GetFinalTableau() / SolutionVector / FinalZ handed over device
 * to device from a solved primal tableau (n_decision = objective.Count). */
int lpr_sens_create_from_tableau(lpr_tableau* t, int32_t n_decision, lpr_sens** out);
int lpr_sens_destroy(lpr_sens* s);
/* numRows, numCols, solutionVector.Count, basicVars.Count, CurrentZ :728, pivots of the last edit */
int lpr_sens_shape(lpr_sens* s, int32_t* rows, int32_t* cols, int32_t* nsol, int32_t* nbasic,
                   double* z, int64_t* last_pivots);
/* CurrentTableau :727 (rows x cols, dense), basicVars, CurrentSolutionVector :729; any may be NULL */
int lpr_sens_read(lpr_sens* s, double* tableau, int32_t* basic, double* solution);
int lpr_sens_read_block(lpr_sens* s, int32_t row0, int32_t nrows, int32_t col0, int32_t ncols,
                        double* out);
int lpr_sens_basic_row(lpr_sens* s, int32_t col, int32_t* row);   /* GetBasicRow :69-77 */
/* (kind 0 dual / 1 primal, leaveRow, enterCol) of every pivot since creation; *count = total */
int lpr_sens_log_read(lpr_sens* s, int32_t* triples, int64_t cap, int64_t* count);
/* out[j] = (init ? init[j] : 0) + sum_{i<nw, in order} w[i] * tableau[i+1, j], j < ncols; nw must
 * be rows-1.  The order-faithful column sums of :636-645 and PerformDuality :690-694. */
int lpr_sens_column_fold(lpr_sens* s, const double* w, int32_t nw, const double* init,
                         int32_t ncols, double* out);
int lpr_sens_resolve_all(lpr_sens* s, int32_t* outcome);                       /* :203-208 */
/* ChangeNonBasicReducedCost :300-321 (index 0-based, new_cbar is the new Z-C entry) */
int lpr_sens_change_nonbasic_cbar(lpr_sens* s, int32_t index, double new_cbar, int32_t* outcome);
int lpr_sens_change_basic(lpr_sens* s, int32_t col, double delta, int32_t* outcome); /* :362-393 */
/* ChangeRHS :427-470 (k = constraint 1..rows-1, new_b replaces the CURRENT tableau RHS) */
int lpr_sens_change_rhs(lpr_sens* s, int32_t k, double new_b, int32_t* outcome);
int lpr_sens_change_nonbasic_column(lpr_sens* s, int32_t row, int32_t col, double new_val,
                                    int32_t* outcome);                          /* :502-531 */
int lpr_sens_add_activity(lpr_sens* s, double c_new, const double* a_new, int32_t na,
                          int32_t* outcome);                                    /* :534-584 */
/* AddNewConstraintNonInteractive :609-659 (tech has cols-1 entries, <= row) */
int lpr_sens_add_constraint(lpr_sens* s, const double* tech, int32_t ntech, double rhs,
                            int32_t* outcome);

#ifdef __cplusplus
}
#endif
#endif /* LPR_ENGINE_H */
