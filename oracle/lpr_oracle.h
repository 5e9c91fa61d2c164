/*
 * lpr_oracle.h -- CPU restatement of the reference's dense-tableau simplex path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lpr_381_group_v22_amd/ or include/ may include, link,
 * import or execute anything in this directory; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do, and there only as the checker / the CPU timing baseline.
 *
 * PARITY UNPINNED BY THE REFERENCE: Storm-Tarran/LPR_381_Group_V22 ships no tests, no golden
 * outputs (data/output_results.txt is 0 bytes) and cannot be compiled here (C# / .NET Framework
 * 4.7.2, no toolchain in the image; the project also references types that do not exist,
 * Program.cs:444,468).  What pins this oracle instead: (1) it is a literal, loop-for-loop
 * restatement of the cited C# lines, compiled with -ffp-contract=off and no fast-math; (2) an
 * independently written Python restatement (tests/ref_py.py) must agree with it bit-for-bit on
 * pivot logs, bases and result bits; (3) the hand-traced results in SURVEY.md section 4;
 * (4) scipy.optimize.linprog objective values on well-posed LPs.
 *
 * Floating point model: IEEE binary64, round-to-nearest-even, every C# operator one rounding
 * (x64 RyuJIT / SSE2 semantics; an x87 32-bit JIT could differ -- DESIGN.md).
 */
#ifndef LPR_ORACLE_H
#define LPR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same numeric values as include/lpr_engine.h (tests assert this) */
enum {
    ORC_OK_OPTIMAL = 0,
    ORC_UNBOUNDED = 1,
    ORC_INFEASIBLE_BASIS = 2,
    ORC_PIVOT_TOO_SMALL = 3,
    ORC_ENTERING_ALREADY_BASIC = 4,
    ORC_PIVOT_LIMIT = 5,
    ORC_BB_NODE_CAP = 6,
    ORC_BAD_ARGUMENT = -1
};

/* ---- synthetic LP generator (spec in DESIGN.md; the device implements the same function) ---- */
uint64_t orc_splitmix64(uint64_t x);
double orc_u01(uint64_t seed, uint64_t stream, uint64_t i, uint64_t j);
/* c[n], A[m*n] row-major, b[m] */
void orc_gen_dense_lp(int m, int n, uint64_t seed, double* c, double* A, double* b);
/* the (m+1) x (n+m+1) tableau of that LP, built directly (for the CPU baseline at full size) */
void orc_gen_dense_tableau(int m, int n, uint64_t seed, double* T, int32_t* basis);

/* ---- PrimalSimplexSolver (Simplex/PrimalSimplexSolver.cs) ---- */
/* ctor :27-87.  T is (m+1) x (n+m+1) row-major, basis has m entries. */
int orc_primal_build(int n, int m, const double* objective, const double* A, int lda,
                     const int32_t* ncoef, const int8_t* relation, const double* rhs, int is_max,
                     double* T, int32_t* basis);
int orc_find_entering(const double* T, int R, int C);          /* :152-167 */
int orc_find_leaving(const double* T, int R, int C, int e);    /* :169-191 */
void orc_pivot(double* T, int R, int C, int r, int e);         /* :193-211 */
/* Solve :102-150.  log_rows/log_cols may be NULL.  Returns status. */
int orc_primal_solve(double* T, int R, int C, int32_t* basis, int64_t max_pivots,
                     int32_t* log_rows, int32_t* log_cols, int64_t log_cap, int64_t* pivots);
void orc_extract_solution(const double* T, int R, int C, int n, double* x, double* z); /* :213-252 */

/* ---- RevisedPrimalSimplexSolver (Simplex/RevisedPrimalSimplexSolver.cs) ---- */
/* MultiplyMatrices :426-441 (zero-skip |a_ik| < 1e-9): R(rA x cB) = A(rA x cA) * B(cA x cB).
 * With (BInverse, A) this is the snapshot product of :360 -- BASELINE's "B^-1 * A". */
void orc_matmul_skip(const double* A, int rA, int cA, const double* B, int cB, double* R);
/* UpdateBInverse :264-275 (scratch: m*m doubles).  0 or ORC_PIVOT_TOO_SMALL. */
int orc_update_binverse(double* Binv, int m, int pivotRow, const double* u, double* scratch);
/* CaptureSnapshot (:294-387) as numbers, one record per snapshot; layout in oracle_revised.c */
int64_t orc_revised_trace_stride(int n, int m);
int orc_revised_solve_trace(int n, int m, const double* objective, const double* A,
                            const double* b, int is_min, int64_t max_iter, double* x,
                            double* finalZ, int32_t* basis, double* trace, int64_t cap,
                            int64_t* snapshots, int64_t* iterations);
/* ctor :41-80 + Solve :82-251 + ExtractSolution :277-287; see oracle_revised.c */
int orc_revised_solve(int n, int m, const double* objective, const double* A, const double* b,
                      int is_min, int64_t max_iter, double* x, double* finalZ, int32_t* basis,
                      double* Binv_out, double* xB_out, int32_t* log_row, int32_t* log_enter,
                      int32_t* log_leave, int64_t log_cap, int64_t* iterations);

/* ONE pass of Solve()'s loop body (:89-215) from a given (B^-1, basis); see oracle_revised.c */
int orc_revised_iterate_from(int n, int m, const double* objective, const double* A,
                             const double* b, int is_min, double* Binv, int32_t* basis,
                             double* xB_out, double* y_out, double* rcX_out, double* rcS_out,
                             double* u_out, double* ratios_out, int32_t* entering,
                             int32_t* leaving_row);

/* ---- Branch & Bound (IntegerProgramming/BranchBoundSimplexSolver.cs, BranchAndBoundAdapter.cs) ---- */
double orc_round_int(double x); /* .NET Framework Math.Round(double)     */
double orc_round4(double x);    /* .NET Framework Math.Round(double, 4)  */
/* BranchAndBoundAdapter.SolveFromPrimal + ExecuteBranchAndBound; see oracle_bb.c for the outputs */
int orc_bb_solve(const double* final_tableau, int rows, int cols, int nvars, int enable_pruning,
                 int node_cap, double* x, double* z, int* found, int* best_node,
                 int* processed, int32_t* rec_parent, int32_t* rec_kind, int32_t* rec_depth,
                 int32_t* rec_var, double* rec_bound, int32_t* rec_status, double* rec_z,
                 int node_rec_cap, int* n_records, int32_t* pop_order, int32_t* piv_trace,
                 int64_t piv_cap, int64_t* n_piv);
/* AddConstraint :694-803 for one constraint (coefficients..., bound, type); out is (rows+1)x(cols+1) */
int orc_bb_add_constraint(const double* base, int rows, int cols, const double* con, int conLen,
                          double* out);
/* RoundTableau :552-567 in place; a popped node's RoundAllTableaux :1047 + GetObjective :892-897 +
 * decision values :805-857 / :899-921 (T rounded in place) */
void orc_bb_round_tableau(double* T, int rows, int cols);
void orc_bb_node_info(double* T, int rows, int cols, int nvars, double* z, double* vals);
/* DoDualSimplex :289-468 (tableauOverride mode): 0 solved, 1 infeasible, 2 exception escaped */
int orc_bb_dual_simplex(const double* start, int rows, int cols, double* out, int* npiv,
                        int32_t* piv_trace, int64_t piv_cap, int64_t* n_piv);

/* ---- cutting-plane side path: Simplex/DualSimplex.cs, Simplex/PrimalSimplexSolver2.cs,
 *      IntegerProgramming/CuttingPlaneSolver.cs (see oracle_cut.c).  T: row 0 = objective row.
 *      Return codes of the two solvers: 0 true / 1 false (infeasible resp. unbounded) /
 *      3 pivot too small (exception) / 5 iteration limit.  Log triples (kind, row, col). ---- */
int orc_dual_solve(double* T, int R, int C, int max_iters, int print_steps, int64_t hard_cap,
                   int32_t* log, int64_t log_cap, int64_t* n_log, int64_t* pivots);
int orc_primal2_solve(double* T, int R, int C, int max_iters, int print_steps, int64_t hard_cap,
                      int32_t* log, int64_t log_cap, int64_t* n_log, int64_t* pivots);
int orc_cutting_plane(double* T, int* R_io, int R_cap, int C, int max_cuts, int64_t hard_cap,
                      int32_t* log, int64_t log_cap, int64_t* n_log, int* cuts);

/* ---- sensitivity re-solve (SensitivityAnalysis/SensitivityAnalyzer.cs; see oracle_sens.c).
 *      Return codes: 0 ok, 1 unbounded, 2 infeasible, 3 zero pivot, 5 iteration limit,
 *      8 ChangeRHS rolled back, 9 IndexOutOfRange in AddNewConstraint, -1 invalid index. ---- */
typedef struct orc_sens orc_sens;
orc_sens* orc_sens_create(const double* finalTableau, int R, int C, const double* solution,
                          int nsol, double z, const int32_t* basic, int nbasic);
void orc_sens_destroy(orc_sens* s);
void orc_sens_shape(const orc_sens* s, int* R, int* C, int* nsol, int* nbasic, double* z);
void orc_sens_read(const orc_sens* s, double* T, int32_t* basic, double* sol);
/* (kind 0 dual / 1 primal, leaveRow, enterCol) of every pivot so far; returns the total count */
int64_t orc_sens_log_read(const orc_sens* s, int32_t* triples, int64_t cap);
int orc_sens_resolve_all(orc_sens* s);
int orc_sens_change_nonbasic_cbar(orc_sens* s, int index, double newCbar);
int orc_sens_change_basic(orc_sens* s, int col, double delta);
int orc_sens_change_rhs(orc_sens* s, int k, double newB);
int orc_sens_change_nonbasic_column(orc_sens* s, int row, int col, double newVal);
int orc_sens_add_activity(orc_sens* s, double cNew, const double* aNew);
int orc_sens_add_constraint(orc_sens* s, const double* tech, int ntech, double rhs);

#ifdef __cplusplus
}
#endif
#endif
