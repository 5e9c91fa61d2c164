/*
 * oracle_cut.c -- literal CPU restatement of the cutting-plane side path ("next" row f3):
 *   DualSimplexSolver        (reference: LPR_381_Group_V22/Simplex/DualSimplex.cs)
 *   PrimalSimplexSolver2     (reference: LPR_381_Group_V22/Simplex/PrimalSimplexSolver2.cs)
 *   CuttingPlaneSolver       (reference: LPR_381_Group_V22/IntegerProgramming/CuttingPlaneSolver.cs)
 *
 * TEST INFRASTRUCTURE ONLY -- see lpr_oracle.h.  PARITY UNPINNED by the reference (no tests; the
 * menu never reaches this code, Program.cs:417-428); pinned by tests/ref_py_cut.py.
 *
 * Layout: one (rows x cols) row-major tableau, row 0 = objectiveRow, rows 1.. = constraintRows
 * (the C# keeps them as double[] + List<double[]>; PrimalSimplexSolver2 copies them into exactly
 * this double[,]).  Row indices in logs are CONSTRAINT indices as the C# uses them: 0-based for
 * DualSimplexSolver / the cut, 1-based tableau rows for PrimalSimplexSolver2.
 *
 * Quirks restated as they are:
 *   - `iter` only advances when printSteps is set, so maxIters is inert in silent runs
 *     (DualSimplex.cs:94,108; PrimalSimplexSolver2.cs:75,90); `print_steps` selects the behaviour.
 *   - the EPS-band comparators whose tie clause can never fire in ascending index order
 *     (DualSimplex.cs:32; PrimalSimplexSolver2.cs:110) and the operator-precedence accident of
 *     PrimalSimplexSolver2.cs:132-133 (`a && b == -1 ? true : i < bestRow`).
 *   - rows whose factor is within EPS of zero are skipped by the elimination (not subtracted).
 *   - the Gomory cut is appended WITHOUT a slack column (CuttingPlaneSolver.cs:104-110).
 *   - List<T>.Sort (:94) is an unstable introsort; for up to 16 elements it is an insertion sort
 *     (stable), which is what is restated: the first row, in index order, among those closest to
 *     0.5.  With more than 16 fractional rows AND an exact tie the .NET choice is
 *     implementation-defined and this oracle may differ.
 */
#include "lpr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPS 1e-9
#define AT(T, C, i, j) ((T)[(size_t)(i) * (C) + (j)])

enum { CUT_OK = 0, CUT_FAIL = 1, CUT_PIVOT_TOO_SMALL = 3, CUT_LIMIT = 5 };

typedef struct {
    int32_t* buf; /* triples (kind, row, col): kind 0 dual, 1 primal2, 2 cut pivot */
    int64_t cap, n;
} CutLog;
static void cl_push(CutLog* l, int kind, int row, int col) {
    if (l && l->buf && l->n < l->cap) {
        l->buf[3 * l->n] = kind;
        l->buf[3 * l->n + 1] = row;
        l->buf[3 * l->n + 2] = col;
    }
    if (l) l->n++;
}

/* Pivot with the |f| <= EPS row skip: DualSimplex.cs:150-178, PrimalSimplexSolver2.cs:145-164,
 * CuttingPlaneSolver.cs:145-176.  pr is a TABLEAU row.  Returns 0 or CUT_PIVOT_TOO_SMALL. */
static int pivot_skip(double* T, int R, int C, int pr, int pc) {
    double* prow = T + (size_t)pr * C;
    double piv = prow[pc];
    if (fabs(piv) <= EPS) return CUT_PIVOT_TOO_SMALL;
    for (int j = 0; j < C; j++) prow[j] /= piv;
    /* the C# eliminates the constraint rows first and the objective last; rows are independent,
     * so the order does not change any value */
    for (int i = 0; i < R; i++) {
        if (i == pr) continue;
        double* row = T + (size_t)i * C;
        double f = row[pc];
        if (fabs(f) > EPS)
            for (int j = 0; j < C; j++) {
                double prod = f * prow[j];
                row[j] = row[j] - prod;
            }
    }
    return 0;
}

/* DualSimplexSolver.Solve, DualSimplex.cs:14-114.  Returns CUT_OK (true), CUT_FAIL (false:
 * infeasible), CUT_LIMIT (false: max iterations, only reachable with print_steps) or
 * CUT_PIVOT_TOO_SMALL (the InvalidOperationException of :155-156). */
static int dual_solve(double* T, int R, int C, int max_iters, int print_steps, int64_t hard_cap,
                      CutLog* log, int64_t* pivots) {
    int iter = 0;
    int64_t done = 0;
    for (;;) {
        int pivotRow = -1; /* constraint index */
        double mostNeg = 0.0;
        for (int r = 0; r < R - 1; r++) { /* :29-37 */
            double rhs = AT(T, C, r + 1, C - 1);
            if (rhs < mostNeg - EPS || (fabs(rhs - mostNeg) <= EPS && pivotRow != -1 && r < pivotRow)) {
                mostNeg = rhs;
                pivotRow = r;
            }
        }
        if (pivotRow == -1) { if (pivots) *pivots = done; return CUT_OK; } /* :40-44 */
        int pivotCol = -1;
        double bestRatio = INFINITY;
        for (int j = 0; j < C - 1; j++) { /* :53-70 */
            double a = AT(T, C, pivotRow + 1, j);
            if (a < -EPS) {
                double num = AT(T, C, 0, j);
                if (fabs(num) > EPS) {
                    double ratio = fabs(num / a);
                    if (ratio < bestRatio - EPS ||
                        (fabs(ratio - bestRatio) <= EPS && (pivotCol == -1 || j < pivotCol))) {
                        bestRatio = ratio;
                        pivotCol = j;
                    }
                }
            }
        }
        if (pivotCol == -1) { if (pivots) *pivots = done; return CUT_FAIL; } /* :72-76 */
        if (hard_cap > 0 && done >= hard_cap) { if (pivots) *pivots = done; return CUT_LIMIT; }
        if (print_steps) ++iter; /* :94 */
        cl_push(log, 0, pivotRow, pivotCol);
        int rc = pivot_skip(T, R, C, pivotRow + 1, pivotCol); /* :98 */
        if (rc) { if (pivots) *pivots = done; return rc; }
        done++;
        if (iter >= max_iters) { if (pivots) *pivots = done; return CUT_LIMIT; } /* :108-112 */
    }
}

/* PrimalSimplexSolver2.Solve, PrimalSimplexSolver2.cs:46-97.  CUT_OK (optimal), CUT_FAIL
 * (unbounded), CUT_LIMIT, CUT_PIVOT_TOO_SMALL. */
static int primal2_solve(double* T, int R, int C, int max_iters, int print_steps, int64_t hard_cap,
                         CutLog* log, int64_t* pivots) {
    int iter = 0;
    int64_t done = 0;
    const int rhs = C - 1;
    for (;;) {
        int pivotCol = -1; /* :102-117 */
        double mostNeg = 0.0;
        for (int j = 0; j < rhs; j++) {
            double c = AT(T, C, 0, j);
            if (c < mostNeg - EPS || (fabs(c - mostNeg) <= EPS && pivotCol != -1 && j < pivotCol)) {
                mostNeg = c;
                pivotCol = j;
            }
        }
        if (pivotCol == -1) { if (pivots) *pivots = done; return CUT_OK; }
        int bestRow = -1; /* :120-141 */
        double bestRatio = INFINITY;
        for (int i = 1; i < R; i++) {
            double a = AT(T, C, i, pivotCol);
            if (a > EPS) {
                double ratio = AT(T, C, i, rhs) / a;
                /* (ratio > EPS && ratio < bestRatio - EPS) ||
                 * ((|ratio - bestRatio| <= EPS && bestRow == -1) ? true : i < bestRow) */
                int second = (fabs(ratio - bestRatio) <= EPS && bestRow == -1) ? 1 : (i < bestRow);
                if ((ratio > EPS && ratio < bestRatio - EPS) || second) {
                    bestRatio = ratio;
                    bestRow = i;
                }
            }
        }
        if (bestRow == -1) { if (pivots) *pivots = done; return CUT_FAIL; }
        if (hard_cap > 0 && done >= hard_cap) { if (pivots) *pivots = done; return CUT_LIMIT; }
        if (print_steps) ++iter; /* :75 */
        cl_push(log, 1, bestRow, pivotCol);
        int rc = pivot_skip(T, R, C, bestRow, pivotCol); /* :79 */
        if (rc) { if (pivots) *pivots = done; return rc; }
        done++;
        if (iter >= max_iters) { if (pivots) *pivots = done; return CUT_LIMIT; } /* :90-95 */
    }
}

int orc_dual_solve(double* T, int R, int C, int max_iters, int print_steps, int64_t hard_cap,
                   int32_t* log, int64_t log_cap, int64_t* n_log, int64_t* pivots) {
    CutLog l = {log, log_cap, 0};
    int rc = dual_solve(T, R, C, max_iters, print_steps, hard_cap, &l, pivots);
    if (n_log) *n_log = l.n;
    return rc;
}

int orc_primal2_solve(double* T, int R, int C, int max_iters, int print_steps, int64_t hard_cap,
                      int32_t* log, int64_t log_cap, int64_t* n_log, int64_t* pivots) {
    CutLog l = {log, log_cap, 0};
    int rc = primal2_solve(T, R, C, max_iters, print_steps, hard_cap, &l, pivots);
    if (n_log) *n_log = l.n;
    return rc;
}

static double frac_part(double a) { /* CuttingPlaneSolver.cs:12-17 */
    double f = a - floor(a);
    if (fabs(f) < EPS || fabs(1 - f) < EPS) return 0.0;
    return f;
}

/*
 * CuttingPlaneSolver.CuttingPlaneSolution, CuttingPlaneSolver.cs:64-229, with its tail recursion
 * (:220) unrolled into a loop.  T holds R rows and has room for R_cap rows (one more row per cut);
 * max_cuts bounds the recursion (<= 0: R_cap - R).
 * Exit codes (which `return` of the C# was taken):
 *   0 "Displayed the Optimal Tableau" :224      1 all RHS integral, no cut needed :87-91
 *   2 no valid pivot column on the cut :134-138  3 pivot too small :146-150
 *   4 dual simplex failed :191                   5 "Cutting-plane step finished" :228
 *   6 max_cuts reached (no C# counterpart)       7 an InvalidOperationException escaped
 */
int orc_cutting_plane(double* T, int* R_io, int R_cap, int C, int max_cuts, int64_t hard_cap,
                      int32_t* log, int64_t log_cap, int64_t* n_log, int* cuts) {
    CutLog l = {log, log_cap, 0};
    if (max_cuts <= 0 || max_cuts > R_cap - *R_io) max_cuts = R_cap - *R_io;
    int ncuts = 0;
    int exitc;
    for (;;) {
        int R = *R_io;
        /* 1-2) fractional rows; the one closest to 0.5, first in index order (see header) */
        int chosen = -1;
        double bestKey = 0;
        for (int i = 0; i < R - 1; i++) {
            double fr = frac_part(AT(T, C, i + 1, C - 1));
            if (fr > EPS) {
                double key = fabs(fr - 0.5);
                if (chosen == -1 || key < bestKey) { chosen = i; bestKey = key; }
            }
        }
        if (chosen == -1) { exitc = 1; break; }
        if (ncuts >= max_cuts) { exitc = 6; break; }
        /* 3-5) cut = -frac(row), appended as a new constraint row */
        for (int j = 0; j < C; j++) AT(T, C, R, j) = -frac_part(AT(T, C, chosen + 1, j));
        *R_io = R + 1;
        R = R + 1;
        ncuts++;
        const int cutRow = R - 1; /* tableau row; constraint index R - 2 */
        /* 6) pivot column on the cut row */
        int pivotCol = -1;
        double bestRatio = INFINITY;
        for (int j = 0; j < C - 1; j++) {
            double a = AT(T, C, cutRow, j);
            if (a < -EPS) {
                double num = AT(T, C, 0, j);
                if (fabs(num) > EPS) {
                    double ratio = fabs(num / a);
                    if (ratio < bestRatio - EPS ||
                        (fabs(ratio - bestRatio) <= EPS && (pivotCol == -1 || j < pivotCol))) {
                        bestRatio = ratio;
                        pivotCol = j;
                    }
                }
            }
        }
        if (pivotCol == -1) { exitc = 2; break; }
        /* 7) pivot on the cut */
        if (fabs(AT(T, C, cutRow, pivotCol)) <= EPS) { exitc = 3; break; }
        cl_push(&l, 2, cutRow - 1, pivotCol);
        pivot_skip(T, R, C, cutRow, pivotCol);
        /* 8) clean-up */
        int needDual = 0, needPrimal = 0;
        for (int i = 1; i < R; i++) if (AT(T, C, i, C - 1) < -EPS) { needDual = 1; break; }
        for (int j = 0; j < C - 1; j++) if (AT(T, C, 0, j) < -EPS) { needPrimal = 1; break; }
        if (needDual) {
            int64_t pv = 0;
            int rc = dual_solve(T, R, C, 10000, 1, hard_cap, &l, &pv); /* printSteps: true :190 */
            if (rc == CUT_PIVOT_TOO_SMALL) { exitc = 7; break; }
            if (rc != CUT_OK) { exitc = 4; break; }
            needPrimal = 0;
            for (int j = 0; j < C - 1; j++) if (AT(T, C, 0, j) < -EPS) { needPrimal = 1; break; }
        }
        if (needPrimal) {
            int64_t pv = 0;
            int rc = primal2_solve(T, R, C, 10000, 1, hard_cap, &l, &pv); /* :199-200 */
            if (rc == CUT_PIVOT_TOO_SMALL) { exitc = 7; break; }
            /* the result of Solve is ignored (:200); GetRows(false) copies the tableau back */
        }
        /* 9) */
        int opt = 1, neg = 0;
        for (int j = 0; j < C - 1; j++) if (AT(T, C, 0, j) < -EPS) { opt = 0; break; }
        for (int i = 1; i < R; i++) if (AT(T, C, i, C - 1) < -EPS) { neg = 1; break; }
        if (opt && !neg) {
            int anyFrac = 0;
            for (int i = 1; i < R; i++) if (frac_part(AT(T, C, i, C - 1)) > EPS) { anyFrac = 1; break; }
            if (anyFrac) continue; /* recursive step :220 */
            exitc = 0;
            break;
        }
        exitc = 5;
        break;
    }
    if (n_log) *n_log = l.n;
    if (cuts) *cuts = ncuts;
    return exitc;
}
