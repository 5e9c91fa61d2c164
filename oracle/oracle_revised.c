/*
 * oracle_revised.c -- literal CPU restatement of RevisedPrimalSimplexSolver
 * (reference: LPR_381_Group_V22/Simplex/RevisedPrimalSimplexSolver.cs).
 *
 * TEST INFRASTRUCTURE ONLY -- see lpr_oracle.h.  PARITY UNPINNED by the reference (no tests, no
 * golden vectors); pinned by tests/ref_py.py + the SURVEY.md section 4 hand trace.
 *
 * Every sum is the C#'s sequential ascending-index loop with the product rounded before the add
 * (`s += a * b`), every comparator is the C#'s EPS-band sequential fold.  The `B` matrix the C#
 * maintains (:201-212) is never read by it and is not kept here.
 */
#include "lpr_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPS 1e-9 /* :12 */

/* MultiplyMatrixVector :398-410 */
static void mat_vec(const double* M, int rows, int cols, const double* v, double* r) {
    for (int i = 0; i < rows; i++) {
        double s = 0;
        for (int j = 0; j < cols; j++) s += M[(size_t)i * cols + j] * v[j];
        r[i] = s;
    }
}

/* MultiplyVectorMatrix :412-424 */
static void vec_mat(const double* v, const double* M, int rows, int cols, double* r) {
    for (int j = 0; j < cols; j++) {
        double s = 0;
        for (int i = 0; i < rows; i++) s += v[i] * M[(size_t)i * cols + j];
        r[j] = s;
    }
}

/* Dot :443-448 */
static double dot(const double* a, const double* b, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

/* MultiplyMatrices :426-441 -- R = A(rA x cA) * B(cA x cB) with the zero-skip on |a_ik| < EPS.
 * This is the `BInvA = MultiplyMatrices(BInverse, A)` of CaptureSnapshot (:360) when called with
 * (BInverse, A), and the `E * BInverse` of UpdateBInverse (:274). */
void orc_matmul_skip(const double* A, int rA, int cA, const double* B, int cB, double* R) {
    memset(R, 0, (size_t)rA * cB * sizeof(double));
    for (int i = 0; i < rA; i++)
        for (int k = 0; k < cA; k++) {
            double aik = A[(size_t)i * cA + k];
            if (fabs(aik) < EPS) continue;
            const double* brow = B + (size_t)k * cB;
            double* rrow = R + (size_t)i * cB;
            for (int j = 0; j < cB; j++) {
                double prod = aik * brow[j];
                rrow[j] = rrow[j] + prod;
            }
        }
}

/* UpdateBInverse :264-275.  E is the identity with column `pivotRow` replaced (:269-272); the
 * product E * BInverse is formed by the same i-k-j loops as MultiplyMatrices, generating E[i,k] on
 * the fly instead of materialising the m x m matrix.  Returns 0, or ORC_PIVOT_TOO_SMALL. */
int orc_update_binverse(double* Binv, int m, int pivotRow, const double* u, double* scratch) {
    double pivot = u[pivotRow];
    if (fabs(pivot) < EPS) return ORC_PIVOT_TOO_SMALL; /* :267 */
    double* R = scratch;
    memset(R, 0, (size_t)m * m * sizeof(double));
    for (int i = 0; i < m; i++) {
        double* rrow = R + (size_t)i * m;
        for (int k = 0; k < m; k++) {
            double aik;
            if (k == pivotRow) aik = (i == pivotRow) ? 1.0 / pivot : -u[i] / pivot; /* :272 */
            else aik = (i == k) ? 1.0 : 0.0;                                       /* :270 */
            if (fabs(aik) < EPS) continue;
            const double* brow = Binv + (size_t)k * m;
            for (int j = 0; j < m; j++) {
                double prod = aik * brow[j];
                rrow[j] = rrow[j] + prod;
            }
        }
    }
    memcpy(Binv, R, (size_t)m * m * sizeof(double));
    return 0;
}

/* .NET Framework Math.Max(double, double): `if (a > b) return a; if (IsNaN(a)) return a; return b;` */
static double dotnet_max0(double v) { return (0.0 > v) ? 0.0 : v; }

/*
 * Solve :82-251 (+ ctor :41-80, ExtractSolution :277-287).
 *   objective[n], A[m*n] row-major, b[m]; every constraint is treated as "<=" (Relation is never
 *   read by the reference).  is_min: c = -cOrig (:51).
 * Outputs (any may be NULL): x[n], *finalZ, basis[m], Binv_out[m*m], xB_out[m];
 *   log_row/log_enter/log_leave: per iteration (leavingRow 0-based, entering var, leaving var).
 * Returns the status; *iterations = completed pivots.
 */
/* One CaptureSnapshot call (:294-387) as numbers: what the C# formats into the text block.
 * Record layout (doubles; indices are exact in a double), S = orc_revised_trace_stride(n, m):
 *   [0] enteringIdx  [1] leavingRow  [2] leavingVarIndex_Pre  [3] enteringRC_pre  [4] zWorking
 *   [5] zOriginal    then y[m], rcX[n], rcS[m], u_pre[m], ratios_pre[m], basisForRatios_Pre[m],
 *   basicVariables (post)[m], xB[m], BInvA[m*n] (MultiplyMatrices(BInverse, A) :360), BInv[m*m]. */
int64_t orc_revised_trace_stride(int n, int m) {
    return 6 + (int64_t)m + n + m + m + m + m + m + m + (int64_t)m * n + (int64_t)m * m;
}

typedef struct {
    double* buf;     /* cap * stride doubles, or NULL: no trace */
    int64_t cap;
    int64_t count;   /* snapshots the C# would have appended (may exceed cap) */
} rev_trace;

static void trace_snapshot(rev_trace* tr, int n, int m, int entering, int leavingRow,
                           int leavingVarPre, double rcPre, double zWorking, double zOriginal,
                           const double* y, const double* rcX, const double* rcS, const double* u,
                           const double* ratios, const int* basisPre, const int* basic,
                           const double* xB, const double* Binv, const double* A) {
    if (!tr || !tr->buf) return;
    if (tr->count < tr->cap) {
        double* r = tr->buf + tr->count * orc_revised_trace_stride(n, m);
        r[0] = entering; r[1] = leavingRow; r[2] = leavingVarPre;
        r[3] = rcPre; r[4] = zWorking; r[5] = zOriginal;
        r += 6;
        memcpy(r, y, sizeof(double) * m); r += m;
        memcpy(r, rcX, sizeof(double) * n); r += n;
        memcpy(r, rcS, sizeof(double) * m); r += m;
        memcpy(r, u, sizeof(double) * m); r += m;
        memcpy(r, ratios, sizeof(double) * m); r += m;
        for (int i = 0; i < m; i++) r[i] = basisPre[i];
        r += m;
        for (int i = 0; i < m; i++) r[i] = basic[i];
        r += m;
        memcpy(r, xB, sizeof(double) * m); r += m;
        orc_matmul_skip(Binv, m, m, A, n, r); /* :360 */
        r += (size_t)m * n;
        memcpy(r, Binv, sizeof(double) * (size_t)m * m);
    }
    tr->count++;
}

static int revised_core(int n, int m, const double* objective, const double* A, const double* b,
                        int is_min, int64_t max_iter, double* x, double* finalZ, int32_t* basis,
                        double* Binv_out, double* xB_out, int32_t* log_row, int32_t* log_enter,
                        int32_t* log_leave, int64_t log_cap, int64_t* iterations, rev_trace* tr);

int orc_revised_solve(int n, int m, const double* objective, const double* A, const double* b,
                      int is_min, int64_t max_iter, double* x, double* finalZ, int32_t* basis,
                      double* Binv_out, double* xB_out, int32_t* log_row, int32_t* log_enter,
                      int32_t* log_leave, int64_t log_cap, int64_t* iterations) {
    return revised_core(n, m, objective, A, b, is_min, max_iter, x, finalZ, basis, Binv_out, xB_out,
                        log_row, log_enter, log_leave, log_cap, iterations, NULL);
}

/* The same solve, also recording every CaptureSnapshot (:232-247 after each pivot, :127-143 at the
 * optimum) into trace[cap * stride]; *snapshots = how many the C# list would hold. */
int orc_revised_solve_trace(int n, int m, const double* objective, const double* A,
                            const double* b, int is_min, int64_t max_iter, double* x,
                            double* finalZ, int32_t* basis, double* trace, int64_t cap,
                            int64_t* snapshots, int64_t* iterations) {
    rev_trace tr;
    tr.buf = trace;
    tr.cap = cap;
    tr.count = 0;
    int st = revised_core(n, m, objective, A, b, is_min, max_iter, x, finalZ, basis, NULL, NULL,
                          NULL, NULL, NULL, 0, iterations, &tr);
    if (snapshots) *snapshots = tr.count;
    return st;
}

/* Entering choice :105-121 over `nonbasic` (nnb indices, any order): the list is sorted
 * (nonBasicVariables.OrderBy(v => v): stable, values distinct) into `sorted`, then the EPS-band
 * fold.  Returns the entering index or -1. */
static int choose_entering(const int* nonbasic, int nnb, int* sorted, int n, const double* rcX,
                           const double* rcS) {
    memcpy(sorted, nonbasic, sizeof(int) * nnb);
    for (int a = 1; a < nnb; a++) { /* insertion sort: the list is almost sorted */
        int v = sorted[a], k = a - 1;
        while (k >= 0 && sorted[k] > v) { sorted[k + 1] = sorted[k]; k--; }
        sorted[k + 1] = v;
    }
    int enteringIdx = -1;
    double bestPosRC = -INFINITY;
    for (int q = 0; q < nnb; q++) {
        int vIdx = sorted[q];
        double rc = (vIdx < n) ? rcX[vIdx] : rcS[vIdx - n];
        if (rc > EPS) {
            if (enteringIdx == -1 || rc > bestPosRC + EPS ||
                (fabs(rc - bestPosRC) <= EPS && vIdx < enteringIdx)) {
                bestPosRC = rc;
                enteringIdx = vIdx;
            }
        }
    }
    return enteringIdx;
}

/* Ratio test :154-176; fills ratios[m] (:161, :174).  Returns the leaving row or -1. */
static int ratio_test(int m, const double* xB, const double* u, const int* basic, double* ratios) {
    int leavingRow = -1;
    double bestRatio = DBL_MAX;
    for (int i = 0; i < m; i++) {
        if (u[i] > EPS) {
            double ratio = xB[i] / u[i];
            ratios[i] = ratio; /* :161 */
            if (ratio < bestRatio - EPS ||
                (fabs(ratio - bestRatio) <= EPS &&
                 (leavingRow == -1 || basic[i] < basic[leavingRow]))) {
                bestRatio = ratio;
                leavingRow = i;
            }
        } else {
            ratios[i] = INFINITY; /* :174 */
        }
    }
    return leavingRow;
}

/* reduced costs :96-102 */
static void reduced_costs(int n, int m, const double* c, const double* A, const double* y,
                          double* col, double* rcX, double* rcS) {
    for (int j = 0; j < n; j++) { /* :96-98  rc = c_j - Dot(y, GetColumn(A, j)) */
        for (int i = 0; i < m; i++) col[i] = A[(size_t)i * n + j];
        rcX[j] = c[j] - dot(y, col, m);
    }
    for (int k = 0; k < m; k++) rcS[k] = -y[k]; /* :100-102 */
}

/* direction :149-151 */
static void direction(int n, int m, const double* A, const double* Binv, int enteringIdx,
                      double* col, double* u) {
    if (enteringIdx < n) {
        for (int i = 0; i < m; i++) col[i] = A[(size_t)i * n + enteringIdx];
        mat_vec(Binv, m, m, col, u);
    } else {
        int k = enteringIdx - n;
        for (int i = 0; i < m; i++) u[i] = Binv[(size_t)i * m + k];
    }
}

/*
 * ONE pass of Solve()'s loop body (:89-215) from a GIVEN state (B^-1, basis) instead of the slack
 * basis: what tests/test_revised_gpu.py compares lpr_revised_step with at m = 4096 once B^-1 has
 * filled in (a whole oracle solve to that point would take hours).  c_B is rebuilt from the basis
 * (:205,211: c_B[i] = c[basic[i]] for structurals, 0 for slacks); the non-basic list is the
 * complement of the basis (only ever iterated sorted, :108).
 *   Binv[m*m], basis[m]: in; updated in place when a pivot is made (status ORC_PIVOT_LIMIT = "one
 *   pivot done, loop continues").  Outputs (pre-pivot, each may be NULL): xB[m], y[m], rcX[n],
 *   rcS[m], u[m], ratios[m], *entering, *leaving_row.
 * Returns ORC_OK_OPTIMAL (no entering variable; nothing modified), ORC_INFEASIBLE_BASIS,
 * ORC_UNBOUNDED, ORC_ENTERING_ALREADY_BASIC, ORC_PIVOT_TOO_SMALL, or ORC_PIVOT_LIMIT.
 */
int orc_revised_iterate_from(int n, int m, const double* objective, const double* A,
                             const double* b, int is_min, double* Binv, int32_t* basis,
                             double* xB_out, double* y_out, double* rcX_out, double* rcS_out,
                             double* u_out, double* ratios_out, int32_t* entering,
                             int32_t* leaving_row) {
    if (n <= 0 || m <= 0 || !Binv || !basis) return ORC_BAD_ARGUMENT;
    int status = ORC_PIVOT_LIMIT;
    double* c = (double*)malloc(sizeof(double) * n);
    double* cB = (double*)calloc(m, sizeof(double));
    double* xB = (double*)calloc(m, sizeof(double));
    double* y = (double*)malloc(sizeof(double) * m);
    double* rcX = (double*)malloc(sizeof(double) * n);
    double* rcS = (double*)malloc(sizeof(double) * m);
    double* col = (double*)malloc(sizeof(double) * m);
    double* u = (double*)calloc(m, sizeof(double));
    double* ratios = (double*)malloc(sizeof(double) * m);
    int* basic = (int*)malloc(sizeof(int) * m);
    int* nonbasic = (int*)malloc(sizeof(int) * (n + m));
    int* sorted = (int*)malloc(sizeof(int) * (n + m));
    char* isb = (char*)calloc((size_t)n + m, 1);
    int nnb = 0, enteringIdx = -1, leavingRow = -1;
    for (int j = 0; j < n; j++) c[j] = is_min ? -objective[j] : objective[j]; /* :51 */
    for (int i = 0; i < m; i++) {
        basic[i] = basis[i];
        isb[basic[i]] = 1;
        cB[i] = basic[i] < n ? c[basic[i]] : 0.0;
        ratios[i] = INFINITY;
    }
    for (int v = 0; v < n + m; v++) if (!isb[v]) nonbasic[nnb++] = v;

    mat_vec(Binv, m, m, b, xB); /* :89 */
    int infeasible = 0;
    for (int i = 0; i < m; i++) if (xB[i] < -EPS) { infeasible = 1; break; } /* :90 */
    if (infeasible) status = ORC_INFEASIBLE_BASIS;
    else {
        vec_mat(cB, Binv, m, m, y); /* :93 */
        reduced_costs(n, m, c, A, y, col, rcX, rcS);
        enteringIdx = choose_entering(nonbasic, nnb, sorted, n, rcX, rcS);
        if (enteringIdx == -1) status = ORC_OK_OPTIMAL;
        else {
            direction(n, m, A, Binv, enteringIdx, col, u);
            leavingRow = ratio_test(m, xB, u, basic, ratios);
            if (leavingRow == -1) status = ORC_UNBOUNDED;
            else if (basic[leavingRow] == enteringIdx) status = ORC_ENTERING_ALREADY_BASIC;
            else {
                double* scratch = (double*)malloc(sizeof(double) * (size_t)m * m);
                int rc2 = orc_update_binverse(Binv, m, leavingRow, u, scratch); /* :215 */
                free(scratch);
                if (rc2 != 0) status = rc2;
                else basis[leavingRow] = enteringIdx; /* :195 */
            }
        }
    }
    if (xB_out) memcpy(xB_out, xB, sizeof(double) * m);
    if (status != ORC_INFEASIBLE_BASIS) {
        if (y_out) memcpy(y_out, y, sizeof(double) * m);
        if (rcX_out) memcpy(rcX_out, rcX, sizeof(double) * n);
        if (rcS_out) memcpy(rcS_out, rcS, sizeof(double) * m);
    }
    if (u_out) memcpy(u_out, u, sizeof(double) * m);
    if (ratios_out) memcpy(ratios_out, ratios, sizeof(double) * m);
    if (entering) *entering = enteringIdx;
    if (leaving_row) *leaving_row = leavingRow;
    free(c); free(cB); free(xB); free(y); free(rcX); free(rcS); free(col); free(u); free(ratios);
    free(basic); free(nonbasic); free(sorted); free(isb);
    return status;
}

static int revised_core(int n, int m, const double* objective, const double* A, const double* b,
                        int is_min, int64_t max_iter, double* x, double* finalZ, int32_t* basis,
                        double* Binv_out, double* xB_out, int32_t* log_row, int32_t* log_enter,
                        int32_t* log_leave, int64_t log_cap, int64_t* iterations, rev_trace* tr) {
    if (n <= 0 || m <= 0) return ORC_BAD_ARGUMENT; /* :43-44 */
    int status = ORC_OK_OPTIMAL;
    double* c = (double*)malloc(sizeof(double) * n);
    double* Binv = (double*)calloc((size_t)m * m, sizeof(double));
    double* scratch = (double*)malloc(sizeof(double) * (size_t)m * m);
    double* cB = (double*)calloc(m, sizeof(double));
    double* xB = (double*)calloc(m, sizeof(double));
    double* y = (double*)malloc(sizeof(double) * m);
    double* rcX = (double*)malloc(sizeof(double) * n);
    double* rcS = (double*)malloc(sizeof(double) * m);
    double* col = (double*)malloc(sizeof(double) * m);
    double* u = (double*)malloc(sizeof(double) * m);
    int* basic = (int*)malloc(sizeof(int) * m);
    int* nonbasic = (int*)malloc(sizeof(int) * (n + m)); /* the C# List<int>, in its own order */
    int* sorted = (int*)malloc(sizeof(int) * (n + m));
    double* ratios = (double*)malloc(sizeof(double) * m);
    int* basisPre = (int*)malloc(sizeof(int) * m);
    double* xtmp = (double*)calloc(n, sizeof(double));
    int nnb = 0;
    int64_t iteration = 0;

    for (int j = 0; j < n; j++) c[j] = is_min ? -objective[j] : objective[j]; /* :51 */
    for (int i = 0; i < m; i++) { /* :72-78 */
        Binv[(size_t)i * m + i] = 1.0;
        basic[i] = n + i;
        cB[i] = 0.0;
    }
    for (int j = 0; j < n; j++) nonbasic[nnb++] = j; /* :79 */

    for (;;) {
        mat_vec(Binv, m, m, b, xB); /* :89 */
        int infeasible = 0;
        for (int i = 0; i < m; i++) if (xB[i] < -EPS) { infeasible = 1; break; } /* :90 */
        if (infeasible) { status = ORC_INFEASIBLE_BASIS; break; }

        vec_mat(cB, Binv, m, m, y); /* :93 */

        reduced_costs(n, m, c, A, y, col, rcX, rcS); /* :96-102 */

        /* :105-121 entering: nonBasicVariables.OrderBy(v => v) (stable; values are distinct) */
        int enteringIdx = choose_entering(nonbasic, nnb, sorted, n, rcX, rcS);

        if (enteringIdx == -1) { /* :124-146: ExtractSolution, then the "Optimal" snapshot */
            status = ORC_OK_OPTIMAL;
            if (tr && tr->buf) {
                memset(xtmp, 0, sizeof(double) * n);
                for (int i = 0; i < m; i++)
                    if (basic[i] < n) xtmp[basic[i]] = dotnet_max0(xB[i]);
                for (int i = 0; i < m; i++) { u[i] = 0.0; ratios[i] = INFINITY; } /* :133-134 */
                trace_snapshot(tr, n, m, -1, -1, -1, 0.0, dot(cB, xB, m), dot(objective, xtmp, n),
                               y, rcX, rcS, u, ratios, basic, basic, xB, Binv, A);
            }
            break;
        }
        if (max_iter > 0 && iteration >= max_iter) { status = ORC_PIVOT_LIMIT; break; }

        direction(n, m, A, Binv, enteringIdx, col, u); /* :149-151 */
        int leavingRow = ratio_test(m, xB, u, basic, ratios); /* :154-176 */
        if (leavingRow == -1) { status = ORC_UNBOUNDED; break; } /* :178-179 */
        int leavingVar = basic[leavingRow];
        if (leavingVar == enteringIdx) { status = ORC_ENTERING_ALREADY_BASIC; break; } /* :182-183 */

        if (iteration < log_cap) {
            if (log_row) log_row[iteration] = leavingRow;
            if (log_enter) log_enter[iteration] = enteringIdx;
            if (log_leave) log_leave[iteration] = leavingVar;
        }

        for (int i = 0; i < m; i++) basisPre[i] = basic[i]; /* :186 */
        double enteringRC_pre = (enteringIdx < n) ? rcX[enteringIdx] : rcS[enteringIdx - n];

        /* :194-198 bookkeeping */
        basic[leavingRow] = enteringIdx;
        for (int q = 0; q < nnb; q++)
            if (nonbasic[q] == enteringIdx) { /* List.Remove: first occurrence */
                memmove(nonbasic + q, nonbasic + q + 1, sizeof(int) * (nnb - q - 1));
                nnb--;
                break;
            }
        int contains = 0;
        for (int q = 0; q < nnb; q++) if (nonbasic[q] == leavingVar) { contains = 1; break; }
        if (!contains) nonbasic[nnb++] = leavingVar;

        cB[leavingRow] = (enteringIdx < n) ? c[enteringIdx] : 0.0; /* :205,211 */

        int rc2 = orc_update_binverse(Binv, m, leavingRow, u, scratch); /* :215 */
        if (rc2 != 0) { status = rc2; break; }
        if (tr && tr->buf) { /* :217-247: post-pivot quantities, then the snapshot */
            mat_vec(Binv, m, m, b, xB);
            vec_mat(cB, Binv, m, m, y);
            reduced_costs(n, m, c, A, y, col, rcX, rcS);
            memset(xtmp, 0, sizeof(double) * n); /* ComputeOriginalZFromCurrentBasis :253-262 */
            for (int i = 0; i < m; i++)
                if (basic[i] < n) xtmp[basic[i]] = dotnet_max0(xB[i]);
            trace_snapshot(tr, n, m, enteringIdx, leavingRow, leavingVar, enteringRC_pre,
                           dot(cB, xB, m), dot(objective, xtmp, n), y, rcX, rcS, u, ratios,
                           basisPre, basic, xB, Binv, A);
        }
        iteration++; /* :249 */
    }

    if (status == ORC_OK_OPTIMAL) { /* ExtractSolution :277-287 */
        double* xs = (double*)calloc(n, sizeof(double));
        for (int i = 0; i < m; i++) {
            int v = basic[i];
            if (v < n) xs[v] = dotnet_max0(xB[i]);
        }
        if (x) memcpy(x, xs, sizeof(double) * n);
        if (finalZ) *finalZ = dot(objective, xs, n); /* Dot(cOrig, x) */
        free(xs);
    }
    if (basis) for (int i = 0; i < m; i++) basis[i] = basic[i];
    if (Binv_out) memcpy(Binv_out, Binv, sizeof(double) * (size_t)m * m);
    if (xB_out) memcpy(xB_out, xB, sizeof(double) * m);
    if (iterations) *iterations = iteration;

    free(c); free(Binv); free(scratch); free(cB); free(xB); free(y); free(rcX); free(rcS);
    free(col); free(u); free(basic); free(nonbasic); free(sorted);
    free(ratios); free(basisPre); free(xtmp);
    return status;
}
