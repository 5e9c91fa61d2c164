/*
 * oracle_primal.c -- literal CPU restatement of PrimalSimplexSolver
 * (reference: LPR_381_Group_V22/Simplex/PrimalSimplexSolver.cs).
 *
 * TEST INFRASTRUCTURE ONLY -- see lpr_oracle.h.  PARITY UNPINNED by the reference (it has no
 * tests or golden vectors); pinned by tests/ref_py.py + SURVEY.md section 4 hand traces.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math (oracle/Makefile).  Loop order, the
 * row-major layout and every floating-point expression follow the C# one-to-one so that the
 * single-threaded timing of this file is also the "reference algorithm, C restatement, 1 core"
 * CPU baseline of bench.py.
 */
#include "lpr_oracle.h"

#include <float.h>
#include <math.h>
#include <stddef.h>
#include <string.h>

/* ------------------------------------------------------------------ generator */

uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

double orc_u01(uint64_t seed, uint64_t stream, uint64_t i, uint64_t j) {
    uint64_t k = orc_splitmix64(seed ^ (stream * 0xD1B54A32D192ED03ULL));
    k = orc_splitmix64(k + i);
    k = orc_splitmix64(k + j);
    return (double)(k >> 11) * 0x1.0p-53;
}

/* b_i = (n/4) * (1 + 0.1 * U): right-hand sides of similar size, so that many constraints compete
 * in the ratio test and the LP needs O(m) pivots (with b_i spread over [1, n/4] a handful of tight
 * rows decide the optimum and the solve ends after ~15 pivots -- useless as a pivot benchmark). */
static double orc_rhs(uint64_t seed, int i, int n) {
    double u = orc_u01(seed, 1, (uint64_t)i, 0);
    double s = u * 0.1;
    double t = 1.0 + s;
    return ((double)n * 0.25) * t;
}

void orc_gen_dense_lp(int m, int n, uint64_t seed, double* c, double* A, double* b) {
    for (int j = 0; j < n; j++) c[j] = orc_u01(seed, 2, 0, (uint64_t)j);
    for (int i = 0; i < m; i++) {
        for (int j = 0; j < n; j++) A[(size_t)i * n + j] = orc_u01(seed, 0, (uint64_t)i, (uint64_t)j);
        b[i] = orc_rhs(seed, i, n);
    }
}

void orc_gen_dense_tableau(int m, int n, uint64_t seed, double* T, int32_t* basis) {
    const int R = m + 1, C = n + m + 1;
    memset(T, 0, (size_t)R * C * sizeof(double));
    for (int j = 0; j < n; j++) T[j] = -orc_u01(seed, 2, 0, (uint64_t)j);
    for (int i = 0; i < m; i++) {
        double* row = T + (size_t)(i + 1) * C;
        for (int j = 0; j < n; j++) row[j] = orc_u01(seed, 0, (uint64_t)i, (uint64_t)j);
        row[n + i] = 1.0;
        row[C - 1] = orc_rhs(seed, i, n);
        if (basis) basis[i] = n + i;
    }
}

/* ------------------------------------------------------------------ ctor :27-87 */

int orc_primal_build(int n, int m, const double* objective, const double* A, int lda,
                     const int32_t* ncoef, const int8_t* relation, const double* rhs, int is_max,
                     double* T, int32_t* basis) {
    if (n < 0 || m < 0 || lda < 0) return ORC_BAD_ARGUMENT;
    const int C = n + m + 1; /* :56 */
    const int R = m + 1;     /* :57 */
    memset(T, 0, (size_t)R * C * sizeof(double)); /* :58 new double[,] is zero-filled */

    /* :61-62 */
    for (int i = 0; i < n; i++) T[i] = is_max ? -objective[i] : objective[i];

    /* :65-83, with the >= negation of :36-41 applied on the fly */
    for (int i = 0; i < m; i++) {
        const int ge = (relation != NULL && relation[i] == 1);
        const int cnt = ncoef ? ncoef[i] : n;
        double* row = T + (size_t)(i + 1) * C;
        for (int j = 0; j < n; j++) {
            if (j < cnt) { /* :70 */
                double a = A[(size_t)i * lda + j];
                row[j] = ge ? -a : a;
            }
        }
        row[n + i] = 1.0;                 /* :75-76 */
        basis[i] = n + i;                 /* :78 */
        row[C - 1] = ge ? -rhs[i] : rhs[i]; /* :82 */
    }
    return ORC_OK_OPTIMAL;
}

/* ------------------------------------------------------------------ FindEnteringVariable :152-167 */

int orc_find_entering(const double* T, int R, int C) {
    (void)R;
    int enteringCol = -1;
    double mostNegative = 0;
    int totalCols = C - 1;
    for (int j = 0; j < totalCols; j++) {
        if (T[j] < mostNegative) {
            mostNegative = T[j];
            enteringCol = j;
        }
    }
    return enteringCol;
}

/* ------------------------------------------------------------------ FindLeavingVariable :169-191 */

int orc_find_leaving(const double* T, int R, int C, int e) {
    int leavingRow = -1;
    double minRatio = DBL_MAX; /* double.MaxValue */
    int rhsCol = C - 1;
    for (int i = 1; i < R; i++) {
        double a = T[(size_t)i * C + e];
        if (a > 1e-9) {
            double ratio = T[(size_t)i * C + rhsCol] / a;
            if (ratio >= 0 && ratio < minRatio) {
                minRatio = ratio;
                leavingRow = i;
            }
        }
    }
    return leavingRow;
}

/* ------------------------------------------------------------------ Pivot :193-211 */

void orc_pivot(double* T, int R, int C, int r, int e) {
    double* prow = T + (size_t)r * C;
    double pivotElement = prow[e];

    for (int j = 0; j < C; j++) prow[j] /= pivotElement; /* :198-199 true division */

    for (int i = 0; i < R; i++) {
        if (i != r) {
            double* row = T + (size_t)i * C;
            double factor = row[e]; /* :206 read once, before the j loop */
            for (int j = 0; j < C; j++) {
                double prod = factor * prow[j]; /* product rounded ... */
                row[j] = row[j] - prod;         /* ... then the difference (:208), no FMA */
            }
        }
    }
}

/* ------------------------------------------------------------------ Solve :102-150 */

int orc_primal_solve(double* T, int R, int C, int32_t* basis, int64_t max_pivots,
                     int32_t* log_rows, int32_t* log_cols, int64_t log_cap, int64_t* pivots) {
    int64_t iteration = 0;
    int status;
    for (;;) {
        int enteringCol = orc_find_entering(T, R, C);
        if (enteringCol == -1) { status = ORC_OK_OPTIMAL; break; }
        int leavingRow = orc_find_leaving(T, R, C, enteringCol);
        if (leavingRow == -1) { status = ORC_UNBOUNDED; break; }
        if (max_pivots > 0 && iteration >= max_pivots) { status = ORC_PIVOT_LIMIT; break; }
        if (log_rows && iteration < log_cap) log_rows[iteration] = leavingRow;
        if (log_cols && iteration < log_cap) log_cols[iteration] = enteringCol;
        ++iteration;                                  /* :138 */
        orc_pivot(T, R, C, leavingRow, enteringCol);  /* :141 */
        if (basis) basis[leavingRow - 1] = enteringCol; /* :142 */
    }
    if (pivots) *pivots = iteration;
    return status;
}

/* ------------------------------------------------------------------ ExtractSolution :213-252 */

void orc_extract_solution(const double* T, int R, int C, int n, double* x, double* z) {
    int rhsCol = C - 1;
    for (int j = 0; j < n; j++) {
        x[j] = 0.0;
        int basicRow = -1;
        int isBasic = 1;
        for (int i = 1; i < R; i++) {
            double v = T[(size_t)i * C + j];
            if (fabs(v - 1.0) < 1e-9) {
                if (basicRow == -1) basicRow = i;
                else { isBasic = 0; break; }
            } else if (fabs(v) > 1e-9) {
                isBasic = 0;
                break;
            }
        }
        if (isBasic && basicRow != -1) x[j] = T[(size_t)basicRow * C + rhsCol];
    }
    if (z) *z = T[rhsCol]; /* FinalZ :113 */
}
