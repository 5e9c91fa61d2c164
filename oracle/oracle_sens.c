/*
 * oracle_sens.c -- literal CPU restatement of the re-solve half of SensitivityAnalyzer ("next" row
 * f4; reference: LPR_381_Group_V22/SensitivityAnalysis/SensitivityAnalyzer.cs).
 *
 * TEST INFRASTRUCTURE ONLY -- see lpr_oracle.h.  PARITY UNPINNED by the reference (no tests).
 *
 * Restated: the constructor (:22-39), RebuildBasicsFromTableau (:706-723), GetBasicRow /
 * IsPivotColumn (:69-84), IsOptimal (:86-96), Pivot (:98-119, rows with |factor| < EPS skipped),
 * ReOptimize (:121-166), DualSimplexIfNeeded (:168-201), ResolveAll (:203-208) and the six edits
 * that call them: ChangeNonBasicReducedCost (:300-321), ChangeBasic (:362-393), ChangeRHS
 * (:427-470, with its snapshot/rollback), ChangeNonBasicColumn (:502-531), AddNewActivity
 * (:534-584), AddNewConstraintNonInteractive (:609-659).  The console prompts are replaced by
 * arguments; the range / shadow-price displays are read-only arithmetic on a few vectors and live
 * in the host mirror.
 *
 * Return codes: 0 ok; 1 "Unbounded during re-optimization" (:151); 2 "Infeasible after RHS change
 * (dual simplex)" (:197); 3 "Zero pivot encountered" (:101); 5 iteration limit (:126, :183);
 * -1 invalid index (the C# prints "Invalid ..." and returns); 8 ChangeRHS rolled back (:462-469).
 */
#include "lpr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPS 1e-9

typedef struct orc_sens {
    int R, C;
    double* T;
    int nb;       /* basicVars.Count */
    int* basic;
    int nsol;     /* solutionVector.Count */
    double* sol;
    double z;
    int32_t* log; /* (kind 0 dual / 1 primal, leaveRow, enterCol) of every pivot performed */
    int64_t nlog, logcap;
} orc_sens;

#define TT(s, i, j) ((s)->T[(size_t)(i) * (s)->C + (j)])

static int is_pivot_column(const orc_sens* s, int prow, int col) { /* :79-84 */
    for (int i = 1; i < s->R; i++)
        if (i != prow && fabs(TT(s, i, col)) > EPS) return 0;
    return 1;
}

static int get_basic_row(const orc_sens* s, int col) { /* :69-77 */
    for (int i = 1; i < s->R; i++)
        if (fabs(TT(s, i, col) - 1.0) < EPS && is_pivot_column(s, i, col)) return i;
    return -1;
}

static void rebuild_basics(orc_sens* s) { /* :706-723 */
    int m = s->R - 1;
    s->basic = (int*)realloc(s->basic, sizeof(int) * (m > 0 ? m : 1));
    s->nb = m;
    for (int i = 0; i < m; i++) s->basic[i] = -1;
    for (int i = 1; i <= m; i++)
        for (int j = 0; j < s->C - 1; j++)
            if (fabs(TT(s, i, j) - 1.0) < EPS && is_pivot_column(s, i, j)) {
                s->basic[i - 1] = j;
                break;
            }
}

static int basic_contains(const orc_sens* s, int j) {
    for (int k = 0; k < s->nb; k++)
        if (s->basic[k] == j) return 1;
    return 0;
}

static int is_optimal(const orc_sens* s) { /* :86-96 */
    for (int j = 0; j < s->C - 1; j++) {
        if (basic_contains(s, j)) continue;
        if (TT(s, 0, j) < -EPS) return 0;
    }
    return 1;
}

static int pivot(orc_sens* s, int enterCol, int leaveRow, int kind) { /* :98-119 */
    double piv = TT(s, leaveRow, enterCol);
    if (fabs(piv) < EPS) return 3;
    if (s->nlog == s->logcap) {
        s->logcap = s->logcap ? 2 * s->logcap : 64;
        s->log = (int32_t*)realloc(s->log, sizeof(int32_t) * 3 * (size_t)s->logcap);
    }
    s->log[3 * s->nlog] = kind;
    s->log[3 * s->nlog + 1] = leaveRow;
    s->log[3 * s->nlog + 2] = enterCol;
    s->nlog++;
    for (int j = 0; j < s->C; j++) TT(s, leaveRow, j) /= piv;
    for (int i = 0; i < s->R; i++) {
        if (i == leaveRow) continue;
        double factor = TT(s, i, enterCol);
        if (fabs(factor) < EPS) continue;
        for (int j = 0; j < s->C; j++) {
            double prod = factor * TT(s, leaveRow, j);
            TT(s, i, j) = TT(s, i, j) - prod;
        }
    }
    int idx = leaveRow - 1;
    if (idx >= 0 && idx < s->nb) s->basic[idx] = enterCol;
    return 0;
}

static int reoptimize(orc_sens* s, int maxIter) { /* :121-166 */
    int iter = 0;
    while (!is_optimal(s)) {
        if (iter++ > maxIter) return 5;
        int enter = -1;
        double mostNeg = 0.0;
        for (int j = 0; j < s->C - 1; j++) {
            if (basic_contains(s, j)) continue;
            double rc = TT(s, 0, j);
            if (rc < mostNeg) { mostNeg = rc; enter = j; }
        }
        if (enter == -1) break;
        int leave = -1;
        double bestRatio = INFINITY;
        for (int i = 1; i < s->R; i++) {
            double aij = TT(s, i, enter);
            if (aij > EPS) {
                double ratio = TT(s, i, s->C - 1) / aij;
                if (ratio < bestRatio - EPS) { bestRatio = ratio; leave = i; }
            }
        }
        if (leave == -1) return 1;
        int rc2 = pivot(s, enter, leave, 1);
        if (rc2) return rc2;
    }
    s->z = TT(s, 0, s->C - 1);
    s->sol = (double*)realloc(s->sol, sizeof(double) * (s->C - 1 > 0 ? s->C - 1 : 1));
    s->nsol = s->C - 1;
    for (int j = 0; j < s->C - 1; j++) {
        int r = get_basic_row(s, j);
        s->sol[j] = (r == -1) ? 0.0 : TT(s, r, s->C - 1);
    }
    return 0;
}

static int dual_if_needed(orc_sens* s, int maxIter) { /* :168-201 */
    int iter = 0;
    for (;;) {
        int leave = -1;
        double mostNeg = 0.0;
        for (int i = 1; i < s->R; i++) {
            double bi = TT(s, i, s->C - 1);
            if (bi < mostNeg - EPS) { mostNeg = bi; leave = i; }
        }
        if (leave == -1) break;
        if (iter++ > maxIter) return 5;
        int enter = -1;
        double bestRatio = INFINITY;
        for (int j = 0; j < s->C - 1; j++) {
            double aij = TT(s, leave, j);
            if (aij < -EPS) {
                double ratio = TT(s, 0, j) / (-aij);
                if (ratio < bestRatio - EPS) { bestRatio = ratio; enter = j; }
            }
        }
        if (enter == -1) return 2;
        int rc = pivot(s, enter, leave, 0);
        if (rc) return rc;
    }
    return 0;
}

static int resolve_all(orc_sens* s) { /* :203-208 */
    rebuild_basics(s);
    int rc = dual_if_needed(s, 10000);
    if (rc) return rc;
    return reoptimize(s, 10000);
}

static double shadow_price(const orc_sens* s, int k /*1..m*/) { /* :212-222, :62-67 */
    int m = s->R - 1;
    int n = s->C - m - 1;
    return TT(s, 0, n + (k - 1));
}

/* ------------------------------------------------------------------ API */

orc_sens* orc_sens_create(const double* finalTableau, int R, int C, const double* solution,
                          int nsol, double z, const int32_t* basic, int nbasic) { /* :22-39 */
    orc_sens* s = (orc_sens*)calloc(1, sizeof(orc_sens));
    s->R = R;
    s->C = C;
    s->T = (double*)malloc(sizeof(double) * (size_t)R * C);
    memcpy(s->T, finalTableau, sizeof(double) * (size_t)R * C);
    s->nsol = nsol;
    s->sol = (double*)malloc(sizeof(double) * (nsol > 0 ? nsol : 1));
    if (nsol > 0) memcpy(s->sol, solution, sizeof(double) * nsol);
    s->z = z;
    s->nb = nbasic;
    s->basic = (int*)malloc(sizeof(int) * (nbasic > 0 ? nbasic : 1));
    for (int i = 0; i < nbasic; i++) s->basic[i] = basic[i];
    TT(s, 0, C - 1) = z; /* :32 */
    rebuild_basics(s);   /* :35 */
    return s;
}

void orc_sens_destroy(orc_sens* s) {
    if (!s) return;
    free(s->T);
    free(s->basic);
    free(s->sol);
    free(s->log);
    free(s);
}

void orc_sens_shape(const orc_sens* s, int* R, int* C, int* nsol, int* nbasic, double* z) {
    if (R) *R = s->R;
    if (C) *C = s->C;
    if (nsol) *nsol = s->nsol;
    if (nbasic) *nbasic = s->nb;
    if (z) *z = s->z;
}

void orc_sens_read(const orc_sens* s, double* T, int32_t* basic, double* sol) {
    if (T) memcpy(T, s->T, sizeof(double) * (size_t)s->R * s->C);
    if (basic) for (int i = 0; i < s->nb; i++) basic[i] = s->basic[i];
    if (sol) memcpy(sol, s->sol, sizeof(double) * s->nsol);
}

int64_t orc_sens_log_read(const orc_sens* s, int32_t* triples, int64_t cap) {
    int64_t k = s->nlog < cap ? s->nlog : cap;
    if (triples && k > 0) memcpy(triples, s->log, sizeof(int32_t) * 3 * (size_t)k);
    return s->nlog;
}

int orc_sens_resolve_all(orc_sens* s) { return resolve_all(s); }

int orc_sens_change_nonbasic_cbar(orc_sens* s, int index, double newCbar) { /* :300-321 */
    if (index < 0 || index >= s->C - 1 || basic_contains(s, index)) return -1;
    TT(s, 0, index) = newCbar;
    return resolve_all(s);
}

int orc_sens_change_basic(orc_sens* s, int col, double delta) { /* :362-393 */
    if (col < 0 || col >= s->C - 1 || !basic_contains(s, col)) return -1;
    int r = get_basic_row(s, col);
    if (r < 0) return -1; /* "Could not locate basic row." :379 */
    for (int j = 0; j < s->C - 1; j++) {
        double prod = delta * TT(s, r, j);
        TT(s, 0, j) = TT(s, 0, j) + prod;
    }
    {
        double prod = delta * TT(s, r, s->C - 1);
        TT(s, 0, s->C - 1) = TT(s, 0, s->C - 1) + prod;
    }
    s->z = TT(s, 0, s->C - 1);
    return resolve_all(s);
}

int orc_sens_change_rhs(orc_sens* s, int k, double newB) { /* :427-470 */
    if (k < 1 || k >= s->R) return -1;
    double* snap = (double*)malloc(sizeof(double) * (size_t)s->R * s->C);
    memcpy(snap, s->T, sizeof(double) * (size_t)s->R * s->C);
    int* bsnap = (int*)malloc(sizeof(int) * (s->nb > 0 ? s->nb : 1));
    memcpy(bsnap, s->basic, sizeof(int) * s->nb);
    int nbsnap = s->nb;
    double oldZ = s->z;
    double oldB = TT(s, k, s->C - 1);
    double delta = newB - oldB;
    int m = s->R - 1, n = s->C - m - 1;
    int sCol = n + (k - 1);
    for (int i = 1; i < s->R; i++) {
        double prod = delta * TT(s, i, sCol);
        TT(s, i, s->C - 1) = TT(s, i, s->C - 1) + prod;
    }
    {
        double prod = shadow_price(s, k) * delta;
        TT(s, 0, s->C - 1) = TT(s, 0, s->C - 1) + prod;
    }
    s->z = TT(s, 0, s->C - 1);
    int rc = dual_if_needed(s, 10000);
    if (!rc) rc = reoptimize(s, 10000);
    if (rc) { /* catch: restore :462-469 (solutionVector is NOT restored by the C#) */
        memcpy(s->T, snap, sizeof(double) * (size_t)s->R * s->C);
        s->z = oldZ;
        s->basic = (int*)realloc(s->basic, sizeof(int) * (nbsnap > 0 ? nbsnap : 1));
        memcpy(s->basic, bsnap, sizeof(int) * nbsnap);
        s->nb = nbsnap;
        rc = 8;
    }
    free(snap);
    free(bsnap);
    return rc;
}

int orc_sens_change_nonbasic_column(orc_sens* s, int row, int col, double newVal) { /* :502-531 */
    if (row < 1 || row >= s->R) return -1;
    if (col < 0 || col >= s->C - 1 || basic_contains(s, col)) return -1;
    double oldVal = TT(s, row, col);
    double delta = newVal - oldVal;
    TT(s, row, col) = newVal;
    double yi = shadow_price(s, row);
    {
        double prod = yi * delta;
        TT(s, 0, col) = TT(s, 0, col) + prod;
    }
    return resolve_all(s);
}

int orc_sens_add_activity(orc_sens* s, double cNew, const double* aNew) { /* :534-584 */
    int m = s->R - 1, n = s->C - m - 1;
    double yTa = 0.0;
    for (int i = 0; i < m; i++) {
        double prod = shadow_price(s, i + 1) * aNew[i];
        yTa = yTa + prod;
    }
    double cbarNew = yTa - cNew;
    int C2 = s->C + 1;
    double* nT = (double*)calloc((size_t)s->R * C2, sizeof(double));
    for (int i = 0; i < s->R; i++) {
        for (int j = 0; j < n; j++) nT[(size_t)i * C2 + j] = TT(s, i, j);
        nT[(size_t)i * C2 + n] = (i == 0) ? cbarNew : aNew[i - 1];
        for (int j = n; j < s->C - 1; j++) nT[(size_t)i * C2 + j + 1] = TT(s, i, j);
        nT[(size_t)i * C2 + s->C] = TT(s, i, s->C - 1);
    }
    free(s->T);
    s->T = nT;
    s->C = C2;
    for (int i = 0; i < s->nb; i++)
        if (s->basic[i] >= n) s->basic[i]++;
    return resolve_all(s);
}

int orc_sens_add_constraint(orc_sens* s, const double* tech, int ntech, double rhs) { /* :609-659 */
    int oldM = s->R - 1, oldNPlusM = s->C - 1;
    if (ntech != oldNPlusM) return -1;
    int R2 = s->R + 1, C2 = s->C + 1;
    double* nT = (double*)calloc((size_t)R2 * C2, sizeof(double));
    for (int i = 0; i < s->R; i++) {
        for (int j = 0; j < s->C - 1; j++) nT[(size_t)i * C2 + j] = TT(s, i, j);
        nT[(size_t)i * C2 + s->C] = TT(s, i, s->C - 1);
    }
    int newSlackCol = s->C - 1;
    for (int i = 0; i < R2; i++) nT[(size_t)i * C2 + newSlackCol] = (i == s->R) ? 1.0 : 0.0;
    for (int j = 0; j < oldNPlusM; j++) {
        double coeff = -tech[j];
        for (int pos = 0; pos < oldM; pos++) {
            int basicCol = s->basic[pos];
            if (basicCol < 0 || basicCol >= ntech) { /* tech[-1] throws IndexOutOfRange in C# */
                free(nT);
                return 9;
            }
            double prod = tech[basicCol] * TT(s, pos + 1, j);
            coeff = coeff + prod;
        }
        nT[(size_t)s->R * C2 + j] = coeff;
    }
    double aX = 0.0;
    int lim = ntech < s->nsol ? ntech : s->nsol;
    for (int j = 0; j < lim; j++) {
        double prod = tech[j] * s->sol[j];
        aX = aX + prod;
    }
    nT[(size_t)s->R * C2 + s->C] = rhs - aX;
    nT[newSlackCol] = 0.0;
    free(s->T);
    s->T = nT;
    s->R = R2;
    s->C = C2;
    s->basic = (int*)realloc(s->basic, sizeof(int) * (s->nb + 1));
    s->basic[s->nb++] = newSlackCol;
    return resolve_all(s);
}
