/*
 * oracle_bb.c -- literal CPU restatement of the Branch & Bound path
 * (reference: LPR_381_Group_V22/IntegerProgramming/BranchBoundSimplexSolver.cs and
 *  IntegerProgramming/BranchAndBoundAdapter.cs).
 *
 * TEST INFRASTRUCTURE ONLY -- see lpr_oracle.h.  PARITY UNPINNED by the reference (no tests, no
 * golden vectors); pinned by tests/ref_py.py (independent restatement) and by the integer optimum
 * of the sample knapsack known by inspection.
 *
 * The C# works on List<List<double>>; here a tableau is a dense row-major matrix (all rows of a
 * reference tableau have the same length).  What is restated, including the quirks:
 *   - .NET Framework Math.Round(x, 4) / Math.Round(x)              (orc_round4 / orc_round_int)
 *   - LINQ Min()+IndexOf "first occurrence" selections, Double.Equals semantics (+0 == -0)
 *   - PerformDualPivot :115-201 (row 0 takes part in the RHS scan), PerformPrimalPivot :203-279
 *   - DoDualSimplex :289-468 in tableauOverride mode, incl. the exception paths that the callers'
 *     try/catch turns into "branch failed"
 *   - BranchAndBound: IdentifyBasicVariables :642-692, AddConstraint :694-803,
 *     CheckIntegerBasicVar :805-857, CreateBranches :859-890, ExtractSolution :899-921,
 *     UpdateOptimalSolution :935-983, ShouldPrunebranch :985-1004, ExecuteBranchAndBound :1006-1233
 *     (DFS stack, lower child first, 20-node cap :1038).
 */
#include "lpr_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ .NET Framework rounding */

/* Math.Round(double): clr/src/classlibnative/float/floatnative.cpp COMDouble::Round */
double orc_round_int(double x) {
    if (isnan(x) || isinf(x)) return x;
    if (fabs(x) < 9.2e18 && x == (double)((long long)x)) return x;
    double tempVal = x + 0.5;
    double flrTempVal = floor(tempVal);
    if (flrTempVal == tempVal && fmod(tempVal, 2.0) != 0) flrTempVal -= 1.0;
    return copysign(flrTempVal, x);
}

/* Math.Round(double, 4): Math.InternalRound, MidpointRounding.ToEven, doubleRoundLimit = 1e16 */
double orc_round4(double x) {
    if (fabs(x) < 1e16) {
        double power10 = 10000.0;
        x = x * power10;
        x = orc_round_int(x);
        x = x / power10;
    }
    return x;
}

/* ------------------------------------------------------------------ tableau helpers */

typedef struct {
    int rows, cols;
    double* a;
} Tab;

static Tab tab_new(int rows, int cols) {
    Tab t;
    t.rows = rows;
    t.cols = cols;
    t.a = (double*)calloc((size_t)rows * cols, sizeof(double));
    return t;
}
static Tab tab_copy(const Tab* s) {
    Tab t = tab_new(s->rows, s->cols);
    memcpy(t.a, s->a, sizeof(double) * (size_t)s->rows * s->cols);
    return t;
}
static void tab_free(Tab* t) {
    free(t->a);
    t->a = NULL;
}
#define AT(t, i, j) ((t)->a[(size_t)(i) * (t)->cols + (j)])

static void tab_round(Tab* t) { /* RoundTableau :552-567 */
    size_t n = (size_t)t->rows * t->cols;
    for (size_t k = 0; k < n; k++) t->a[k] = orc_round4(t->a[k]);
}
static void tab_clean_negzero(Tab* t) { /* `if (row[i] == -0.0) row[i] = 0.0;` :307-313 */
    size_t n = (size_t)t->rows * t->cols;
    for (size_t k = 0; k < n; k++)
        if (t->a[k] == 0.0) t->a[k] = 0.0;
}

/* ------------------------------------------------------------------ pivots */

/* PerformDualPivot :115-201.  Returns 1 and *out on success, 0 for the `(tableau, null)` exits. */
static int dual_pivot(const Tab* T, Tab* out, int* prow_out, int* pcol_out) {
    const int R = T->rows, C = T->cols;
    /* :118-123 */
    int any = 0;
    double minRhs = 0;
    for (int i = 0; i < R; i++) {
        double x = AT(T, i, C - 1);
        if (x < 0) {
            if (!any || x < minRhs) minRhs = x; /* LINQ Min over the negatives */
            any = 1;
        }
    }
    if (!any) return 0;
    int pr = -1;
    for (int i = 0; i < R; i++)
        if (AT(T, i, C - 1) == minRhs) { pr = i; break; } /* rhsValues.IndexOf(minRhs) */
    /* :126-143 */
    double* th = (double*)malloc(sizeof(double) * (C > 1 ? C - 1 : 1));
    for (int i = 0; i < C - 1; i++) {
        if (AT(T, pr, i) < 0) th[i] = fabs(AT(T, 0, i) / AT(T, pr, i));
        else th[i] = INFINITY;
    }
    /* :146-148 */
    int all0inf = 1;
    for (int i = 0; i < C - 1; i++)
        if (!(th[i] == 0 || th[i] == INFINITY)) { all0inf = 0; break; }
    double minPos;
    if (all0inf) minPos = 0;
    else {
        int have = 0;
        minPos = INFINITY; /* DefaultIfEmpty(+inf) */
        for (int i = 0; i < C - 1; i++)
            if (th[i] > 0) {
                if (!have || th[i] < minPos) minPos = th[i];
                have = 1;
            }
    }
    /* :154 IndexOf (Double.Equals: NaN equals NaN, +0 equals -0) */
    int pc = -1;
    for (int i = 0; i < C - 1; i++)
        if (th[i] == minPos || (isnan(th[i]) && isnan(minPos))) { pc = i; break; }
    free(th);
    if (pc < 0) return 0; /* tableau[rowIndex][-1] throws -> caught -> (tableau, null) :165-172 */
    double p = AT(T, pr, pc);
    *out = tab_new(R, C);
    for (int j = 0; j < C; j++) { /* :174-178 */
        double v = AT(T, pr, j) / p;
        if (v == 0.0) v = 0.0;
        AT(out, pr, j) = v;
    }
    for (int i = 0; i < R; i++) { /* :182-190 */
        if (i == pr) continue;
        double f = AT(T, i, pc);
        for (int j = 0; j < C; j++) {
            double prod = f * AT(out, pr, j);
            AT(out, i, j) = AT(T, i, j) - prod;
        }
    }
    *prow_out = pr;
    *pcol_out = pc;
    return 1;
}

/* PerformPrimalPivot :203-279 (isMinimization == false).  0 = the `(null, null)` exits. */
static int primal_pivot(const Tab* T, Tab* out, int* prow_out, int* pcol_out) {
    const int R = T->rows, C = T->cols;
    /* :206-218 */
    int any = 0;
    double pv = 0;
    for (int j = 0; j < C - 1; j++) {
        double x = AT(T, 0, j);
        if (x < 0 && x != 0) {
            if (!any || x < pv) pv = x;
            any = 1;
        }
    }
    if (!any) return 0; /* Min() on an empty sequence throws -> (null, null) */
    int pc = -1;
    for (int j = 0; j < C; j++) /* tableau[0].IndexOf(pivotValue): the whole row */
        if (AT(T, 0, j) == pv) { pc = j; break; }
    /* :221-225 */
    int nt = R - 1;
    double* th = (double*)malloc(sizeof(double) * (nt > 0 ? nt : 1));
    for (int i = 1; i < R; i++)
        th[i - 1] = (AT(T, i, pc) != 0) ? AT(T, i, C - 1) / AT(T, i, pc) : INFINITY;
    int allNeg = 1; /* All() of an empty list is true */
    for (int i = 0; i < nt; i++)
        if (!(th[i] < 0)) { allNeg = 0; break; }
    if (allNeg) { free(th); return 0; }
    /* :233-244 */
    int anyPos = 0, has0 = 0;
    for (int i = 0; i < nt; i++) {
        if (th[i] > 0 && th[i] != INFINITY) anyPos = 1;
        if (th[i] == 0) has0 = 1;
    }
    double minTheta;
    if (!anyPos) {
        if (has0) minTheta = 0.0;
        else { free(th); return 0; }
    } else {
        int have = 0;
        minTheta = 0;
        for (int i = 0; i < nt; i++)
            if (th[i] > 0 && th[i] != INFINITY) {
                if (!have || th[i] < minTheta) minTheta = th[i];
                have = 1;
            }
    }
    if (minTheta == INFINITY && !has0) { free(th); return 0; } /* :246 */
    int idx = -1;
    for (int i = 0; i < nt; i++)
        if (th[i] == minTheta) { idx = i; break; } /* thetas.IndexOf(minTheta) */
    free(th);
    int pr = idx + 1; /* :249 */
    double p = AT(T, pr, pc);
    if (p == 0) return 0; /* :252 */
    *out = tab_new(R, C);
    for (int j = 0; j < C; j++) { /* :257-261 */
        double v = AT(T, pr, j) / p;
        if (v == 0.0) v = 0.0;
        AT(out, pr, j) = v;
    }
    for (int i = 0; i < R; i++) { /* :263-271 */
        if (i == pr) continue;
        double f = AT(T, i, pc);
        for (int j = 0; j < C; j++) {
            double prod = f * AT(out, pr, j);
            AT(out, i, j) = AT(T, i, j) - prod;
        }
    }
    *prow_out = pr;
    *pcol_out = pc;
    return 1;
}

/* Pivot trace shared with the tests: (node id, phase 0 dual / 1 primal, row, col). */
typedef struct {
    int32_t* buf;
    int64_t cap, n;
} PivTrace;
static void trace_push(PivTrace* tr, int node, int phase, int row, int col) {
    if (tr && tr->buf && tr->n < tr->cap) {
        tr->buf[4 * tr->n] = node;
        tr->buf[4 * tr->n + 1] = phase;
        tr->buf[4 * tr->n + 2] = row;
        tr->buf[4 * tr->n + 3] = col;
    }
    if (tr) tr->n++;
}

/*
 * DoDualSimplex :289-468 with tableauOverride != null.  `start` is consumed (the C# mutates the
 * override in place and keeps it as tableaux[0]).
 * Result: 0 = solved (last tableau in *last, pivots counted),  1 = infeasible (optimalValue null,
 * :324-331),  2 = an exception escaped (RemoveAt / Last() on an empty list, :396-399,:466).
 */
static int do_dual_simplex(Tab start, Tab* last, int* npiv, int node, PivTrace* tr) {
    Tab cur = start;       /* tableaux.Last() */
    Tab prev;              /* the one before it (needed for the RemoveAt of :396) */
    int have_prev = 0;
    int count = 1;         /* tableaux.Count */
    int pivots = 0;        /* pivotColumns.Count */
    prev.a = NULL;
    prev.rows = prev.cols = 0;

    for (;;) { /* dual phase :305-343 */
        tab_clean_negzero(&cur);
        int ok = 1;
        for (int i = 0; i < cur.rows; i++)
            if (!(AT(&cur, i, cur.cols - 1) >= -1e-9)) { ok = 0; break; }
        if (ok) break;
        Tab nt;
        int pr, pc;
        if (!dual_pivot(&cur, &nt, &pr, &pc)) {
            if (have_prev) tab_free(&prev);
            tab_free(&cur);
            return 1;
        }
        trace_push(tr, node, 0, pr, pc);
        tab_clean_negzero(&nt);
        if (have_prev) tab_free(&prev);
        prev = cur;
        have_prev = 1;
        cur = nt;
        count++;
        pivots++;
    }

    int isOptimal = 1; /* :345-348 */
    for (int j = 0; j < cur.cols - 1; j++)
        if (!(AT(&cur, 0, j) >= 0)) { isOptimal = 0; break; }

    if (!isOptimal) {
        for (;;) { /* primal phase :352-390 */
            tab_clean_negzero(&cur);
            isOptimal = 1;
            for (int j = 0; j < cur.cols - 1; j++)
                if (!(AT(&cur, 0, j) >= 0)) { isOptimal = 0; break; }
            if (isOptimal) break;
            Tab nt;
            int pr, pc;
            if (!primal_pivot(&cur, &nt, &pr, &pc)) break; /* thetaCol.ToList() on null -> catch -> break */
            trace_push(tr, node, 1, pr, pc);
            if (have_prev) tab_free(&prev);
            prev = cur;
            have_prev = 1;
            cur = nt;
            count++;
            pivots++;
        }
        int allNonNeg = 1; /* :392-393, strict 0, row 0 included */
        for (int i = 0; i < cur.rows; i++)
            if (!(AT(&cur, i, cur.cols - 1) >= 0)) { allNonNeg = 0; break; }
        if (!allNonNeg) { /* :395-400 */
            if (pivots == 0) { /* pivotColumns.RemoveAt(-1) throws (tableaux already shortened) */
                if (have_prev) tab_free(&prev);
                tab_free(&cur);
                return 2;
            }
            tab_free(&cur);
            cur = prev;
            have_prev = 0;
            count--;
            pivots--;
            trace_push(tr, node, 2, -1, -1); /* marks "last tableau dropped" */
        }
    }
    if (have_prev) tab_free(&prev);
    if (count == 0) { tab_free(&cur); return 2; }
    *last = cur;
    *npiv = pivots;
    return 0;
}

/* ------------------------------------------------------------------ BranchAndBound helpers */

#define BB_EPS 1e-6 /* :493 */

/* `(int)d` (:870-871) as .NET Framework 4.7.2's x64 JIT compiles it (cvttsd2si): out of int's range,
 * or NaN, gives 0x80000000.  Spelt out because the C cast is undefined there. */
static int orc_to_int32(double x) {
    if (!(x > -2147483649.0 && x < 2147483648.0)) return (-2147483647 - 1);
    return (int)x;
}

static int is_integer(double v) { /* :595-599 */
    double r = orc_round4(v);
    return fabs(r - orc_round_int(r)) <= BB_EPS;
}

/* IdentifyBasicVariables :642-692 on one tableau; returns count, fills `out` (cap = cols). */
static int identify_basic(const Tab* T, int* out) {
    const int R = T->rows, C = T->cols;
    int nb = 0;
    int* cand = (int*)malloc(sizeof(int) * C);
    int* key = (int*)malloc(sizeof(int) * C);
    for (int k = 0; k < C; k++) {
        double sum = 0; /* Enumerable.Sum: sequential */
        for (int i = 0; i < R; i++) sum += orc_round4(AT(T, i, k));
        sum = orc_round4(sum);
        if (fabs(sum - 1.0) <= BB_EPS) cand[nb++] = k;
    }
    for (int q = 0; q < nb; q++) { /* key = col.Contains(1.0) ? col.IndexOf(1.0) : col.Count */
        int k = cand[q];
        key[q] = R;
        for (int i = 0; i < R; i++)
            if (orc_round4(AT(T, i, k)) == 1.0) { key[q] = i; break; }
    }
    /* OrderBy is a stable sort */
    for (int a = 1; a < nb; a++) {
        int kv = key[a], cv = cand[a], b = a - 1;
        while (b >= 0 && key[b] > kv) { key[b + 1] = key[b]; cand[b + 1] = cand[b]; b--; }
        key[b + 1] = kv;
        cand[b + 1] = cv;
    }
    for (int q = 0; q < nb; q++) out[q] = cand[q];
    free(cand);
    free(key);
    return nb;
}

/* AddConstraint :694-803 for ONE new constraint con[0..n+1] = (coefficients..., bound, type).
 * Returns the "outputTab" (adjusted tableau). */
static Tab add_constraint(const Tab* base, const double* con, int conLen) {
    Tab working = tab_copy(base);
    tab_round(&working); /* :701-702 */
    int* basic = (int*)malloc(sizeof(int) * working.cols);
    int nb = identify_basic(&working, basic); /* :703-704 */
    const int R = working.rows, C = working.cols;
    Tab upd = tab_new(R + 1, C + 1);
    for (int i = 0; i < R; i++) { /* Insert(Count - 1, 0.0) :716-719 */
        for (int j = 0; j < C - 1; j++) AT(&upd, i, j) = AT(&working, i, j);
        AT(&upd, i, C - 1) = 0.0;
        AT(&upd, i, C) = AT(&working, i, C - 1);
    }
    /* new row :721-744 (length C + 1) */
    for (int i = 0; i < conLen - 2; i++) AT(&upd, R, i) = orc_round4(con[i]);
    AT(&upd, R, C) = orc_round4(con[conLen - 2]);
    int slackPos = ((C + 1) - 1) - 1; /* ((newConstraint.Count - newConstraints.Count) - 1) + k */
    AT(&upd, R, slackPos) = (con[conLen - 1] == 1) ? -1.0 : 1.0;
    tab_round(&upd); /* :747 */
    Tab outp = tab_copy(&upd); /* :750 */
    const int crow = upd.rows - 1; /* :754 */
    const int type = (int)con[conLen - 1];
    const int reverse = (type == 1);
    for (int q = 0; q < nb; q++) { /* :756-796 */
        int colIndex = basic[q];
        double coefficient = orc_round4(AT(&outp, crow, colIndex));
        if (fabs(coefficient) > BB_EPS) {
            int pivotRow = -1;
            for (int rowIndex = 0; rowIndex < outp.rows - 1; rowIndex++)
                if (fabs(orc_round4(AT(&outp, rowIndex, colIndex)) - 1.0) <= BB_EPS) {
                    pivotRow = rowIndex;
                    break;
                }
            if (pivotRow >= 0) {
                for (int col = 0; col < outp.cols; col++) {
                    double pivotVal = orc_round4(AT(&outp, pivotRow, col));
                    double constraintVal = orc_round4(AT(&outp, crow, col));
                    double prod, newVal;
                    if (reverse) {
                        prod = coefficient * constraintVal;
                        newVal = pivotVal - prod;
                    } else {
                        prod = coefficient * pivotVal;
                        newVal = constraintVal - prod;
                    }
                    AT(&outp, crow, col) = orc_round4(newVal);
                }
            }
        }
    }
    tab_round(&outp); /* :799 */
    free(basic);
    tab_free(&working);
    tab_free(&upd);
    return outp;
}

/* CheckIntegerBasicVar :805-857 / ExtractSolution :899-921 share the decision-value scan. */
static void decision_values(const Tab* T, int nvars, double* vals) {
    for (int i = 0; i < nvars; i++) {
        vals[i] = 0.0;
        for (int j = 0; j < T->rows; j++) {
            double v = orc_round4(AT(T, j, i));
            if (fabs(v - 1.0) <= BB_EPS) {
                vals[i] = orc_round4(AT(T, j, T->cols - 1));
                break;
            }
        }
    }
}

/* ------------------------------------------------------------------ ExecuteBranchAndBound */

typedef struct {
    Tab tab;        /* last tableau of the node's list */
    int depth;
    int id;         /* index into the node-record arrays */
} StackNode;

/*
 * BranchAndBoundAdapter.SolveFromPrimal :9-24 + ExecuteBranchAndBound :1006-1233.
 *   final_tableau: the primal solver's FinalTableau (rows x cols), nvars = SolutionVector.Count.
 *   node_cap: 20 in the reference (:1038); > 0 lifts/changes it, <= 0 means 20.
 * Outputs:
 *   x[nvars], *z (-inf and *found = 0 if no integer solution), *best_node (record id of the
 *   optimal branch), *processed (branchCount);
 *   node records (cap node_rec_cap): parent id, kind (0 root, 1 lower, 2 upper), depth, branch var,
 *   bound, status (0 solved / 1 infeasible / 2 failed), z (rounded objective of the solved node);
 *   pop order (ids) in pop_order[0..*processed);
 *   pivot trace (node id, phase, row, col) as 4-int records.
 * Returns ORC_OK_OPTIMAL, or ORC_BB_NODE_CAP if the loop stopped at the cap with nodes left.
 */
int orc_bb_solve(const double* final_tableau, int rows, int cols, int nvars, int enable_pruning,
                 int node_cap, double* x, double* z, int* found, int* best_node,
                 int* processed, int32_t* rec_parent, int32_t* rec_kind, int32_t* rec_depth,
                 int32_t* rec_var, double* rec_bound, int32_t* rec_status, double* rec_z,
                 int node_rec_cap, int* n_records, int32_t* pop_order, int32_t* piv_trace,
                 int64_t piv_cap, int64_t* n_piv) {
    if (node_cap <= 0) node_cap = 20;
    PivTrace tr;
    tr.buf = piv_trace;
    tr.cap = piv_cap;
    tr.n = 0;

    int nrec = 0;
#define REC(parent, kind, depth, var, bound, status, zz)                 \
    do {                                                                 \
        if (nrec < node_rec_cap) {                                       \
            if (rec_parent) rec_parent[nrec] = (parent);                 \
            if (rec_kind) rec_kind[nrec] = (kind);                       \
            if (rec_depth) rec_depth[nrec] = (depth);                    \
            if (rec_var) rec_var[nrec] = (var);                          \
            if (rec_bound) rec_bound[nrec] = (bound);                    \
            if (rec_status) rec_status[nrec] = (status);                 \
            if (rec_z) rec_z[nrec] = (zz);                               \
        }                                                                \
        nrec++;                                                          \
    } while (0)

    Tab root = tab_new(rows, cols);
    memcpy(root.a, final_tableau, sizeof(double) * (size_t)rows * cols);
    tab_round(&root); /* :1021 */

    double optimalValue = -INFINITY; /* :1024 (isMinimization is always false, :695,:24) */
    int haveOptimal = 0;
    double* optimalSolution = (double*)calloc(nvars > 0 ? nvars : 1, sizeof(double));
    int optimalNode = -1;
    int branchCount = 0;

    int scap = 64, sp = 0;
    StackNode* stack = (StackNode*)malloc(sizeof(StackNode) * scap);
    REC(-1, 0, 0, -1, 0.0, 0, orc_round4(AT(&root, 0, cols - 1)));
    stack[sp].tab = root;
    stack[sp].depth = 0;
    stack[sp].id = 0;
    sp++;

    double* vals = (double*)malloc(sizeof(double) * (nvars > 0 ? nvars : 1));
    double* con = (double*)malloc(sizeof(double) * (nvars + 2));
    int iteration = 0;
    int status = ORC_OK_OPTIMAL;

    while (sp > 0) {
        iteration++;
        if (iteration > node_cap) { status = ORC_BB_NODE_CAP; break; } /* :1038-1042 */
        StackNode node = stack[--sp];
        if (pop_order && branchCount < node_rec_cap) pop_order[branchCount] = node.id;
        branchCount++;
        tab_round(&node.tab); /* :1047 */

        double objVal = orc_round4(AT(&node.tab, 0, node.tab.cols - 1)); /* GetObjective :892-897 */
        if (enable_pruning) { /* ShouldPrunebranch :985-1004 */
            if (haveOptimal && objVal <= optimalValue) {
                tab_free(&node.tab);
                continue;
            }
        }
        /* UpdateOptimalSolution :935-983 */
        decision_values(&node.tab, nvars, vals);
        int allInt = 1;
        for (int i = 0; i < nvars; i++)
            if (!is_integer(vals[i])) { allInt = 0; break; }
        if (allInt && objVal > optimalValue) {
            optimalValue = objVal;
            memcpy(optimalSolution, vals, sizeof(double) * nvars);
            haveOptimal = 1;
            optimalNode = node.id;
        }
        /* CreateBranches :859-890 via CheckIntegerBasicVar :805-857 */
        int bestVar = -1;
        double bestValue = 0, minDist = INFINITY;
        for (int i = 0; i < nvars; i++) {
            if (!is_integer(vals[i])) {
                double frac = vals[i] - floor(vals[i]);
                double dist = fabs(frac - 0.5);
                if (dist < minDist) {
                    minDist = dist;
                    bestVar = i;
                    bestValue = vals[i];
                }
            }
        }
        if (bestVar < 0) { /* :1070-1076 integer node */
            tab_free(&node.tab);
            continue;
        }
        int upperInt = orc_to_int32(ceil(bestValue));
        int lowerInt = orc_to_int32(floor(bestValue));

        StackNode kids[2];
        int nk = 0;
        for (int side = 0; side < 2; side++) { /* lower :1083-1148, upper :1150-1208 */
            for (int i = 0; i < nvars; i++) con[i] = (i == bestVar) ? 1.0 : 0.0;
            con[nvars] = side == 0 ? (double)lowerInt : (double)upperInt;
            con[nvars + 1] = side == 0 ? 0.0 : 1.0;
            Tab adj = add_constraint(&node.tab, con, nvars + 2);
            Tab last;
            int npiv = 0;
            int rid = nrec;
            int rc = do_dual_simplex(adj, &last, &npiv, rid, &tr);
            if (rc == 0) {
                tab_round(&last); /* RoundAllTableaux :1124,:1187 */
                REC(node.id, side + 1, node.depth + 1, bestVar, con[nvars], 0,
                    orc_round4(AT(&last, 0, last.cols - 1)));
                kids[nk].tab = last;
                kids[nk].depth = node.depth + 1;
                kids[nk].id = rid;
                nk++;
            } else {
                REC(node.id, side + 1, node.depth + 1, bestVar, con[nvars], rc, 0.0);
            }
        }
        for (int i = nk - 1; i >= 0; i--) { /* :1210-1213 */
            if (sp == scap) {
                scap *= 2;
                stack = (StackNode*)realloc(stack, sizeof(StackNode) * scap);
            }
            stack[sp++] = kids[i];
        }
        tab_free(&node.tab);
    }
    while (sp > 0) tab_free(&stack[--sp].tab);

    if (x) memcpy(x, optimalSolution, sizeof(double) * nvars);
    if (z) *z = optimalValue;
    if (found) *found = haveOptimal;
    if (best_node) *best_node = optimalNode;
    if (processed) *processed = branchCount;
    if (n_records) *n_records = nrec;
    if (n_piv) *n_piv = tr.n;
    free(stack);
    free(vals);
    free(con);
    free(optimalSolution);
    return status;
#undef REC
}

/* Exposed for unit tests of the pieces. */
int orc_bb_add_constraint(const double* base, int rows, int cols, const double* con, int conLen,
                          double* out /* (rows+1) x (cols+1) */) {
    Tab b = tab_new(rows, cols);
    memcpy(b.a, base, sizeof(double) * (size_t)rows * cols);
    Tab o = add_constraint(&b, con, conLen);
    memcpy(out, o.a, sizeof(double) * (size_t)o.rows * o.cols);
    tab_free(&b);
    tab_free(&o);
    return 0;
}

/* RoundTableau :552-567 in place (RoundAllTableaux after a child's DoDualSimplex, :1124,:1187). */
void orc_bb_round_tableau(double* T, int rows, int cols) {
    Tab t;
    t.a = T;
    t.rows = rows;
    t.cols = cols;
    tab_round(&t);
}

/* What ExecuteBranchAndBound does with a popped node before it branches: RoundAllTableaux :1047
 * (T is rounded IN PLACE), GetObjective :892-897 -> *z, the decision values of
 * CheckIntegerBasicVar :805-857 / ExtractSolution :899-921 -> vals[nvars]. */
void orc_bb_node_info(double* T, int rows, int cols, int nvars, double* z, double* vals) {
    Tab t;
    t.a = T;
    t.rows = rows;
    t.cols = cols;
    tab_round(&t);
    if (z) *z = orc_round4(AT(&t, 0, cols - 1));
    if (vals) decision_values(&t, nvars, vals);
}

/* DoDualSimplex on one tableau: returns 0/1/2 as do_dual_simplex; `out` gets the last tableau. */
int orc_bb_dual_simplex(const double* start, int rows, int cols, double* out, int* npiv,
                        int32_t* piv_trace, int64_t piv_cap, int64_t* n_piv) {
    Tab s = tab_new(rows, cols);
    memcpy(s.a, start, sizeof(double) * (size_t)rows * cols);
    PivTrace tr;
    tr.buf = piv_trace;
    tr.cap = piv_cap;
    tr.n = 0;
    Tab last;
    int np = 0;
    int rc = do_dual_simplex(s, &last, &np, 0, &tr);
    if (rc == 0) {
        memcpy(out, last.a, sizeof(double) * (size_t)rows * cols);
        tab_free(&last);
    }
    if (npiv) *npiv = np;
    if (n_piv) *n_piv = tr.n;
    return rc;
}
