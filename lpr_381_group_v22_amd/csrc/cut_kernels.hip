// cut_kernels.hip -- gfx950 kernels + C ABI of the cutting-plane side path ("next" row f3):
//   DualSimplexSolver.Solve        LPR_381_Group_V22/Simplex/DualSimplex.cs:14-114
//   PrimalSimplexSolver2.Solve     LPR_381_Group_V22/Simplex/PrimalSimplexSolver2.cs:46-97
//   CuttingPlaneSolver             LPR_381_Group_V22/IntegerProgramming/CuttingPlaneSolver.cs:64-229
// They work on the same device tableau as the primal path (row 0 = objectiveRow, rows 1.. =
// constraintRows).  Same pivot shape as PrimalSimplexSolver with two differences: rows whose
// factor is within 1e-9 of zero are left untouched (DualSimplex.cs:166, PrimalSimplexSolver2.cs:160)
// and every selection is an EPS-band sequential fold ("better by more than EPS"), replayed exactly
// by the next-take search also used by the revised solver.  Not pipelined: this path is dead code
// in the reference's menu (Program.cs:417-428); parity is the bar here, not the roofline.
#include "engine_common.hpp"

#include <new>

#pragma clang fp contract(off)

namespace lpr {

constexpr double kCutEps = 1e-9;  // DualSimplex.cs:8, PrimalSimplexSolver2.cs, CuttingPlaneSolver.cs:10

struct CutState {
    int32_t status;      // kRunning or an lpr_status
    int32_t pr, pc;      // pivot (tableau row, column) of the pending update
    int32_t print_steps; // `iter` only advances when printSteps is set (DualSimplex.cs:94)
    int32_t pending;     // a pivot has been staged/applied and not yet counted
    int32_t pad;
    int64_t iter;        // the C# `iter`
    int64_t done;        // pivots performed
    int64_t max_iters;   // maxIters (:108 / :90)
    int64_t hard_cap;    // extra stop (no C# counterpart), <= 0: none
    int64_t log_n, log_cap;
    int32_t scratch[8];  // cutting-plane step results
};

__device__ __forceinline__ int cut_block_min_int(int v, int* lds) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int nwaves = blockDim.x / kWave;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, kWave));
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    int r = lds[0];
    for (int w = 1; w < nwaves; ++w) r = min(r, lds[w]);
    return r;
}

// |num / a| over the negative entries of row `prow` with a non-zero objective entry; NaN = skip
__device__ __forceinline__ double dual_ratio(const double* __restrict__ T, int ld, int prow,
                                             int j) {
    const double a = T[(size_t)prow * ld + j];
    if (!(a < -kCutEps)) return NAN;
    const double num = T[j];
    if (!(fabs(num) > kCutEps)) return NAN;
    return fabs(ieee_div(num, a));
}

// Replays `for j ascending: if (ratio < best - EPS || (|ratio - best| <= EPS && (pc == -1 ||
// j < pc))) take` (DualSimplex.cs:53-70, CuttingPlaneSolver.cs:116-132) on row `prow`.
__device__ int fold_dual_column(const double* __restrict__ T, int ld, int C, int prow, int* lds) {
    const int tid = threadIdx.x, nt = blockDim.x;
    int pc = -1;
    double best = INFINITY;
    for (;;) {
        int first = INT_MAX;
        for (int j = tid; j < C - 1; j += nt) {
            if (j <= pc) continue;
            const double ratio = dual_ratio(T, ld, prow, j);
            if (ratio != ratio) continue;
            if (ratio < best - kCutEps ||
                (fabs(ratio - best) <= kCutEps && (pc == -1 || j < pc))) {
                first = j;
                break;
            }
        }
        first = cut_block_min_int(first, lds);
        if (first == INT_MAX) break;
        pc = first;
        best = dual_ratio(T, ld, prow, pc);
    }
    return pc;
}

enum : int { kCutDual = 0, kCutPrimal2 = 1 };

// One loop head of DualSimplexSolver.Solve (:24-112) or PrimalSimplexSolver2.Solve (:49-96).
__global__ __launch_bounds__(1024) void k_cut_select(double* __restrict__ T, int ld, int R, int C,
                                                     double* __restrict__ rowbuf,
                                                     double* __restrict__ colbuf,
                                                     int32_t* __restrict__ log, CutState* st,
                                                     int mode) {
    __shared__ int lds[16];
    if (st->status != kRunning) return;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int rhs = C - 1;
    int pr = -1, pc = -1;

    // Close the previous pivot first: `if (iter >= maxIters) return false` comes AFTER the pivot in
    // the C# (:108 / :90).  It is done here, not in k_cut_update, so that no update workgroup can
    // see the status change while others of the same launch are still running.
    int64_t done_now;
    {
        const int pending = st->pending;
        const int64_t it = st->iter, mx = st->max_iters;
        done_now = st->done + (pending ? 1 : 0);
        __syncthreads();
        if (pending) {
            if (tid == 0) {
                st->pending = 0;
                st->done += 1;
            }
            if (it >= mx) {
                if (tid == 0) st->status = LPR_PIVOT_LIMIT;
                return;
            }
        }
    }

    if (mode == kCutDual) {
        // pivot row: `rhs < mostNeg - EPS || (|rhs - mostNeg| <= EPS && pivotRow != -1 &&
        // r < pivotRow)` over constraint rows r ascending (:29-37)
        int prow = -1;  // constraint index
        double mostNeg = 0.0;
        for (;;) {
            int first = INT_MAX;
            for (int r = tid; r < R - 1; r += nt) {
                if (r <= prow) continue;
                const double v = T[(size_t)(r + 1) * ld + rhs];
                if (v < mostNeg - kCutEps ||
                    (fabs(v - mostNeg) <= kCutEps && prow != -1 && r < prow)) {
                    first = r;
                    break;
                }
            }
            first = cut_block_min_int(first, lds);
            if (first == INT_MAX) break;
            prow = first;
            mostNeg = T[(size_t)(prow + 1) * ld + rhs];
        }
        if (prow < 0) {
            if (tid == 0) st->status = LPR_OK_OPTIMAL;  // "Dual phase complete" :40-44
            return;
        }
        pr = prow + 1;
        pc = fold_dual_column(T, ld, C, pr, lds);
        if (pc < 0) {
            if (tid == 0) st->status = LPR_INFEASIBLE_BASIS;  // return false :72-76
            return;
        }
    } else {
        // entering column: `c < mostNeg - EPS || (...)` (:102-117)
        double mostNeg = 0.0;
        for (;;) {
            int first = INT_MAX;
            for (int j = tid; j < rhs; j += nt) {
                if (j <= pc) continue;
                const double c = T[j];
                if (c < mostNeg - kCutEps ||
                    (fabs(c - mostNeg) <= kCutEps && pc != -1 && j < pc)) {
                    first = j;
                    break;
                }
            }
            first = cut_block_min_int(first, lds);
            if (first == INT_MAX) break;
            pc = first;
            mostNeg = T[pc];
        }
        if (pc < 0) {
            if (tid == 0) st->status = LPR_OK_OPTIMAL;  // :54-60
            return;
        }
        // leaving row (:120-141), with the C#'s operator precedence:
        // (ratio > EPS && ratio < best - EPS) || ((|ratio - best| <= EPS && bestRow == -1) ? true
        //                                                                           : i < bestRow)
        double best = INFINITY;
        for (;;) {
            int first = INT_MAX;
            for (int i = 1 + tid; i < R; i += nt) {
                if (i <= pr) continue;
                const double a = T[(size_t)i * ld + pc];
                if (!(a > kCutEps)) continue;
                const double ratio = ieee_div(T[(size_t)i * ld + rhs], a);
                const bool second = (fabs(ratio - best) <= kCutEps && pr == -1) ? true : (i < pr);
                if ((ratio > kCutEps && ratio < best - kCutEps) || second) {
                    first = i;
                    break;
                }
            }
            first = cut_block_min_int(first, lds);
            if (first == INT_MAX) break;
            pr = first;
            best = ieee_div(T[(size_t)pr * ld + rhs], T[(size_t)pr * ld + pc]);
        }
        if (pr < 0) {
            if (tid == 0) st->status = LPR_UNBOUNDED;  // return false :63-68
            return;
        }
    }

    if (st->hard_cap > 0 && done_now >= st->hard_cap) {
        if (tid == 0) st->status = LPR_PIVOT_LIMIT;
        return;
    }
    const double piv = T[(size_t)pr * ld + pc];
    if (fabs(piv) <= kCutEps) {
        if (tid == 0) st->status = LPR_PIVOT_TOO_SMALL;  // InvalidOperationException :155 / :148
        return;
    }
    for (int j = tid; j < ld; j += nt) rowbuf[j] = (j < C) ? ieee_div(T[(size_t)pr * ld + j], piv) : 0.0;
    for (int i = tid; i < R; i += nt) colbuf[i] = T[(size_t)i * ld + pc];
    if (tid == 0) {
        st->pr = pr;
        st->pc = pc;
        st->pending = 1;  // closed by the next k_cut_select
        if (st->print_steps) st->iter += 1;  // ++iter inside `if (printSteps)` (:94 / :75)
        if (st->log_n < st->log_cap) {
            int32_t* e = log + 3 * st->log_n;
            e[0] = mode;
            e[1] = (mode == kCutDual) ? pr - 1 : pr;  // the C#'s own row numbering
            e[2] = pc;
        }
        st->log_n += 1;
    }
}

// Pivot with the row skip (DualSimplex.cs:150-178, PrimalSimplexSolver2.cs:145-164): row i is
// rewritten only if |f_i| > EPS.
template <int TR>
__global__ __launch_bounds__(256) void k_cut_update(double* __restrict__ T, int ld, int R,
                                                    const double* __restrict__ rowbuf,
                                                    const double* __restrict__ colbuf,
                                                    CutState* st, int check_status, int finish) {
    if (check_status && st->status != kRunning) return;
    const int ld2 = ld >> 1;
    const int c2 = blockIdx.x * blockDim.x + threadIdx.x;
    const int i0 = blockIdx.y * TR;
    const int r = st->pr;
    (void)finish;  // bookkeeping lives in k_cut_select (no state is written by this kernel)
    if (c2 >= ld2) return;
    const double2 pr2 = reinterpret_cast<const double2*>(rowbuf)[c2];
    double2* __restrict__ T2 = reinterpret_cast<double2*>(T);
#pragma unroll
    for (int k = 0; k < TR; ++k) {
        const int i = i0 + k;
        if (i >= R) break;
        if (i == r) {
            T2[(size_t)i * ld2 + c2] = pr2;
            continue;
        }
        const double f = colbuf[i];
        if (!(fabs(f) > kCutEps)) continue;  // the row is not touched
        double2 x = T2[(size_t)i * ld2 + c2];
        const double px = f * pr2.x;
        const double py = f * pr2.y;
        x.x = x.x - px;
        x.y = x.y - py;
        T2[(size_t)i * ld2 + c2] = x;
    }
}

__device__ __forceinline__ double cut_frac(double a) {  // CuttingPlaneSolver.cs:12-17
    const double f = a - floor(a);
    if (fabs(f) < kCutEps || fabs(1 - f) < kCutEps) return 0.0;
    return f;
}

// CuttingPlaneSolution steps 1-6 (:76-138): choose the constraint whose fractional RHS is closest
// to 0.5 (first in index order among equals -- see oracle/oracle_cut.c on List.Sort), append
// cut = -frac(row) as tableau row R, pick the pivot column on it.  scratch[0] = chosen constraint
// (-1: all RHS integral), scratch[1] = pivot column (-1: none).  Also stages rowbuf / colbuf for
// the pivot on the cut.
__global__ __launch_bounds__(1024) void k_cut_add(double* __restrict__ T, int ld, int R, int C,
                                                  double* __restrict__ rowbuf,
                                                  double* __restrict__ colbuf,
                                                  int32_t* __restrict__ log, CutState* st) {
    __shared__ int lds[16];
    __shared__ double lds_v[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int rhs = C - 1;
    // lexicographic min of (|frac - 0.5|, index) over rows with frac > EPS
    double bk = INFINITY;
    int bi = INT_MAX;
    for (int i = tid; i < R - 1; i += nt) {
        const double fr = cut_frac(T[(size_t)(i + 1) * ld + rhs]);
        if (fr > kCutEps) {
            const double key = fabs(fr - 0.5);
            if (bi == INT_MAX || key < bk) {
                bk = key;
                bi = i;
            }
        }
    }
    {  // block reduce (value, index) -- associative: smaller key, then smaller index
        const int lane = tid & (kWave - 1), wave = tid / kWave, nw = nt / kWave;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ok = __shfl_xor(bk, off, kWave);
            const int oi = __shfl_xor(bi, off, kWave);
            if (oi != INT_MAX && (bi == INT_MAX || ok < bk || (ok == bk && oi < bi))) {
                bk = ok;
                bi = oi;
            }
        }
        __syncthreads();
        if (lane == 0) {
            lds_v[wave] = bk;
            lds[wave] = bi;
        }
        __syncthreads();
        bk = lds_v[0];
        bi = lds[0];
        for (int w = 1; w < nw; ++w) {
            const double ok = lds_v[w];
            const int oi = lds[w];
            if (oi != INT_MAX && (bi == INT_MAX || ok < bk || (ok == bk && oi < bi))) {
                bk = ok;
                bi = oi;
            }
        }
        __syncthreads();
    }
    if (bi == INT_MAX) {
        if (tid == 0) {
            st->scratch[0] = -1;
            st->scratch[1] = -1;
        }
        return;
    }
    const int chosen = bi;
    for (int j = tid; j < ld; j += nt)  // cut row (:99-110), padding stays 0
        T[(size_t)R * ld + j] = (j < C) ? -cut_frac(T[(size_t)(chosen + 1) * ld + j]) : 0.0;
    __syncthreads();
    __threadfence_block();
    const int pc = fold_dual_column(T, ld, C, R, lds);  // :113-132 on the new row
    if (tid == 0) {
        st->scratch[0] = chosen;
        st->scratch[1] = pc;
    }
    if (pc < 0) return;
    const double piv = T[(size_t)R * ld + pc];
    if (fabs(piv) <= kCutEps) {  // :145-150
        if (tid == 0) st->scratch[1] = -2;
        return;
    }
    for (int j = tid; j < ld; j += nt) rowbuf[j] = (j < C) ? ieee_div(T[(size_t)R * ld + j], piv) : 0.0;
    for (int i = tid; i < R + 1; i += nt) colbuf[i] = T[(size_t)i * ld + pc];
    if (tid == 0) {
        st->pr = R;
        st->pc = pc;
        if (st->log_n < st->log_cap) {
            int32_t* e = log + 3 * st->log_n;
            e[0] = 2;
            e[1] = R - 1;  // cutRowIdx (constraint index)
            e[2] = pc;
        }
        st->log_n += 1;
    }
}

// needDual / needPrimal / anyFractional (:183-184, :215-217) -> scratch[2..4]
__global__ __launch_bounds__(1024) void k_cut_flags(const double* __restrict__ T, int ld, int R,
                                                    int C, CutState* st) {
    const int tid = threadIdx.x, nt = blockDim.x;
    int neg = 0, nonopt = 0, frac = 0;
    for (int i = 1 + tid; i < R; i += nt) {
        const double v = T[(size_t)i * ld + (C - 1)];
        if (v < -kCutEps) neg = 1;
        if (cut_frac(v) > kCutEps) frac = 1;
    }
    for (int j = tid; j < C - 1; j += nt)
        if (T[j] < -kCutEps) nonopt = 1;
    neg = __syncthreads_or(neg);
    nonopt = __syncthreads_or(nonopt);
    frac = __syncthreads_or(frac);
    if (tid == 0) {
        st->scratch[2] = neg;
        st->scratch[3] = nonopt;
        st->scratch[4] = frac;
    }
}

}  // namespace lpr

// ---------------------------------------------------------------------------------------------
// host side

struct lpr_cut_ctx {  // per-tableau scratch of this path, hung off the tableau lazily
    lpr::CutState* state = nullptr;
    lpr::CutState* h_state = nullptr;
    int32_t* log = nullptr;
    int64_t log_cap = 0;
    int row_cap = 0;  // rows the tableau buffer can hold
};

using namespace lpr;

namespace {

int cut_ensure(lpr_tableau* t) {
    if (t->cut) return LPR_OK_OPTIMAL;
    lpr_cut_ctx* c = new (std::nothrow) lpr_cut_ctx();
    if (!c) return LPR_OUT_OF_MEMORY;
    c->log_cap = 1 << 16;
    c->row_cap = t->rows;
    hipError_t err = hipMalloc(&c->state, sizeof(CutState));
    if (err == hipSuccess) err = hipHostMalloc(&c->h_state, sizeof(CutState));
    if (err == hipSuccess) err = hipMalloc(&c->log, (size_t)c->log_cap * 3 * sizeof(int32_t));
    if (err != hipSuccess) {
        set_error("cut-path scratch allocation failed: %s", hipGetErrorString(err));
        hipFree(c->state);
        hipFree(c->log);
        if (c->h_state) hipHostFree(c->h_state);
        delete c;
        return LPR_DEVICE_ERROR;
    }
    std::memset(c->h_state, 0, sizeof(CutState));
    c->h_state->log_cap = c->log_cap;
    t->cut = c;
    return LPR_OK_OPTIMAL;
}

// make room for `rows_needed` tableau rows (the cut appends one row per call)
int cut_grow_rows(lpr_tableau* t, int rows_needed) {
    lpr_cut_ctx* c = static_cast<lpr_cut_ctx*>(t->cut);
    if (rows_needed <= c->row_cap) return LPR_OK_OPTIMAL;
    hipStream_t s = t->eng->stream;
    const int cap = rows_needed + 8;
    double *nT = nullptr, *ncol = nullptr;
    LPR_HIP(hipMalloc(&nT, (size_t)cap * t->ld * sizeof(double)));
    LPR_HIP(hipMemsetAsync(nT, 0, (size_t)cap * t->ld * sizeof(double), s));
    LPR_HIP(hipMemcpyAsync(nT, t->T, (size_t)t->rows * t->ld * sizeof(double),
                           hipMemcpyDeviceToDevice, s));
    LPR_HIP(hipMalloc(&ncol, (size_t)align_up(cap, 16) * sizeof(double)));
    LPR_HIP(hipStreamSynchronize(s));
    hipFree(t->T);
    hipFree(t->colbuf);
    hipFree(t->T2);  // the fused path's second buffer no longer matches the shape
    t->T2 = nullptr;
    t->T = nT;
    t->colbuf = ncol;
    // next_col / next_rhs / basis belong to the pipelined primal path; re-size them too so that a
    // later lpr_primal_solve on the grown tableau stays valid
    double *nc = nullptr, *nr = nullptr;
    int32_t* nb = nullptr;
    LPR_HIP(hipMalloc(&nc, (size_t)align_up(cap, 16) * sizeof(double)));
    LPR_HIP(hipMalloc(&nr, (size_t)align_up(cap, 16) * sizeof(double)));
    LPR_HIP(hipMalloc(&nb, (size_t)cap * sizeof(int32_t)));
    LPR_HIP(hipMemsetAsync(nb, 0xff, (size_t)cap * sizeof(int32_t), s));
    if (t->rows > 1)
        LPR_HIP(hipMemcpyAsync(nb, t->basis, (size_t)(t->rows - 1) * sizeof(int32_t),
                               hipMemcpyDeviceToDevice, s));
    LPR_HIP(hipStreamSynchronize(s));
    hipFree(t->next_col);
    hipFree(t->next_rhs);
    hipFree(t->basis);
    t->next_col = nc;
    t->next_rhs = nr;
    t->basis = nb;
    if (t->graph) {  // (a stale graph would also be rejected by its key: rows / T changed)
        hipGraphExecDestroy(t->graph);
        t->graph = nullptr;
        t->graph_batch = 0;
        t->graph_variant = -1;
        t->graph_key = lpr_tableau::GraphKey();
    }
    c->row_cap = cap;
    return LPR_OK_OPTIMAL;
}

void cut_launch_update(lpr_tableau* t, int check_status, int finish) {
    lpr_cut_ctx* c = static_cast<lpr_cut_ctx*>(t->cut);
    constexpr int TR = 8;
    dim3 grid((t->ld / 2 + 255) / 256, (t->rows + TR - 1) / TR);
    hipLaunchKernelGGL((k_cut_update<TR>), grid, dim3(256), 0, t->eng->stream, t->T, t->ld, t->rows,
                       t->rowbuf, t->colbuf, c->state, check_status, finish);
}

// DualSimplexSolver.Solve / PrimalSimplexSolver2.Solve driver; returns the lpr_status
int cut_run_solver(lpr_tableau* t, int mode, int max_iters, int print_steps, int64_t hard_cap,
                   int64_t* pivots) {
    lpr_cut_ctx* c = static_cast<lpr_cut_ctx*>(t->cut);
    hipStream_t s = t->eng->stream;
    CutState* hs = c->h_state;
    const int64_t log_n = hs->log_n;
    std::memset(hs, 0, sizeof(CutState));
    hs->status = kRunning;
    hs->print_steps = print_steps ? 1 : 0;
    hs->max_iters = max_iters;
    hs->hard_cap = hard_cap;
    hs->log_n = log_n;
    hs->log_cap = c->log_cap;
    LPR_HIP(hipMemcpyAsync(c->state, hs, sizeof(CutState), hipMemcpyHostToDevice, s));
    const int batch = 16;
    for (;;) {
        for (int k = 0; k < batch; ++k) {
            hipLaunchKernelGGL(k_cut_select, dim3(1), dim3(1024), 0, s, t->T, t->ld, t->rows,
                               t->cols, t->rowbuf, t->colbuf, c->log, c->state, mode);
            cut_launch_update(t, 1, 1);
        }
        LPR_HIP(hipGetLastError());
        LPR_HIP(hipMemcpyAsync(hs, c->state, sizeof(CutState), hipMemcpyDeviceToHost, s));
        LPR_HIP(hipStreamSynchronize(s));
        if (hs->status != kRunning) break;
    }
    if (pivots) *pivots = hs->done;
    return hs->status;
}

}  // namespace

void lpr_cut_release(lpr_tableau* t) {  // called from release_device (lpr_engine.hip)
    lpr_cut_ctx* c = static_cast<lpr_cut_ctx*>(t->cut);
    if (!c) return;
    hipFree(c->state);
    hipFree(c->log);
    if (c->h_state) hipHostFree(c->h_state);
    delete c;
    t->cut = nullptr;
}

#define LPR_LIVE_T(t)                                                            \
    do {                                                                         \
        if (!(t) || !(t)->eng) {                                                 \
            set_error("tableau handle is null or its engine has been closed");   \
            return LPR_BAD_ARGUMENT;                                             \
        }                                                                        \
    } while (0)

extern "C" {

int lpr_dual_solve(lpr_tableau* t, int max_iters, int print_steps, int64_t hard_cap,
                   lpr_solve_result* res) {
    LPR_LIVE_T(t);
    if (!res || t->rows < 2) {
        set_error("lpr_dual_solve: no constraint rows");  // ArgumentException :17
        return LPR_BAD_ARGUMENT;
    }
    LPR_HIP(hipSetDevice(t->eng->device));
    int rc = cut_ensure(t);
    if (rc != LPR_OK_OPTIMAL) return rc;
    int64_t piv = 0;
    const int st = cut_run_solver(t, kCutDual, max_iters, print_steps, hard_cap, &piv);
    if (st < 0) return st;
    res->status = st;
    res->block = 1;
    res->pivots = piv;
    res->total_pivots = piv;
    res->z = 0.0;
    LPR_HIP(hipMemcpy(&res->z, t->T + (t->cols - 1), sizeof(double), hipMemcpyDeviceToHost));
    return st;
}

int lpr_primal2_solve(lpr_tableau* t, int max_iters, int print_steps, int64_t hard_cap,
                      lpr_solve_result* res) {
    LPR_LIVE_T(t);
    if (!res || t->rows < 2) {
        set_error("lpr_primal2_solve: no constraint rows");  // ArgumentException :27
        return LPR_BAD_ARGUMENT;
    }
    LPR_HIP(hipSetDevice(t->eng->device));
    int rc = cut_ensure(t);
    if (rc != LPR_OK_OPTIMAL) return rc;
    int64_t piv = 0;
    const int st = cut_run_solver(t, kCutPrimal2, max_iters, print_steps, hard_cap, &piv);
    if (st < 0) return st;
    res->status = st;
    res->block = 1;
    res->pivots = piv;
    res->total_pivots = piv;
    res->z = 0.0;
    LPR_HIP(hipMemcpy(&res->z, t->T + (t->cols - 1), sizeof(double), hipMemcpyDeviceToHost));
    return st;
}

int lpr_cutting_plane(lpr_tableau* t, int max_cuts, int64_t hard_cap, int32_t* exit_code,
                      int32_t* cuts) {
    LPR_LIVE_T(t);
    if (!exit_code || t->rows < 2) {
        set_error("lpr_cutting_plane: no constraint rows");  // ArgumentException :68
        return LPR_BAD_ARGUMENT;
    }
    LPR_HIP(hipSetDevice(t->eng->device));
    int rc = cut_ensure(t);
    if (rc != LPR_OK_OPTIMAL) return rc;
    if (max_cuts <= 0) max_cuts = 64;
    lpr_cut_ctx* c = static_cast<lpr_cut_ctx*>(t->cut);
    hipStream_t s = t->eng->stream;
    CutState* hs = c->h_state;
    int ncuts = 0, ex = 5;
    for (;;) {
        if (ncuts >= max_cuts) {
            // do not add a cut we are not allowed to; but "all integral" still wins (exit 1)
            hipLaunchKernelGGL(k_cut_flags, dim3(1), dim3(1024), 0, s, t->T, t->ld, t->rows,
                               t->cols, c->state);
            LPR_HIP(hipMemcpyAsync(hs, c->state, sizeof(CutState), hipMemcpyDeviceToHost, s));
            LPR_HIP(hipStreamSynchronize(s));
            ex = hs->scratch[4] ? 6 : 1;
            break;
        }
        rc = cut_grow_rows(t, t->rows + 1);
        if (rc != LPR_OK_OPTIMAL) return rc;
        // steps 1-6 (+ staging of the pivot on the cut)
        hs->status = LPR_OK_OPTIMAL;
        LPR_HIP(hipMemcpyAsync(&c->state->log_n, &hs->log_n, 2 * sizeof(int64_t),
                               hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_cut_add, dim3(1), dim3(1024), 0, s, t->T, t->ld, t->rows, t->cols,
                           t->rowbuf, t->colbuf, c->log, c->state);
        LPR_HIP(hipGetLastError());
        LPR_HIP(hipMemcpyAsync(hs, c->state, sizeof(CutState), hipMemcpyDeviceToHost, s));
        LPR_HIP(hipStreamSynchronize(s));
        if (hs->scratch[0] < 0) { ex = 1; break; }  // "All RHS are integers" :87-91
        t->rows += 1;                                // the cut row is part of the tableau now
        ncuts += 1;
        if (hs->scratch[1] == -1) { ex = 2; break; }  // no valid pivot column :134-138
        if (hs->scratch[1] == -2) { ex = 3; break; }  // pivot too small :146-150
        cut_launch_update(t, 0, 0);                   // step 7
        hipLaunchKernelGGL(k_cut_flags, dim3(1), dim3(1024), 0, s, t->T, t->ld, t->rows, t->cols,
                           c->state);
        LPR_HIP(hipGetLastError());
        LPR_HIP(hipMemcpyAsync(hs, c->state, sizeof(CutState), hipMemcpyDeviceToHost, s));
        LPR_HIP(hipStreamSynchronize(s));
        bool needDual = hs->scratch[2] != 0, needPrimal = hs->scratch[3] != 0;
        if (needDual) {  // :186-194, printSteps: true
            const int st = cut_run_solver(t, kCutDual, 10000, 1, hard_cap, nullptr);
            if (st < 0) return st;
            if (st == LPR_PIVOT_TOO_SMALL) { ex = 7; break; }
            if (st != LPR_OK_OPTIMAL) { ex = 4; break; }
            hipLaunchKernelGGL(k_cut_flags, dim3(1), dim3(1024), 0, s, t->T, t->ld, t->rows,
                               t->cols, c->state);
            LPR_HIP(hipMemcpyAsync(hs, c->state, sizeof(CutState), hipMemcpyDeviceToHost, s));
            LPR_HIP(hipStreamSynchronize(s));
            needPrimal = hs->scratch[3] != 0;
        }
        if (needPrimal) {  // :196-212; the result of Solve is ignored by the C#
            const int st = cut_run_solver(t, kCutPrimal2, 10000, 1, hard_cap, nullptr);
            if (st < 0) return st;
            if (st == LPR_PIVOT_TOO_SMALL) { ex = 7; break; }
        }
        hipLaunchKernelGGL(k_cut_flags, dim3(1), dim3(1024), 0, s, t->T, t->ld, t->rows, t->cols,
                           c->state);
        LPR_HIP(hipMemcpyAsync(hs, c->state, sizeof(CutState), hipMemcpyDeviceToHost, s));
        LPR_HIP(hipStreamSynchronize(s));
        if (!hs->scratch[3] && !hs->scratch[2]) {  // :215
            if (hs->scratch[4]) continue;           // another Gomory cut (:217-222)
            ex = 0;                                 // "Displayed the Optimal Tableau" :224
            break;
        }
        ex = 5;  // "Cutting-plane step finished" :228
        break;
    }
    *exit_code = ex;
    if (cuts) *cuts = ncuts;
    return LPR_OK_OPTIMAL;
}

int lpr_cut_log_read(lpr_tableau* t, int32_t* triples, int64_t cap, int64_t* count) {
    LPR_LIVE_T(t);
    if (!count || cap < 0) return LPR_BAD_ARGUMENT;
    lpr_cut_ctx* c = static_cast<lpr_cut_ctx*>(t->cut);
    int64_t k = c ? c->h_state->log_n : 0;
    if (c && k > c->log_cap) k = c->log_cap;
    if (k > cap) k = cap;
    *count = k;
    if (k == 0 || !triples) return LPR_OK_OPTIMAL;
    LPR_HIP(hipSetDevice(t->eng->device));
    LPR_HIP(hipMemcpy(triples, c->log, (size_t)k * 3 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return LPR_OK_OPTIMAL;
}

}  // extern "C"
