// sens_engine.hip -- gfx950 kernels + C ABI of the sensitivity re-solve ("next" row f4):
//   SensitivityAnalyzer   LPR_381_Group_V22/SensitivityAnalysis/SensitivityAnalyzer.cs
//     ctor :22-39, GetBasicRow/IsPivotColumn :69-84, IsOptimal :86-96, Pivot :98-119,
//     ReOptimize :121-166, DualSimplexIfNeeded :168-201, ResolveAll :203-208, the six edits
//     :300-321 :362-393 :427-470 :502-531 :534-584 :609-659, RebuildBasicsFromTableau :706-723.
// The analyzer owns its own copy of the final tableau (the C# clones it, :24); it lives in HBM
// with the same layout as the primal path (row 0 = Z row, rows padded to 16 doubles).  The edits
// touch O(1) .. O(rows + cols) entries and are done with scalar copies / tiny kernels; what costs
// is the re-solve after them: the dual / primal ratio folds ("better by more than EPS", replayed
// exactly by the next-take search) and the rank-1 pivot with the C#'s |factor| < EPS row skip.
#include "engine_common.hpp"
#include "fold_common.hpp"

#include <algorithm>
#include <new>

#pragma clang fp contract(off)

struct lpr_sens {
    lpr_engine* eng = nullptr;
    int R = 0, C = 0, ld = 0;
    double* T = nullptr;        // R x ld
    double* rowbuf = nullptr;   // ld
    double* colbuf = nullptr;   // R (padded)
    double* sol = nullptr;      // ld: solutionVector after a successful ReOptimize
    double* fold_w = nullptr;   // R: weights of a column fold
    double* fold_io = nullptr;  // ld: init / result of a column fold, also a staged new column
    double* rhsd = nullptr;     // R: dense copy of the RHS column during a run
    double* snapT = nullptr;    // ChangeRHS snapshot of T (lazy, kept until the shape changes)
    int32_t* snapI = nullptr;   // ... of basic + bcount
    int32_t* basic = nullptr;   // R-1 (padded): basicVars
    int32_t* bcount = nullptr;  // ld: how many positions of basicVars hold column j
    int32_t* cnt = nullptr;     // ld: rows 1.. with |T[i][j]| > EPS
    int32_t* rowsum = nullptr;  // ld: sum of those row indices (the row itself when cnt == 1)
    int32_t* cand = nullptr;    // ld: GetBasicRow(j)
    int32_t* log = nullptr;     // 3 * log_cap
    int64_t log_cap = 0, log_n = 0;
    void* state = nullptr;      // SensState, device
    void* h_state = nullptr;    // pinned mirror
    // host mirrors of the small members of the C# object
    std::vector<int32_t> h_basic;
    std::vector<double> h_sol;
    double z = 0.0;
    int64_t pivots = 0;         // pivots of the last edit
};

namespace lpr {

constexpr double kSensEps = 1e-9;     // SensitivityAnalyzer.cs:20
constexpr int kSensMaxIter = 10000;   // default argument of ReOptimize / DualSimplexIfNeeded
constexpr int32_t kNoBasic = 0x7f7f7f7f;  // "no column yet" while basicVars is rebuilt (memset 0x7f)

struct SensState {
    int32_t status;      // kRunning or an lpr_sens_outcome
    int32_t phase;       // 0 DualSimplexIfNeeded, 1 ReOptimize
    int32_t pr, pc;      // pivot of the pending update
    int32_t iter_dual;   // the C#'s `iter` of each loop
    int32_t iter_primal;
    int32_t sweep;       // parity of the update sweep direction
    int32_t pad;
    int64_t done;        // pivots of this run
    int64_t log_n, log_cap;
};

__device__ __forceinline__ int sens_block_min_int(int v, int* lds) {
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

// ---- RebuildBasicsFromTableau / GetBasicRow -------------------------------------------------
// IsPivotColumn(i, j) && |T[i][j] - 1| < EPS  <=>  exactly one row (1..R-1) of column j has
// |v| > EPS, it is row i, and |v - 1| < EPS (a row within EPS of 1 is itself counted).  So one
// pass over the tableau yields GetBasicRow for every column.
__global__ __launch_bounds__(256) void k_sens_colscan(const double* __restrict__ T, int ld, int R,
                                                      int ncols, int rows_per_group,
                                                      int32_t* __restrict__ cnt,
                                                      int32_t* __restrict__ rowsum) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    const int i0 = 1 + blockIdx.y * rows_per_group;
    const int i1 = min(R, i0 + rows_per_group);
    int c = 0, rs = 0;
    for (int i = i0; i < i1; ++i) {
        const double v = T[(size_t)i * ld + j];
        if (fabs(v) > kSensEps) {
            c += 1;
            rs += i;
        }
    }
    if (c) {
        atomicAdd(&cnt[j], c);
        atomicAdd(&rowsum[j], rs);
    }
}

// cand[j] = GetBasicRow(j); optionally basicVars[i-1] = first such j per row (:710-722) and
// solutionVector[j] (:160-165).
__global__ __launch_bounds__(256) void k_sens_cand(const double* __restrict__ T, int ld, int C,
                                                   const int32_t* __restrict__ cnt,
                                                   const int32_t* __restrict__ rowsum,
                                                   int32_t* __restrict__ cand,
                                                   int32_t* __restrict__ basic,
                                                   double* __restrict__ sol) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= C - 1) return;
    int r = -1;
    if (cnt[j] == 1) {
        const int i = rowsum[j];
        if (fabs(T[(size_t)i * ld + j] - 1.0) < kSensEps) r = i;
    }
    cand[j] = r;
    if (basic && r >= 1) atomicMin(&basic[r - 1], j);
    if (sol) sol[j] = (r == -1) ? 0.0 : T[(size_t)r * ld + (C - 1)];
}

__global__ __launch_bounds__(256) void k_sens_basics_finish(int32_t* __restrict__ basic, int m,
                                                            int32_t* __restrict__ bcount) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int b = basic[i];
    if (b == kNoBasic)
        basic[i] = -1;
    else
        atomicAdd(&bcount[b], 1);
}

// GetBasicRow of one column -> out[0]
__global__ __launch_bounds__(1024) void k_sens_basic_row(const double* __restrict__ T, int ld,
                                                         int R, int col, int32_t* out) {
    int c = 0, rs = 0;
    for (int i = 1 + threadIdx.x; i < R; i += blockDim.x)
        if (fabs(T[(size_t)i * ld + col]) > kSensEps) {
            c += 1;
            rs += i;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c += __shfl_xor(c, off, kWave);
        rs += __shfl_xor(rs, off, kWave);
    }
    __shared__ int lc[16], lr[16];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) {
        lc[wave] = c;
        lr[wave] = rs;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tc = 0, tr = 0;
        for (int w = 0; w < (int)(blockDim.x / kWave); ++w) {
            tc += lc[w];
            tr += lr[w];
        }
        int r = -1;
        if (tc == 1 && fabs(T[(size_t)tr * ld + col] - 1.0) < kSensEps) r = tr;
        out[0] = r;
    }
}

// ---- one loop head of DualSimplexIfNeeded (:171-200) or ReOptimize (:124-157) --------------
// (the EPS-band folds are eps_fold, fold_common.hpp)
// dense copy of one tableau column (the RHS before a run)
__global__ __launch_bounds__(256) void k_sens_gather_col(const double* __restrict__ T, int ld, int R,
                                                         int col, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) out[i] = T[(size_t)i * ld + col];
}

// rhsd = dense RHS column, kept current by k_sens_update.
__global__ __launch_bounds__(1024) void k_sens_select(double* __restrict__ T, int ld, int R, int C,
                                                      double* __restrict__ rowbuf,
                                                      double* __restrict__ colbuf,
                                                      const double* __restrict__ rhsd,
                                                      int32_t* __restrict__ basic,
                                                      int32_t* __restrict__ bcount,
                                                      int32_t* __restrict__ log, SensState* st) {
    __shared__ int lds[32];
    __shared__ double lds_v[32];
    if (st->status != kRunning) return;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int rhs = C - 1;
    int phase = st->phase;
    const int it_d = st->iter_dual, it_p = st->iter_primal;
    __syncthreads();
    int leave = -1, enter = -1, kind = 0;
    constexpr int K = 16;

    if (phase == 0) {
        // `bi < mostNeg - EPS`, mostNeg = 0.0 at the start, rows ascending (:174-178)
        leave = eps_fold<K>(1, R, 0.0, [&](int i) { return rhsd[i]; }, lds, lds_v);
        if (leave == -1) {
            phase = 1;  // break (:180) -> ReOptimize
            if (tid == 0) st->phase = 1;
        } else {
            if (it_d > kSensMaxIter) {  // `if (iter++ > maxIter) throw` (:183)
                if (tid == 0) st->status = LPR_SENS_ITER_LIMIT;
                return;
            }
            // `a < -EPS: ratio = cbar / (-a); ratio < best - EPS`, columns ascending (:186-195)
            const double* __restrict__ lrow = T + (size_t)leave * ld;
            enter = eps_fold<K>(
                0, rhs, INFINITY,
                [&](int j) {
                    const double a = lrow[j];
                    return (a < -kSensEps) ? ieee_div(T[j], -a) : NAN;
                },
                lds, lds_v);
            if (enter == -1) {
                if (tid == 0) st->status = LPR_SENS_INFEASIBLE;  // :197
                return;
            }
            if (tid == 0) st->iter_dual = it_d + 1;
            for (int i = tid; i < R; i += nt) colbuf[i] = T[(size_t)i * ld + enter];
            __syncthreads();
        }
    }
    if (phase == 1) {
        kind = 1;
        // IsOptimal (:86-96) and the entering column `rc < mostNeg`, first index (:131-141)
        int notopt = 0;
        double bv = 0.0;
        int bj = INT_MAX;
        for (int j = tid; j < rhs; j += nt) {
            if (bcount[j] > 0) continue;
            const double rc = T[j];
            if (rc < -kSensEps) notopt = 1;
            if (rc < bv) {  // ascending j per thread: strict < keeps the first index
                bv = rc;
                bj = j;
            }
        }
        notopt = __syncthreads_or(notopt);
        if (!notopt) {
            if (tid == 0) st->status = LPR_SENS_OK;
            return;
        }
        if (it_p > kSensMaxIter) {  // `if (iter++ > maxIter) throw` (:126)
            if (tid == 0) st->status = LPR_SENS_ITER_LIMIT;
            return;
        }
        {  // block arg-min, smaller value first, then smaller index
            const int lane = tid & (kWave - 1), wave = tid / kWave, nw = nt / kWave;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_xor(bv, off, kWave);
                const int oj = __shfl_xor(bj, off, kWave);
                if (oj != INT_MAX && (bj == INT_MAX || ov < bv || (ov == bv && oj < bj))) {
                    bv = ov;
                    bj = oj;
                }
            }
            if (lane == 0) {
                lds_v[wave] = bv;
                lds[wave] = bj;
            }
            __syncthreads();
            bv = lds_v[0];
            bj = lds[0];
            for (int w = 1; w < nw; ++w) {
                const double ov = lds_v[w];
                const int oj = lds[w];
                if (oj != INT_MAX && (bj == INT_MAX || ov < bv || (ov == bv && oj < bj))) {
                    bv = ov;
                    bj = oj;
                }
            }
            __syncthreads();
        }
        if (bj == INT_MAX) {  // `if (enter == -1) break` (:142) -- falls out to the epilogue
            if (tid == 0) st->status = LPR_SENS_OK;
            return;
        }
        enter = bj;
        // one strided pass: the entering column becomes dense (it is also the factor column)
        for (int i = tid; i < R; i += nt) colbuf[i] = T[(size_t)i * ld + enter];
        __syncthreads();
        // `a > EPS: ratio = rhs / a; ratio < best - EPS`, rows ascending (:144-150)
        leave = eps_fold<K>(
            1, R, INFINITY,
            [&](int i) {
                const double a = colbuf[i];
                return (a > kSensEps) ? ieee_div(rhsd[i], a) : NAN;
            },
            lds, lds_v);
        if (leave == -1) {
            if (tid == 0) st->status = LPR_SENS_UNBOUNDED;  // :151
            return;
        }
        if (tid == 0) st->iter_primal = it_p + 1;
    }

    // Pivot (:98-119): stage the normalised row (the factor column is in colbuf already)
    const double piv = T[(size_t)leave * ld + enter];
    if (fabs(piv) < kSensEps) {
        if (tid == 0) st->status = LPR_SENS_ZERO_PIVOT;  // :101
        return;
    }
    for (int j = tid; j < ld; j += nt) rowbuf[j] = (j < C) ? ieee_div(T[(size_t)leave * ld + j], piv) : 0.0;
    __syncthreads();
    if (tid == 0) {
        st->pr = leave;
        st->pc = enter;
        st->sweep ^= 1;
        st->done += 1;
        const int old = basic[leave - 1];  // basicVars[leaveRow - 1] = enterCol (:117-118)
        if (old >= 0) bcount[old] -= 1;
        basic[leave - 1] = enter;
        bcount[enter] += 1;
        if (st->log_n < st->log_cap) {
            int32_t* e = log + 3 * st->log_n;
            e[0] = kind;
            e[1] = leave;
            e[2] = enter;
        }
        st->log_n += 1;
    }
}

// rows i != leaveRow with |factor| < EPS are left untouched (:110).  Tile = TR rows x 256 double2
// columns; all loads of a lane are issued before its first store, alternate pivots sweep the grid
// in opposite directions (the tail of one sweep is still in the Infinity Cache for the next).
template <int TR>
__global__ __launch_bounds__(256) void k_sens_update(double* __restrict__ T, int ld, int R, int C,
                                                     const double* __restrict__ rowbuf,
                                                     const double* __restrict__ colbuf,
                                                     double* __restrict__ rhsd,
                                                     const SensState* st) {
    if (st->status != kRunning) return;
    const int ld2 = ld >> 1;
    int ct = blockIdx.x, rt = blockIdx.y;
    if (st->sweep & 1) {
        ct = gridDim.x - 1 - ct;
        rt = gridDim.y - 1 - rt;
    }
    const int c2 = ct * 256 + threadIdx.x;
    const int i0 = rt * TR;
    const int r = st->pr;
    if (c2 >= ld2) return;
    const double2 pr2 = reinterpret_cast<const double2*>(rowbuf)[c2];
    double2* __restrict__ T2 = reinterpret_cast<double2*>(T);
    const int rhs = C - 1;
    const bool owns_rhs = (c2 == (rhs >> 1));
    double2 x[TR];
#pragma unroll
    for (int k = 0; k < TR; ++k) {
        const int i = i0 + k;
        if (i < R) x[k] = T2[(size_t)i * ld2 + c2];
    }
#pragma unroll
    for (int k = 0; k < TR; ++k) {
        const int i = i0 + k;
        if (i < R) {
            const double f = colbuf[i];
            const bool is_r = (i == r);
            if (is_r || !(fabs(f) < kSensEps)) {
                double2 o;
                const double px = f * pr2.x;
                const double py = f * pr2.y;
                o.x = x[k].x - px;
                o.y = x[k].y - py;
                if (is_r) o = pr2;
                T2[(size_t)i * ld2 + c2] = o;
                if (owns_rhs) rhsd[i] = (rhs & 1) ? o.y : o.x;
            }
        }
    }
}

// ---- edits ----------------------------------------------------------------------------------
// row 0 += delta * row r over all C columns (ChangeBasic :384-387)
__global__ __launch_bounds__(256) void k_sens_row_axpy(double* __restrict__ T, int ld, int C, int r,
                                                       double delta) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= C) return;
    const double prod = delta * T[(size_t)r * ld + j];
    T[j] = T[j] + prod;
}

// RHS column += delta * column sCol for rows 0..R-1 (ChangeRHS :445-450; row 0 is
// `+= ShadowPrices()[k-1] * delta`, the same product)
__global__ __launch_bounds__(256) void k_sens_rhs_axpy(double* __restrict__ T, int ld, int R, int C,
                                                       int sCol, double delta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const double prod = delta * T[(size_t)i * ld + sCol];
    T[(size_t)i * ld + (C - 1)] = T[(size_t)i * ld + (C - 1)] + prod;
}

// new tableau with one column inserted at position p (values colvals[i], or 0) -- AddNewActivity
// :553-570, AddNewConstraintNonInteractive :621-629.  The new buffer may have more rows.
__global__ __launch_bounds__(256) void k_sens_insert_col(const double* __restrict__ old, int ldo,
                                                         int R, int C, double* __restrict__ neu,
                                                         int ldn, int p,
                                                         const double* __restrict__ colvals) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= ldn || i >= R) return;
    double v = 0.0;
    if (j < p)
        v = old[(size_t)i * ldo + j];
    else if (j == p)
        v = colvals ? colvals[i] : 0.0;
    else if (j <= C)
        v = old[(size_t)i * ldo + (j - 1)];
    neu[(size_t)i * ldn + j] = v;
}

// out[j] = init[j] + sum_{i=0..m-1, in this order} w[i] * T[i+1][j], every product rounded before
// it is added (AddNewConstraintNonInteractive :636-645, PerformDuality :690-694).  One thread per
// column, loads of 8 rows in flight.
__global__ __launch_bounds__(64) void k_sens_colfold(const double* __restrict__ T, int ld, int m,
                                                     int ncols, const double* __restrict__ w,
                                                     const double* __restrict__ init,
                                                     double* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    double acc = init ? init[j] : 0.0;
    int i = 0;
    for (; i + 8 <= m; i += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = T[(size_t)(i + 1 + k) * ld + j];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double prod = w[i + k] * v[k];
            acc = acc + prod;
        }
    }
    for (; i < m; ++i) {
        const double prod = w[i] * T[(size_t)(i + 1) * ld + j];
        acc = acc + prod;
    }
    out[j] = acc;
}

}  // namespace lpr

// ---------------------------------------------------------------------------------------------
// host side

using namespace lpr;

namespace {

void sens_free_shape(lpr_sens* s) {
    hipFree(s->T); hipFree(s->rowbuf); hipFree(s->colbuf); hipFree(s->sol); hipFree(s->fold_w);
    hipFree(s->fold_io); hipFree(s->basic); hipFree(s->bcount); hipFree(s->cnt);
    hipFree(s->rowsum); hipFree(s->cand); hipFree(s->rhsd); hipFree(s->snapT); hipFree(s->snapI);
    s->T = s->rowbuf = s->colbuf = s->sol = s->fold_w = s->fold_io = s->rhsd = s->snapT = nullptr;
    s->basic = s->bcount = s->cnt = s->rowsum = s->cand = s->snapI = nullptr;
}

void sens_release_device(lpr_sens* s) {
    hipSetDevice(s->eng->device);
    if (s->eng->stream) hipStreamSynchronize(s->eng->stream);
    sens_free_shape(s);
    hipFree(s->log);
    hipFree(s->state);
    if (s->h_state) hipHostFree(s->h_state);
    s->log = nullptr;
    s->state = s->h_state = nullptr;
}

// allocate everything that depends on the shape except T (zeroed); on failure nothing is kept
int sens_alloc_aux(lpr_sens* s, int R, int ld, double** T_out) {
    hipError_t err = hipSuccess;
    auto chk = [&](hipError_t x) { if (err == hipSuccess) err = x; };
    const size_t D = sizeof(double), I = sizeof(int32_t);
    const int rp = align_up(R + 1, 16);
    double *T = nullptr, *rowbuf = nullptr, *colbuf = nullptr, *sol = nullptr, *fw = nullptr,
           *fio = nullptr, *rhsd = nullptr;
    int32_t *basic = nullptr, *bcount = nullptr, *cnt = nullptr, *rowsum = nullptr,
            *cand = nullptr;
    chk(hipMalloc(&T, (size_t)R * ld * D));
    chk(hipMalloc(&rowbuf, (size_t)ld * D));
    chk(hipMalloc(&colbuf, (size_t)rp * D));
    chk(hipMalloc(&sol, (size_t)ld * D));
    chk(hipMalloc(&fw, (size_t)rp * D));
    chk(hipMalloc(&fio, (size_t)std::max(ld, rp) * D));
    chk(hipMalloc(&rhsd, (size_t)rp * D));
    chk(hipMalloc(&basic, (size_t)rp * I));
    chk(hipMalloc(&bcount, (size_t)ld * I));
    chk(hipMalloc(&cnt, (size_t)ld * I));
    chk(hipMalloc(&rowsum, (size_t)ld * I));
    chk(hipMalloc(&cand, (size_t)ld * I));
    if (err != hipSuccess) {
        set_error("device allocation for a sensitivity analyzer (%d x %d) failed: %s", R, ld,
                  hipGetErrorString(err));
        hipFree(T); hipFree(rowbuf); hipFree(colbuf); hipFree(sol); hipFree(fw); hipFree(fio);
        hipFree(rhsd); hipFree(basic); hipFree(bcount); hipFree(cnt); hipFree(rowsum); hipFree(cand);
        return err == hipErrorOutOfMemory ? LPR_OUT_OF_MEMORY : LPR_DEVICE_ERROR;
    }
    hipStream_t st = s->eng->stream;
    LPR_HIP(hipMemsetAsync(T, 0, (size_t)R * ld * D, st));
    LPR_HIP(hipMemsetAsync(bcount, 0, (size_t)ld * I, st));
    LPR_HIP(hipMemsetAsync(basic, 0xff, (size_t)rp * I, st));
    LPR_HIP(hipMemsetAsync(sol, 0, (size_t)ld * D, st));
    // the old aux arrays go, the old T stays with the caller until it has been copied from
    hipFree(s->rowbuf); hipFree(s->colbuf); hipFree(s->sol); hipFree(s->fold_w);
    hipFree(s->fold_io); hipFree(s->basic); hipFree(s->bcount); hipFree(s->cnt);
    hipFree(s->rowsum); hipFree(s->cand); hipFree(s->rhsd);
    hipFree(s->snapT); hipFree(s->snapI);  // a snapshot of the old shape is of no use
    s->snapT = nullptr;
    s->snapI = nullptr;
    s->rhsd = rhsd;
    s->rowbuf = rowbuf; s->colbuf = colbuf; s->sol = sol; s->fold_w = fw; s->fold_io = fio;
    s->basic = basic; s->bcount = bcount; s->cnt = cnt; s->rowsum = rowsum; s->cand = cand;
    *T_out = T;
    return LPR_OK_OPTIMAL;
}

int sens_colscan(lpr_sens* s) {
    hipStream_t st = s->eng->stream;
    const int nc = s->C - 1;
    LPR_HIP(hipMemsetAsync(s->cnt, 0, (size_t)s->ld * sizeof(int32_t), st));
    LPR_HIP(hipMemsetAsync(s->rowsum, 0, (size_t)s->ld * sizeof(int32_t), st));
    if (nc <= 0 || s->R < 2) return LPR_OK_OPTIMAL;
    const int rpg = 64;
    dim3 grid((nc + 255) / 256, (s->R - 1 + rpg - 1) / rpg);
    hipLaunchKernelGGL(k_sens_colscan, grid, dim3(256), 0, st, s->T, s->ld, s->R, nc, rpg, s->cnt,
                       s->rowsum);
    LPR_HIP(hipGetLastError());
    return LPR_OK_OPTIMAL;
}

int sens_fetch_basic(lpr_sens* s) {
    const int m = s->R - 1;
    s->h_basic.assign(m, -1);
    if (m > 0)
        LPR_HIP(hipMemcpyAsync(s->h_basic.data(), s->basic, (size_t)m * sizeof(int32_t),
                               hipMemcpyDeviceToHost, s->eng->stream));
    LPR_HIP(hipStreamSynchronize(s->eng->stream));
    return LPR_OK_OPTIMAL;
}

// RebuildBasicsFromTableau (:706-723)
int sens_rebuild(lpr_sens* s) {
    hipStream_t st = s->eng->stream;
    const int m = s->R - 1, nc = s->C - 1;
    int rc = sens_colscan(s);
    if (rc != LPR_OK_OPTIMAL) return rc;
    LPR_HIP(hipMemsetAsync(s->basic, 0x7f, (size_t)std::max(m, 1) * sizeof(int32_t), st));
    LPR_HIP(hipMemsetAsync(s->bcount, 0, (size_t)s->ld * sizeof(int32_t), st));
    if (nc > 0)
        hipLaunchKernelGGL(k_sens_cand, dim3((nc + 255) / 256), dim3(256), 0, st, s->T, s->ld, s->C,
                           s->cnt, s->rowsum, s->cand, s->basic, (double*)nullptr);
    if (m > 0)
        hipLaunchKernelGGL(k_sens_basics_finish, dim3((m + 255) / 256), dim3(256), 0, st, s->basic,
                           m, s->bcount);
    LPR_HIP(hipGetLastError());
    return LPR_OK_OPTIMAL;
}

bool basic_contains(const lpr_sens* s, int j) {
    return std::find(s->h_basic.begin(), s->h_basic.end(), j) != s->h_basic.end();
}

int read_elem(lpr_sens* s, int i, int j, double* v) {
    LPR_HIP(hipMemcpyAsync(v, s->T + (size_t)i * s->ld + j, sizeof(double), hipMemcpyDeviceToHost,
                           s->eng->stream));
    LPR_HIP(hipStreamSynchronize(s->eng->stream));
    return LPR_OK_OPTIMAL;
}

int write_elem(lpr_sens* s, int i, int j, double v) {
    LPR_HIP(hipMemcpyAsync(s->T + (size_t)i * s->ld + j, &v, sizeof(double), hipMemcpyHostToDevice,
                           s->eng->stream));
    LPR_HIP(hipStreamSynchronize(s->eng->stream));  // &v is a stack slot
    return LPR_OK_OPTIMAL;
}

// DualSimplexIfNeeded + ReOptimize (with RebuildBasicsFromTableau first: ResolveAll).  The outcome
// is an lpr_sens_outcome; an API failure is returned as a negative lpr_status.
int sens_run(lpr_sens* s, bool rebuild, int* outcome) {
    hipStream_t st = s->eng->stream;
    if (rebuild) {
        int rc = sens_rebuild(s);
        if (rc != LPR_OK_OPTIMAL) return rc;
    }
    SensState* hs = static_cast<SensState*>(s->h_state);
    SensState* ds = static_cast<SensState*>(s->state);
    std::memset(hs, 0, sizeof(SensState));
    hs->status = kRunning;
    hs->log_n = s->log_n;
    hs->log_cap = s->log_cap;
    LPR_HIP(hipMemcpyAsync(ds, hs, sizeof(SensState), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_sens_gather_col, dim3((s->R + 255) / 256), dim3(256), 0, st, s->T, s->ld,
                       s->R, s->C - 1, s->rhsd);
    constexpr int TR = 32;
    const dim3 ugrid((s->ld / 2 + 255) / 256, (s->R + TR - 1) / TR);
    const int batch = 4;
    for (;;) {
        for (int k = 0; k < batch; ++k) {
            hipLaunchKernelGGL(k_sens_select, dim3(1), dim3(1024), 0, st, s->T, s->ld, s->R, s->C,
                               s->rowbuf, s->colbuf, s->rhsd, s->basic, s->bcount, s->log, ds);
            hipLaunchKernelGGL((k_sens_update<TR>), ugrid, dim3(256), 0, st, s->T, s->ld, s->R,
                               s->C, s->rowbuf, s->colbuf, s->rhsd, ds);
        }
        LPR_HIP(hipGetLastError());
        LPR_HIP(hipMemcpyAsync(hs, ds, sizeof(SensState), hipMemcpyDeviceToHost, st));
        LPR_HIP(hipStreamSynchronize(st));
        if (hs->status != kRunning) break;
    }
    s->log_n = hs->log_n;
    s->pivots = hs->done;
    *outcome = hs->status;
    if (hs->status == LPR_SENS_OK) {  // ReOptimize's epilogue (:159-165)
        int rc = sens_colscan(s);
        if (rc != LPR_OK_OPTIMAL) return rc;
        const int nc = s->C - 1;
        if (nc > 0)
            hipLaunchKernelGGL(k_sens_cand, dim3((nc + 255) / 256), dim3(256), 0, st, s->T, s->ld,
                               s->C, s->cnt, s->rowsum, s->cand, (int32_t*)nullptr, s->sol);
        LPR_HIP(hipGetLastError());
        s->h_sol.assign(nc, 0.0);
        if (nc > 0)
            LPR_HIP(hipMemcpyAsync(s->h_sol.data(), s->sol, (size_t)nc * sizeof(double),
                                   hipMemcpyDeviceToHost, st));
        LPR_HIP(hipMemcpyAsync(&s->z, s->T + (s->C - 1), sizeof(double), hipMemcpyDeviceToHost,
                               st));
    }
    return sens_fetch_basic(s);
}

int sens_new(lpr_engine* e, int rows, int cols, lpr_sens** out) {
    if (!e || !out || rows < 2 || cols < rows || rows > 65535) {
        // cols < rows would make n = cols - m - 1 negative: the C# indexes tableau[0, n + i - 1]
        set_error("lpr_sens: bad tableau shape %d x %d", rows, cols);
        return LPR_BAD_ARGUMENT;
    }
    LPR_HIP(hipSetDevice(e->device));
    lpr_sens* s = new (std::nothrow) lpr_sens();
    if (!s) return LPR_OUT_OF_MEMORY;
    s->eng = e;
    s->R = rows;
    s->C = cols;
    s->ld = align_up(cols, kLdAlign);
    s->log_cap = 1 << 16;
    hipError_t err = hipMalloc(&s->log, (size_t)s->log_cap * 3 * sizeof(int32_t));
    if (err == hipSuccess) err = hipMalloc(&s->state, sizeof(SensState));
    if (err == hipSuccess) err = hipHostMalloc(&s->h_state, sizeof(SensState));
    int rc = LPR_OK_OPTIMAL;
    if (err != hipSuccess) {
        set_error("sensitivity analyzer allocation failed: %s", hipGetErrorString(err));
        rc = LPR_DEVICE_ERROR;
    } else {
        rc = sens_alloc_aux(s, rows, s->ld, &s->T);
    }
    if (rc != LPR_OK_OPTIMAL) {
        sens_release_device(s);
        delete s;
        return rc;
    }
    e->live_sens.push_back(s);
    *out = s;
    return LPR_OK_OPTIMAL;
}

// replace the tableau by `nT` (nR x nC, already filled) -- aux arrays were re-made by the caller
void sens_adopt(lpr_sens* s, double* nT, int nR, int nC, int nld) {
    hipFree(s->T);
    s->T = nT;
    s->R = nR;
    s->C = nC;
    s->ld = nld;
}

}  // namespace

namespace lpr {
void sens_orphan(lpr_sens* s) {  // lpr_engine_close
    sens_release_device(s);
    s->eng = nullptr;
}
}  // namespace lpr

#define LPR_LIVE_S(s)                                                                       \
    do {                                                                                    \
        if (!(s) || !(s)->eng) {                                                            \
            set_error("sensitivity handle is null or its engine has been closed");          \
            return LPR_BAD_ARGUMENT;                                                        \
        }                                                                                   \
        LPR_HIP(hipSetDevice((s)->eng->device));                                            \
    } while (0)

extern "C" {

int lpr_sens_create(lpr_engine* e, const double* final_tableau, int32_t rows, int32_t cols,
                    const double* solution, int32_t nsol, double z, lpr_sens** out) {
    if (!final_tableau || nsol < 0 || (nsol > 0 && !solution)) {
        set_error("lpr_sens_create: null tableau / solution");
        return LPR_BAD_ARGUMENT;
    }
    lpr_sens* s = nullptr;
    int rc = sens_new(e, rows, cols, &s);
    if (rc != LPR_OK_OPTIMAL) return rc;
    hipStream_t st = e->stream;
    auto fail = [&](int code) {
        lpr_sens_destroy(s);
        return code;
    };
    if (hipMemcpy2DAsync(s->T, (size_t)s->ld * sizeof(double), final_tableau,
                         (size_t)cols * sizeof(double), (size_t)cols * sizeof(double), rows,
                         hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        set_error("lpr_sens_create: tableau upload failed");
        return fail(LPR_DEVICE_ERROR);
    }
    s->h_sol.assign(solution, solution + nsol);
    s->z = z;
    rc = write_elem(s, 0, cols - 1, z);  // tableau[0, numCols - 1] = finalZ (:32)
    if (rc == LPR_OK_OPTIMAL) rc = sens_rebuild(s);  // :35
    if (rc == LPR_OK_OPTIMAL) rc = sens_fetch_basic(s);
    if (rc != LPR_OK_OPTIMAL) return fail(rc);
    *out = s;
    return LPR_OK_OPTIMAL;
}

int lpr_sens_create_from_tableau(lpr_tableau* t, int32_t n_decision, lpr_sens** out) {
    if (!t || !t->eng || !out || n_decision < 0 || n_decision > t->cols - 1) {
        set_error("lpr_sens_create_from_tableau: bad arguments");
        return LPR_BAD_ARGUMENT;
    }
    lpr_sens* s = nullptr;
    int rc = sens_new(t->eng, t->rows, t->cols, &s);
    if (rc != LPR_OK_OPTIMAL) return rc;
    auto fail = [&](int code) {
        lpr_sens_destroy(s);
        return code;
    };
    hipStream_t st = t->eng->stream;
    // primalSolver.GetFinalTableau() / SolutionVector / FinalZ (Program.cs:147-151), device to device
    if (hipMemcpyAsync(s->T, t->T, (size_t)t->rows * t->ld * sizeof(double),
                       hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("lpr_sens_create_from_tableau: copy failed");
        return fail(LPR_DEVICE_ERROR);
    }
    s->h_sol.assign(n_decision, 0.0);
    rc = lpr_extract_solution(t, n_decision, s->h_sol.data(), &s->z);
    if (rc == LPR_OK_OPTIMAL) rc = sens_rebuild(s);
    if (rc == LPR_OK_OPTIMAL) rc = sens_fetch_basic(s);
    if (rc != LPR_OK_OPTIMAL) return fail(rc);
    *out = s;
    return LPR_OK_OPTIMAL;
}

int lpr_sens_destroy(lpr_sens* s) {
    if (!s) return LPR_BAD_ARGUMENT;
    if (s->eng) {
        sens_release_device(s);
        auto& lv = s->eng->live_sens;
        for (size_t k = 0; k < lv.size(); ++k)
            if (lv[k] == s) {
                lv.erase(lv.begin() + k);
                break;
            }
    }
    delete s;
    return LPR_OK_OPTIMAL;
}

int lpr_sens_shape(lpr_sens* s, int32_t* rows, int32_t* cols, int32_t* nsol, int32_t* nbasic,
                   double* z, int64_t* last_pivots) {
    if (!s) return LPR_BAD_ARGUMENT;
    if (rows) *rows = s->R;
    if (cols) *cols = s->C;
    if (nsol) *nsol = (int32_t)s->h_sol.size();
    if (nbasic) *nbasic = (int32_t)s->h_basic.size();
    if (z) *z = s->z;
    if (last_pivots) *last_pivots = s->pivots;
    return LPR_OK_OPTIMAL;
}

int lpr_sens_read(lpr_sens* s, double* tableau, int32_t* basic, double* solution) {
    LPR_LIVE_S(s);
    if (tableau) {
        LPR_HIP(hipMemcpy2DAsync(tableau, (size_t)s->C * sizeof(double), s->T,
                                 (size_t)s->ld * sizeof(double), (size_t)s->C * sizeof(double),
                                 s->R, hipMemcpyDeviceToHost, s->eng->stream));
        LPR_HIP(hipStreamSynchronize(s->eng->stream));
    }
    if (basic) std::copy(s->h_basic.begin(), s->h_basic.end(), basic);
    if (solution) std::copy(s->h_sol.begin(), s->h_sol.end(), solution);
    return LPR_OK_OPTIMAL;
}

int lpr_sens_read_block(lpr_sens* s, int32_t row0, int32_t nrows, int32_t col0, int32_t ncols,
                        double* out) {
    LPR_LIVE_S(s);
    if (!out || row0 < 0 || col0 < 0 || nrows < 0 || ncols < 0 || row0 + nrows > s->R ||
        col0 + ncols > s->C) {
        set_error("lpr_sens_read_block: block out of range");
        return LPR_BAD_ARGUMENT;
    }
    if (nrows == 0 || ncols == 0) return LPR_OK_OPTIMAL;
    LPR_HIP(hipMemcpy2DAsync(out, (size_t)ncols * sizeof(double),
                             s->T + (size_t)row0 * s->ld + col0, (size_t)s->ld * sizeof(double),
                             (size_t)ncols * sizeof(double), nrows, hipMemcpyDeviceToHost,
                             s->eng->stream));
    LPR_HIP(hipStreamSynchronize(s->eng->stream));
    return LPR_OK_OPTIMAL;
}

int lpr_sens_basic_row(lpr_sens* s, int32_t col, int32_t* row) {
    LPR_LIVE_S(s);
    if (!row || col < 0 || col >= s->C) {
        set_error("lpr_sens_basic_row: column out of range");
        return LPR_BAD_ARGUMENT;
    }
    hipStream_t st = s->eng->stream;
    hipLaunchKernelGGL(k_sens_basic_row, dim3(1), dim3(1024), 0, st, s->T, s->ld, s->R, col,
                       s->cand);
    LPR_HIP(hipGetLastError());
    LPR_HIP(hipMemcpyAsync(row, s->cand, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    LPR_HIP(hipStreamSynchronize(st));
    return LPR_OK_OPTIMAL;
}

int lpr_sens_log_read(lpr_sens* s, int32_t* triples, int64_t cap, int64_t* count) {
    LPR_LIVE_S(s);
    if (!count || cap < 0) return LPR_BAD_ARGUMENT;
    int64_t k = std::min(std::min(s->log_n, s->log_cap), cap);
    *count = s->log_n;
    if (k == 0 || !triples) return LPR_OK_OPTIMAL;
    LPR_HIP(hipMemcpy(triples, s->log, (size_t)k * 3 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return LPR_OK_OPTIMAL;
}

int lpr_sens_column_fold(lpr_sens* s, const double* w, int32_t nw, const double* init,
                         int32_t ncols, double* out) {
    LPR_LIVE_S(s);
    if (!w || !out || nw != s->R - 1 || ncols < 0 || ncols > s->C) {
        set_error("lpr_sens_column_fold: need %d weights and at most %d columns", s->R - 1, s->C);
        return LPR_BAD_ARGUMENT;
    }
    if (ncols == 0) return LPR_OK_OPTIMAL;
    hipStream_t st = s->eng->stream;
    LPR_HIP(hipMemcpyAsync(s->fold_w, w, (size_t)nw * sizeof(double), hipMemcpyHostToDevice, st));
    if (init)
        LPR_HIP(hipMemcpyAsync(s->fold_io, init, (size_t)ncols * sizeof(double),
                               hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_sens_colfold, dim3((ncols + 63) / 64), dim3(64), 0, st, s->T, s->ld, nw,
                       ncols, s->fold_w, init ? s->fold_io : (const double*)nullptr, s->fold_io);
    LPR_HIP(hipGetLastError());
    LPR_HIP(hipMemcpyAsync(out, s->fold_io, (size_t)ncols * sizeof(double), hipMemcpyDeviceToHost,
                           st));
    LPR_HIP(hipStreamSynchronize(st));
    return LPR_OK_OPTIMAL;
}

int lpr_sens_resolve_all(lpr_sens* s, int32_t* outcome) {
    LPR_LIVE_S(s);
    if (!outcome) return LPR_BAD_ARGUMENT;
    int oc = 0;
    int rc = sens_run(s, true, &oc);
    if (rc != LPR_OK_OPTIMAL) return rc;
    *outcome = oc;
    return LPR_OK_OPTIMAL;
}

int lpr_sens_change_nonbasic_cbar(lpr_sens* s, int32_t index, double new_cbar, int32_t* outcome) {
    LPR_LIVE_S(s);
    if (!outcome) return LPR_BAD_ARGUMENT;
    s->pivots = 0;
    if (index < 0 || index >= s->C - 1 || basic_contains(s, index)) {  // :306-310
        *outcome = LPR_SENS_INVALID_INDEX;
        return LPR_OK_OPTIMAL;
    }
    int rc = write_elem(s, 0, index, new_cbar);  // :318
    if (rc != LPR_OK_OPTIMAL) return rc;
    return lpr_sens_resolve_all(s, outcome);
}

int lpr_sens_change_basic(lpr_sens* s, int32_t col, double delta, int32_t* outcome) {
    LPR_LIVE_S(s);
    if (!outcome) return LPR_BAD_ARGUMENT;
    s->pivots = 0;
    if (col < 0 || col >= s->C - 1 || !basic_contains(s, col)) {  // :368-372
        *outcome = LPR_SENS_INVALID_INDEX;
        return LPR_OK_OPTIMAL;
    }
    int32_t r = -1;
    int rc = lpr_sens_basic_row(s, col, &r);
    if (rc != LPR_OK_OPTIMAL) return rc;
    if (r < 0) {  // "Could not locate basic row." :379
        *outcome = LPR_SENS_INVALID_INDEX;
        return LPR_OK_OPTIMAL;
    }
    hipStream_t st = s->eng->stream;
    hipLaunchKernelGGL(k_sens_row_axpy, dim3((s->C + 255) / 256), dim3(256), 0, st, s->T, s->ld,
                       s->C, r, delta);  // :384-387
    LPR_HIP(hipGetLastError());
    rc = read_elem(s, 0, s->C - 1, &s->z);  // finalZ = tableau[0, numCols-1] :388
    if (rc != LPR_OK_OPTIMAL) return rc;
    return lpr_sens_resolve_all(s, outcome);
}

int lpr_sens_change_rhs(lpr_sens* s, int32_t k, double new_b, int32_t* outcome) {
    LPR_LIVE_S(s);
    if (!outcome) return LPR_BAD_ARGUMENT;
    s->pivots = 0;
    if (k < 1 || k >= s->R) {  // :430-431
        *outcome = LPR_SENS_INVALID_INDEX;
        return LPR_OK_OPTIMAL;
    }
    hipStream_t st = s->eng->stream;
    // snapshot (:437-439): tableau, finalZ, basicVars (+ the membership counts that shadow it)
    const size_t tb = (size_t)s->R * s->ld * sizeof(double);
    const int m = s->R - 1;
    const size_t ib = (size_t)(m + s->ld) * sizeof(int32_t);
    if (!s->snapT) {
        if (hipMalloc(&s->snapT, tb) != hipSuccess || hipMalloc(&s->snapI, ib) != hipSuccess) {
            hipFree(s->snapT);
            s->snapT = nullptr;
            s->snapI = nullptr;
            set_error("lpr_sens_change_rhs: snapshot allocation failed");
            return LPR_OUT_OF_MEMORY;
        }
    }
    double* snapT = s->snapT;
    int32_t* snapI = s->snapI;
    auto done = [&](int code) {
        hipStreamSynchronize(st);
        return code;
    };
    if (hipMemcpyAsync(snapT, s->T, tb, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(snapI, s->basic, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToDevice,
                       st) != hipSuccess ||
        hipMemcpyAsync(snapI + m, s->bcount, (size_t)s->ld * sizeof(int32_t),
                       hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("lpr_sens_change_rhs: snapshot copy failed");
        return done(LPR_DEVICE_ERROR);
    }
    const double oldZ = s->z;
    double oldB = 0.0;
    int rc = read_elem(s, k, s->C - 1, &oldB);
    if (rc != LPR_OK_OPTIMAL) return done(rc);
    const double delta = new_b - oldB;  // :441-442
    const int n = s->C - m - 1;
    const int sCol = n + (k - 1);
    hipLaunchKernelGGL(k_sens_rhs_axpy, dim3((s->R + 255) / 256), dim3(256), 0, st, s->T, s->ld,
                       s->R, s->C, sCol, delta);  // :445-450
    if (hipGetLastError() != hipSuccess) return done(LPR_DEVICE_ERROR);
    rc = read_elem(s, 0, s->C - 1, &s->z);  // :451
    if (rc != LPR_OK_OPTIMAL) return done(rc);
    int oc = 0;
    rc = sens_run(s, false, &oc);  // DualSimplexIfNeeded(); ReOptimize(); :455-456
    if (rc != LPR_OK_OPTIMAL) return done(rc);
    if (oc != LPR_SENS_OK) {  // catch: restore (:462-469); solutionVector is not restored
        if (hipMemcpyAsync(s->T, snapT, tb, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(s->basic, snapI, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToDevice,
                           st) != hipSuccess ||
            hipMemcpyAsync(s->bcount, snapI + m, (size_t)s->ld * sizeof(int32_t),
                           hipMemcpyDeviceToDevice, st) != hipSuccess) {
            set_error("lpr_sens_change_rhs: rollback copy failed");
            return done(LPR_DEVICE_ERROR);
        }
        s->z = oldZ;
        rc = sens_fetch_basic(s);
        if (rc != LPR_OK_OPTIMAL) return done(rc);
        oc = LPR_SENS_ROLLED_BACK;
    }
    *outcome = oc;
    return done(LPR_OK_OPTIMAL);
}

int lpr_sens_change_nonbasic_column(lpr_sens* s, int32_t row, int32_t col, double new_val,
                                    int32_t* outcome) {
    LPR_LIVE_S(s);
    if (!outcome) return LPR_BAD_ARGUMENT;
    s->pivots = 0;
    if (row < 1 || row >= s->R || col < 0 || col >= s->C - 1 || basic_contains(s, col)) {
        *outcome = LPR_SENS_INVALID_INDEX;  // :505-516
        return LPR_OK_OPTIMAL;
    }
    const int m = s->R - 1, n = s->C - m - 1;
    double oldVal = 0.0, yi = 0.0, cbar = 0.0;
    int rc = read_elem(s, row, col, &oldVal);
    if (rc == LPR_OK_OPTIMAL) rc = write_elem(s, row, col, new_val);  // :523
    // ShadowPrices()[row-1] is read AFTER the entry was changed (:525); it is a row-0 entry, so
    // the order only matters in that the write above never touches row 0 (row >= 1)
    if (rc == LPR_OK_OPTIMAL) rc = read_elem(s, 0, n + (row - 1), &yi);
    if (rc == LPR_OK_OPTIMAL) rc = read_elem(s, 0, col, &cbar);
    if (rc != LPR_OK_OPTIMAL) return rc;
    const double delta = new_val - oldVal;
    const double prod = yi * delta;
    rc = write_elem(s, 0, col, cbar + prod);  // :526
    if (rc != LPR_OK_OPTIMAL) return rc;
    return lpr_sens_resolve_all(s, outcome);
}

int lpr_sens_add_activity(lpr_sens* s, double c_new, const double* a_new, int32_t na,
                          int32_t* outcome) {
    LPR_LIVE_S(s);
    const int m = s->R - 1, n = s->C - m - 1;
    if (!outcome || !a_new || na != m) {
        set_error("lpr_sens_add_activity: need %d column entries", m);
        return LPR_BAD_ARGUMENT;
    }
    s->pivots = 0;
    hipStream_t st = s->eng->stream;
    // c̄_new = y^T a_new - c_new, summed in index order (:543-551)
    std::vector<double> y(m), colv(s->R);
    if (m > 0) {
        LPR_HIP(hipMemcpyAsync(y.data(), s->T + n, (size_t)m * sizeof(double),
                               hipMemcpyDeviceToHost, st));
        LPR_HIP(hipStreamSynchronize(st));
    }
    double yTa = 0.0;
    for (int i = 0; i < m; ++i) {
        const double prod = y[i] * a_new[i];
        yTa = yTa + prod;
    }
    colv[0] = yTa - c_new;
    for (int i = 0; i < m; ++i) colv[i + 1] = a_new[i];
    // new tableau with the column inserted before the slacks (:553-570)
    const int C2 = s->C + 1, ld2 = align_up(C2, kLdAlign);
    double* oldT = s->T;
    const int oldld = s->ld, oldC = s->C;
    double* nT = nullptr;
    int rc = sens_alloc_aux(s, s->R, ld2, &nT);
    if (rc != LPR_OK_OPTIMAL) return rc;
    LPR_HIP(hipMemcpyAsync(s->fold_io, colv.data(), (size_t)s->R * sizeof(double),
                           hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_sens_insert_col, dim3((ld2 + 255) / 256, s->R), dim3(256), 0, st, oldT,
                       oldld, s->R, oldC, nT, ld2, n, s->fold_io);
    LPR_HIP(hipGetLastError());
    LPR_HIP(hipStreamSynchronize(st));  // colv is read by the copy above
    s->T = oldT;
    sens_adopt(s, nT, s->R, C2, ld2);
    // basicVars[i]++ for indices >= n (:575-577) is overwritten by ResolveAll's rebuild (:581)
    return lpr_sens_resolve_all(s, outcome);
}

int lpr_sens_add_constraint(lpr_sens* s, const double* tech, int32_t ntech, double rhs,
                            int32_t* outcome) {
    LPR_LIVE_S(s);
    if (!outcome || (ntech > 0 && !tech)) return LPR_BAD_ARGUMENT;
    s->pivots = 0;
    const int oldM = s->R - 1, oldNM = s->C - 1;
    if (ntech != oldNM) {  // :616-617 ArgumentException
        *outcome = LPR_SENS_INVALID_INDEX;
        return LPR_OK_OPTIMAL;
    }
    // coeff_j = -tech[j] + sum_pos tech[basicVars[pos]] * tableau[pos+1, j] (:636-645).  A row
    // without a basic column has basicVars[pos] = -1: tech[-1] throws in the C#.
    std::vector<double> w(oldM), init(oldNM);
    for (int pos = 0; pos < oldM; ++pos) {
        const int bc = s->h_basic[pos];
        if (oldNM > 0 && (bc < 0 || bc >= ntech)) {
            *outcome = LPR_SENS_INDEX_OUT_OF_RANGE;
            return LPR_OK_OPTIMAL;
        }
        w[pos] = tech[bc];
    }
    for (int j = 0; j < oldNM; ++j) init[j] = -tech[j];
    double aX = 0.0;  // :647-651
    const int lim = std::min<int>(ntech, (int)s->h_sol.size());
    for (int j = 0; j < lim; ++j) {
        const double prod = tech[j] * s->h_sol[j];
        aX = aX + prod;
    }
    hipStream_t st = s->eng->stream;
    const int R2 = s->R + 1, C2 = s->C + 1, ld2 = align_up(C2, kLdAlign);
    // the fold runs on the OLD tableau, into a scratch row of its own
    double* d_w = nullptr;
    double* d_row = nullptr;
    if (hipMalloc(&d_w, (size_t)std::max(oldM, 1) * sizeof(double)) != hipSuccess ||
        hipMalloc(&d_row, (size_t)ld2 * sizeof(double)) != hipSuccess) {
        hipFree(d_w);
        set_error("lpr_sens_add_constraint: scratch allocation failed");
        return LPR_OUT_OF_MEMORY;
    }
    auto done = [&](int code) {
        hipStreamSynchronize(st);
        hipFree(d_w);
        hipFree(d_row);
        return code;
    };
    bool ok = hipMemsetAsync(d_row, 0, (size_t)ld2 * sizeof(double), st) == hipSuccess;
    if (ok && oldM > 0)
        ok = hipMemcpyAsync(d_w, w.data(), (size_t)oldM * sizeof(double), hipMemcpyHostToDevice,
                            st) == hipSuccess;
    if (ok && oldNM > 0)
        ok = hipMemcpyAsync(d_row, init.data(), (size_t)oldNM * sizeof(double),
                            hipMemcpyHostToDevice, st) == hipSuccess;
    if (!ok) {
        set_error("lpr_sens_add_constraint: upload failed");
        return done(LPR_DEVICE_ERROR);
    }
    if (oldNM > 0)
        hipLaunchKernelGGL(k_sens_colfold, dim3((oldNM + 63) / 64), dim3(64), 0, st, s->T, s->ld,
                           oldM, oldNM, d_w, d_row, d_row);
    const double tail[2] = {1.0, rhs - aX};  // new slack, new RHS (:652-653)
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(d_row + oldNM, tail, sizeof(tail), hipMemcpyHostToDevice, st) !=
            hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        set_error("lpr_sens_add_constraint: new row failed");
        return done(LPR_DEVICE_ERROR);
    }
    double* oldT = s->T;
    const int oldld = s->ld, oldR = s->R, oldC = s->C;
    double* nT = nullptr;
    int rc = sens_alloc_aux(s, R2, ld2, &nT);
    if (rc != LPR_OK_OPTIMAL) return done(rc);
    // old rows with a zero column for the new slack before the RHS (:621-631), then the new row
    hipLaunchKernelGGL(k_sens_insert_col, dim3((ld2 + 255) / 256, oldR), dim3(256), 0, st, oldT,
                       oldld, oldR, oldC, nT, ld2, oldC - 1, (const double*)nullptr);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(nT + (size_t)oldR * ld2, d_row, (size_t)ld2 * sizeof(double),
                       hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        hipFree(nT);
        set_error("lpr_sens_add_constraint: tableau rebuild failed");
        return done(LPR_DEVICE_ERROR);
    }
    s->T = oldT;
    sens_adopt(s, nT, R2, C2, ld2);
    // basicVars.Add(newSlackCol) (:656) is overwritten by ResolveAll's rebuild (:658)
    rc = lpr_sens_resolve_all(s, outcome);
    return done(rc);
}

}  // extern "C"
