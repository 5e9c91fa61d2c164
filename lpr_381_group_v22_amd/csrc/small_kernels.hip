// small_kernels.hip -- primal simplex pivots on CACHE-RESIDENT tableaux (R <= 1024 rows, ld <= 2048
// columns: BASELINE configs[1], m = 512, n = 1024, 6.3 MB; every Branch & Bound root of that size).
// (reference: LPR_381_Group_V22/Simplex/PrimalSimplexSolver.cs:102-211)
//
// At these sizes a pivot costs latency, not bytes: the K-pivots-per-sweep scheme of
// overlap_kernels.hip decides a pivot in ~6 dependent round trips, two of them hand-offs between
// the workgroups that share the row and column work (6.5 us per pivot at m = 512).  Here ONE
// workgroup of 512 lanes decides all K = 16 pivots of a block: a lane owns two rows (column gather,
// ratio test, right-hand side) and two pairs of columns (pivot row, Z row), the block's factor
// columns live in LDS, its pivot rows in the lanes' registers, and what the other kernels hand
// through L2 goes through LDS and a workgroup barrier.  Per pivot: two gathers from the tableau in
// L2 / Infinity Cache (column e, row r), two (value, index) arg-min reductions, seven barriers.
// Then one in-place sweep applies the block to every element (k_small_sweep), each element going
// through the same K rounded multiply / rounded subtract steps, in the same order, as K separate
// C# pivots would put it through -- the stored bits are identical (see block_kernels.hip for the
// argument; nothing is re-associated, no FMA).
//
// What pivot q of a block needs of the not yet materialised tableau T^(q-1):
//   Z row         carried:  z <- z - (f_q[0] * p_q)                  (:208 on row 0)
//   RHS column    carried:  b_i <- b_i - (f_q[i] * p_q[rhs]),  b_r = p_q[rhs]
//   column e_q    gathered from T^(0) in memory, taken through pivots 1..q-1 of the block
//   row r_q       gathered likewise, then divided by the pivot element (:195-199)
// "taken through pivot s": x <- x - (f_s[i] * p_s[j]), or p_s[j] itself on the pivot row r_s.
#include "engine_common.hpp"
#include "select_common.hpp"

#include <cstdlib>
#include <new>

#pragma clang fp contract(off)

namespace lpr {

constexpr int kSmallK = 16;       // pivots per block
constexpr int kSmallNT = 512;     // lanes of the heads' workgroup
constexpr int kSmallMaxR = 1024;  // rows (one per lane)
constexpr int kSmallMaxLd = 2048; // padded columns (one pair per lane)

struct SmallState {
    int32_t status;    // kRunning or the final lpr_status
    int32_t pending;   // kRunning, or the status that ends the solve once this block's sweep is done
    int32_t kdone;     // pivots staged by the heads for the sweep that follows
    int32_t pad;
    int64_t iter;      // pivots applied (the heads add kdone: their sweep always follows in-stream)
    int64_t max_iter;  // <= 0: no limit
    int64_t log_cap;
    int32_t r[kSmallK];  // staged pivot rows
    int32_t e[kSmallK];  // staged pivot columns
};

struct lpr_small_ctx {
    double* prow = nullptr;   // [K][ld]  normalised pivot rows of the block
    double* fcol = nullptr;   // [K][Rp]  factor columns (the column of T^(q-1) before pivot q)
    SmallState* st = nullptr;
    SmallState* h_st = nullptr;  // pinned
    int Rp = 0;
    unsigned long long* dbg = nullptr;  // LPR_SMALL_STAMPS=1
};

// ------------------------------------------------------------------------------------------
// The K loop heads of one block (Solve :107-142), one workgroup of 512 lanes; a lane owns rows
// t, t + 512 and column pairs t, t + 512 (the pivot rows' slices stay in registers: 128 VGPRs of the
// 256 that two waves per SIMD allow; with one row and one pair per lane of a 1024-lane workgroup
// they no longer fit the 128 VGPRs of four waves per SIMD and went to scratch).
constexpr int kNI = kSmallMaxR / kSmallNT;  // items (rows / column pairs) per lane
static_assert(kNI * kSmallNT == kSmallMaxR && 2 * kNI * kSmallNT == kSmallMaxLd, "lane map");

// State of the heads' workgroup, one copy per lane; every array is indexed with compile-time
// constants only (the pivot number is a template parameter), so all of it stays in registers.
struct SmallHead {
    const double* __restrict__ T;
    const double2* __restrict__ T2;
    int ld, ld2, R, C, rhs, Rp, t;
    double* __restrict__ prow_g;
    double* __restrict__ fcol_g;
    int32_t* __restrict__ basis;
    int32_t* __restrict__ log;
    SmallState* st;
    unsigned long long* dbg;
    int64_t it0, max_iter, log_cap;
    double* s_f;        // LDS [K][Rp]: f_s[i]
    double* lds_v;
    int* lds_i;
    double2* s_pe2;     // LDS [K]: p_s at the pair that holds column e (the reader picks .x / .y: a
                        // select between two elements of a register array becomes a dynamically
                        // indexed access and sends the whole array to scratch)
    double2* s_prhs2;
    double* s_piv;
    double* s_f0;
    int row[kNI], cp[kNI];
    bool has_row[kNI], has_col[kNI];
    double2 z2[kNI];            // carried: this lane's pairs of the Z row
    double b[kNI];              // carried: this lane's right-hand sides
    double2 myp[kSmallK][kNI];  // p_s at this lane's column pairs
    int rr[kSmallK];            // pivot rows of the block so far (wave-uniform: SGPRs)
    int pending;

    __device__ __forceinline__ void stamp(int q, int k) const {
        // diagnostic stamps of pivot 8 of the block (LPR_SMALL_STAMPS=1; dbg == nullptr otherwise)
        if (dbg && q == 8 && t == 0) dbg[k] = __builtin_amdgcn_s_memrealtime();
    }

    // Loop head of pivot Q of the block (Solve :107-142).  false: the solve ends here (pending).
    template <int Q>
    __device__ __forceinline__ bool step() {
        stamp(Q, 0);
        // ---- FindEnteringVariable (:152-167): most negative Z entry over j < C - 1, lowest index
        Cand c;
        c.v = 0.0;
        c.i = -1;
#pragma unroll
        for (int u = 0; u < kNI; ++u) {
            if (has_col[u]) {
                const int j0 = 2 * cp[u];
                if (j0 < rhs && z2[u].x < c.v) { c.v = z2[u].x; c.i = j0; }
                if (j0 + 1 < rhs && z2[u].y < c.v) { c.v = z2[u].y; c.i = j0 + 1; }
            }
        }
        c = dpp_block_cand_min16(c, lds_v, lds_i, 0);
        const int e = c.i;
        if (e < 0) {
            pending = LPR_OK_OPTIMAL;
            return false;
        }
        stamp(Q, 1);
        // ---- column e of T^(Q-1): from memory, then through the block's earlier pivots
        double x[kNI];
#pragma unroll
        for (int u = 0; u < kNI; ++u) x[u] = has_row[u] ? T[(size_t)row[u] * ld + e] : 0.0;
#pragma unroll
        for (int u = 0; u < kNI; ++u) {
            if (cp[u] == (e >> 1)) {
#pragma unroll
                for (int s = 0; s < Q; ++s) s_pe2[s] = myp[s][u];
            }
        }
        __syncthreads();
        if (dbg && Q == 8 && t == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg[2] = __builtin_amdgcn_s_memrealtime();
        }
        if constexpr (Q > 0) {
            // operands of the Q steps first, all at once (they do not depend on x: left inside
            // the steps every one of them waited for its own LDS round trips, 1.1 us for 8 steps)
            double pev[Q];
            double fv[Q][kNI];
#pragma unroll
            for (int s = 0; s < Q; ++s) {
                pev[s] = (e & 1) ? s_pe2[s].y : s_pe2[s].x;
#pragma unroll
                for (int u = 0; u < kNI; ++u)
                    fv[s][u] = s_f[(size_t)s * Rp + (has_row[u] ? row[u] : 0)];
            }
#pragma unroll
            for (int s = 0; s < Q; ++s) {
#pragma unroll
                for (int u = 0; u < kNI; ++u) {
                    if (row[u] == rr[s]) {
                        x[u] = pev[s];
                    } else {
                        const double prod = fv[s][u] * pev[s];
                        x[u] = x[u] - prod;
                    }
                }
            }
        }
        stamp(Q, 3);
        // ---- FindLeavingVariable (:169-191): min ratio over rows 1.., a > 1e-9, ratio >= 0
        Cand m;
        m.v = 0.0;
        m.i = -1;
        double rnum[kNI], rden[kNI], rq[kNI];
#pragma unroll
        for (int u = 0; u < kNI; ++u) {  // (a lane without a candidate divides 0 by 1)
            const bool cand = has_row[u] && row[u] >= 1 && x[u] > 1e-9;
            rnum[u] = cand ? b[u] : 0.0;
            rden[u] = cand ? x[u] : 1.0;
        }
        ieee_div_n<kNI>(rnum, rden, rq);
#pragma unroll
        for (int u = 0; u < kNI; ++u) {
            if (has_row[u] && row[u] >= 1 && x[u] > 1e-9) {
                const double ratio = rq[u];
                if (ratio >= 0 && (m.i < 0 || ratio < m.v)) {  // rows ascend with u: ties keep the lower
                    m.v = ratio;
                    m.i = row[u];
                }
            }
        }
        m = dpp_block_cand_min16(m, lds_v, lds_i, 1);
        const int r = m.i;
        if (r < 0) {
            pending = LPR_UNBOUNDED;
            return false;
        }
        if (max_iter > 0 && it0 + Q >= max_iter) {
            pending = LPR_PIVOT_LIMIT;
            return false;
        }
        stamp(Q, 4);
        // ---- row r of T^(Q-1), normalised (:195-199)
        double2 y2[kNI];
#pragma unroll
        for (int u = 0; u < kNI; ++u)
            y2[u] = has_col[u] ? T2[(size_t)r * ld2 + cp[u]] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < kNI; ++u) {
            if (row[u] == r) *s_piv = x[u];   // the pivot element: the column's entry on row r
            if (row[u] == 0) *s_f0 = x[u];    // f_Q[0]
            if (has_row[u]) s_f[(size_t)Q * Rp + row[u]] = x[u];
        }
        __syncthreads();
        if (dbg && Q == 8 && t == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg[5] = __builtin_amdgcn_s_memrealtime();
        }
        if constexpr (Q > 0) {
            double fr[Q];
#pragma unroll
            for (int s = 0; s < Q; ++s) fr[s] = s_f[(size_t)s * Rp + r];
#pragma unroll
            for (int s = 0; s < Q; ++s) {
                if (r == rr[s]) {
#pragma unroll
                    for (int u = 0; u < kNI; ++u) y2[u] = myp[s][u];
                } else {
#pragma unroll
                    for (int u = 0; u < kNI; ++u) {
                        const double px = fr[s] * myp[s][u].x;
                        const double py = fr[s] * myp[s][u].y;
                        y2[u].x = y2[u].x - px;
                        y2[u].y = y2[u].y - py;
                    }
                }
            }
        }
        stamp(Q, 6);
        const double piv = *s_piv;
        double2 p2[kNI];
        {
            double pn[2 * kNI], pd[2 * kNI], pq[2 * kNI];
#pragma unroll
            for (int u = 0; u < kNI; ++u) {
                pn[2 * u] = y2[u].x;
                pn[2 * u + 1] = y2[u].y;
                pd[2 * u] = pd[2 * u + 1] = piv;
            }
            ieee_div_n<2 * kNI>(pn, pd, pq);
#pragma unroll
            for (int u = 0; u < kNI; ++u) p2[u] = make_double2(pq[2 * u], pq[2 * u + 1]);
        }
#pragma unroll
        for (int u = 0; u < kNI; ++u) {
            if (has_col[u] && (rhs >> 1) == cp[u]) *s_prhs2 = p2[u];
            if (has_col[u]) reinterpret_cast<double2*>(prow_g + (size_t)Q * ld)[cp[u]] = p2[u];
            if (has_row[u]) fcol_g[(size_t)Q * Rp + row[u]] = x[u];
            myp[Q][u] = p2[u];
        }
        rr[Q] = r;
        __syncthreads();
        stamp(Q, 7);
        // ---- carry the Z row and the right-hand side to T^(Q) (:202-210 on row 0 / column rhs)
        {
            const double f0 = *s_f0, prhs = (rhs & 1) ? s_prhs2->y : s_prhs2->x;
#pragma unroll
            for (int u = 0; u < kNI; ++u) {
                const double px = f0 * p2[u].x;   // (r >= 1: row 0 is never the pivot row)
                const double py = f0 * p2[u].y;
                z2[u].x = z2[u].x - px;
                z2[u].y = z2[u].y - py;
                if (row[u] == r) {
                    b[u] = prhs;
                } else {
                    const double prod = x[u] * prhs;
                    b[u] = b[u] - prod;
                }
            }
        }
        stamp(Q, 8);
        if (t == 0) {
            st->r[Q] = r;
            st->e[Q] = e;
            basis[r - 1] = e;  // :142
            const int64_t it = it0 + Q;
            if (it < log_cap) {
                log[2 * it] = r;
                log[2 * it + 1] = e;
            }
        }
        return true;
    }
};

// pivots Q, Q + 1, ... of the block; returns the number staged
template <int Q>
__device__ __forceinline__ int small_run(SmallHead& h) {
    if constexpr (Q == kSmallK) {
        return Q;
    } else {
        if (!h.template step<Q>()) return Q;
        return small_run<Q + 1>(h);
    }
}

__global__ __launch_bounds__(kSmallNT) void k_small_heads(const double* __restrict__ T, int ld, int R,
                                                          int C, double* __restrict__ prow_g,
                                                          double* __restrict__ fcol_g, int Rp,
                                                          int32_t* __restrict__ basis,
                                                          int32_t* __restrict__ log,
                                                          SmallState* st,
                                                          unsigned long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) double s_f[];  // [K][Rp]: f_s[i]
    __shared__ double lds_v[32];
    __shared__ int lds_i[32];
    __shared__ double2 s_pe2[kSmallK];
    __shared__ double2 s_prhs2;
    __shared__ double s_piv, s_f0;
    const int t = threadIdx.x;
    const int32_t status = st->status;
    const int32_t pend0 = st->pending;
    if (status != kRunning) return;
    if (pend0 != kRunning) {  // the sweep of the block before this one has applied its pivots
        if (t == 0) {
            st->status = pend0;
            st->kdone = 0;
        }
        return;
    }
    SmallHead h;
    h.T = T;
    h.T2 = reinterpret_cast<const double2*>(T);
    h.ld = ld; h.ld2 = ld >> 1; h.R = R; h.C = C; h.rhs = C - 1; h.Rp = Rp; h.t = t;
    h.prow_g = prow_g; h.fcol_g = fcol_g; h.basis = basis; h.log = log; h.st = st; h.dbg = dbg;
    h.it0 = st->iter; h.max_iter = st->max_iter; h.log_cap = st->log_cap;
    h.s_f = s_f; h.lds_v = lds_v; h.lds_i = lds_i; h.s_pe2 = s_pe2; h.s_prhs2 = &s_prhs2;
    h.s_piv = &s_piv; h.s_f0 = &s_f0;
    h.pending = kRunning;
#pragma unroll
    for (int u = 0; u < kNI; ++u) {
        h.row[u] = t + u * kSmallNT;
        h.cp[u] = t + u * kSmallNT;
        h.has_row[u] = h.row[u] < R;
        h.has_col[u] = h.cp[u] < h.ld2;
        h.z2[u] = h.has_col[u] ? h.T2[h.cp[u]] : make_double2(0.0, 0.0);
        h.b[u] = h.has_row[u] ? T[(size_t)h.row[u] * ld + h.rhs] : 0.0;
#pragma unroll
        for (int s = 0; s < kSmallK; ++s) h.myp[s][u] = make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int s = 0; s < kSmallK; ++s) h.rr[s] = -1;
    // (one copy of the loop head per pivot of the block: the chains of pivot q then have exactly q
    // steps -- rolled, with the steps of pivots that do not exist yet skipped by a uniform branch,
    // the heads took 17 % longer: 150 k against 176 k pivots/s at m = 512)
    const int q = small_run<0>(h);
    if (t == 0) {
        st->kdone = q;
        st->iter = h.it0 + q;
        st->pending = h.pending;
        if (h.pending != kRunning && q == 0) st->status = h.pending;  // nothing left to sweep
    }
}

// ------------------------------------------------------------------------------------------
// The block applied to the tableau, in place.  grid (ceil(ld2 / 256), ceil(R / TR)).  Each element
// goes through the staged pivots in order; the pivot rows' slices of a lane stay in registers, the
// factors f_s[i] are the same for the whole workgroup.
template <int TR>
__global__ __launch_bounds__(256) void k_small_sweep(double* __restrict__ T, int ld, int R,
                                                     const double* __restrict__ prow,
                                                     const double* __restrict__ fcol, int Rp,
                                                     const SmallState* __restrict__ st) {
    const int kd = st->kdone;
    if (kd <= 0) return;
    const int ld2 = ld >> 1;
    const int c2 = blockIdx.x * blockDim.x + threadIdx.x;
    const int i0 = blockIdx.y * TR;
    if (c2 >= ld2) return;
    // Straight-line code for all K pivots and TR rows: a pivot that was not staged has p = 0 and
    // f = 0, and x - (0 * 0) is x for every x; a row past the end is computed on row R - 1's data
    // and not stored.  (With a branch per (pivot, row) the kernel was 424 branches long and every
    // factor was waited for on its own: 12.6 us for 6.3 MB.)
    double2 p[kSmallK];
    int rs[kSmallK];
#pragma unroll
    for (int s = 0; s < kSmallK; ++s) {
        const double2 v = reinterpret_cast<const double2*>(prow + (size_t)(s < kd ? s : 0) * ld)[c2];
        p[s].x = (s < kd) ? v.x : 0.0;
        p[s].y = (s < kd) ? v.y : 0.0;
        rs[s] = (s < kd) ? st->r[s] : -1;
    }
    double2* __restrict__ T2 = reinterpret_cast<double2*>(T);
    double2 x[TR];
    int ir[TR];
#pragma unroll
    for (int k = 0; k < TR; ++k) {
        ir[k] = (i0 + k < R) ? i0 + k : R - 1;
        x[k] = T2[(size_t)ir[k] * ld2 + c2];
    }
#pragma unroll
    for (int s = 0; s < kSmallK; ++s) {
        double f[TR];  // the same for every lane: scalar loads
#pragma unroll
        for (int k = 0; k < TR; ++k) {
            const double fv = fcol[(size_t)(s < kd ? s : 0) * Rp + ir[k]];
            f[k] = (s < kd) ? fv : 0.0;
        }
#pragma unroll
        for (int k = 0; k < TR; ++k) {
            const double px = f[k] * p[s].x;
            const double py = f[k] * p[s].y;
            const double ox = x[k].x - px;
            const double oy = x[k].y - py;
            const bool piv = ir[k] == rs[s];
            x[k].x = piv ? p[s].x : ox;
            x[k].y = piv ? p[s].y : oy;
        }
    }
#pragma unroll
    for (int k = 0; k < TR; ++k)
        if (i0 + k < R) T2[(size_t)(i0 + k) * ld2 + c2] = x[k];
}

// ------------------------------------------------------------------------------------------
// host side (called by lpr_engine.hip)

bool small_fits(const lpr_tableau* t) {
    return t->rows >= 2 && t->rows <= kSmallMaxR && t->ld <= kSmallMaxLd &&
           (size_t)kSmallK * (size_t)align_up(t->rows, 16) * sizeof(double) <= (size_t)(140 << 10);
}
int small_pivots_per_block() { return kSmallK; }

void small_release(lpr_tableau* t) {
    lpr_small_ctx* c = static_cast<lpr_small_ctx*>(t->small);
    if (!c) return;
    hipFree(c->prow);
    hipFree(c->fcol);
    hipFree(c->st);
    hipFree(c->dbg);
    if (c->h_st) hipHostFree(c->h_st);
    delete c;
    t->small = nullptr;
}

int small_ensure(lpr_tableau* t) {
    if (t->small) return LPR_OK_OPTIMAL;
    lpr_small_ctx* c = new (std::nothrow) lpr_small_ctx();
    if (!c) return LPR_OUT_OF_MEMORY;
    c->Rp = align_up(t->rows, 16);
    hipError_t err = hipMalloc(&c->prow, (size_t)kSmallK * t->ld * sizeof(double));
    if (err == hipSuccess) err = hipMalloc(&c->fcol, (size_t)kSmallK * c->Rp * sizeof(double));
    if (err == hipSuccess) err = hipMalloc(&c->st, sizeof(SmallState));
    if (err == hipSuccess) err = hipHostMalloc(&c->h_st, sizeof(SmallState));
    if (const char* sv = std::getenv("LPR_SMALL_STAMPS"); err == hipSuccess && sv && sv[0] == '1') {
        err = hipMalloc(&c->dbg, 16 * sizeof(unsigned long long));
        if (err == hipSuccess) err = hipMemset(c->dbg, 0, 16 * sizeof(unsigned long long));
    }
    if (err == hipSuccess)
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_small_heads),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 140 << 10);
    t->small = c;
    if (err != hipSuccess) {
        set_error("small-tableau path: %s", hipGetErrorString(err));
        small_release(t);
        return err == hipErrorOutOfMemory ? LPR_OUT_OF_MEMORY : LPR_DEVICE_ERROR;
    }
    return LPR_OK_OPTIMAL;
}

int small_upload_state(lpr_tableau* t, int64_t iter, int64_t max_iter) {
    lpr_small_ctx* c = static_cast<lpr_small_ctx*>(t->small);
    std::memset(c->h_st, 0, sizeof(SmallState));
    c->h_st->status = kRunning;
    c->h_st->pending = kRunning;
    c->h_st->iter = iter;
    c->h_st->max_iter = max_iter;
    c->h_st->log_cap = t->log_cap;
    LPR_HIP(hipMemcpyAsync(c->st, c->h_st, sizeof(SmallState), hipMemcpyHostToDevice,
                           t->eng->stream));
    return LPR_OK_OPTIMAL;
}

int small_set_log_cap(lpr_tableau* t) {
    lpr_small_ctx* c = static_cast<lpr_small_ctx*>(t->small);
    c->h_st->log_cap = t->log_cap;
    LPR_HIP(hipMemcpyAsync(&c->st->log_cap, &c->h_st->log_cap, sizeof(int64_t),
                           hipMemcpyHostToDevice, t->eng->stream));
    return LPR_OK_OPTIMAL;
}

// one block: the K loop heads, then the sweep
void small_launch_block(lpr_tableau* t) {
    lpr_small_ctx* c = static_cast<lpr_small_ctx*>(t->small);
    hipStream_t s = t->eng->stream;
    const size_t lds = (size_t)kSmallK * c->Rp * sizeof(double);
    hipLaunchKernelGGL(k_small_heads, dim3(1), dim3(kSmallNT), lds, s, t->T, t->ld, t->rows,
                       t->cols, c->prow, c->fcol, c->Rp, t->basis, t->log, c->st, c->dbg);
    constexpr int TR = 4;
    hipLaunchKernelGGL((k_small_sweep<TR>), dim3((t->ld / 2 + 255) / 256, (t->rows + TR - 1) / TR),
                       dim3(256), 0, s, t->T, t->ld, t->rows, c->prow, c->fcol, c->Rp, c->st);
}

int small_poll(lpr_tableau* t, int32_t* status, int64_t* iter) {
    lpr_small_ctx* c = static_cast<lpr_small_ctx*>(t->small);
    hipStream_t s = t->eng->stream;
    LPR_HIP(hipMemcpyAsync(c->h_st, c->st, sizeof(SmallState), hipMemcpyDeviceToHost, s));
    LPR_HIP(hipStreamSynchronize(s));
    if (c->dbg) {
        unsigned long long d[9];
        LPR_HIP(hipMemcpy(d, c->dbg, sizeof d, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "small heads, pivot 8 of a block (us): entering %.2f | column gather %.2f "
                     "chain %.2f | ratio+leaving %.2f | row gather %.2f chain %.2f | divide+stores "
                     "%.2f | carry %.2f | total %.2f\n",
                     (d[1] - d[0]) * 0.01, (d[2] - d[1]) * 0.01, (d[3] - d[2]) * 0.01,
                     (d[4] - d[3]) * 0.01, (d[5] - d[4]) * 0.01, (d[6] - d[5]) * 0.01,
                     (d[7] - d[6]) * 0.01, (d[8] - d[7]) * 0.01, (d[8] - d[0]) * 0.01);
    }
    *status = c->h_st->status;
    *iter = c->h_st->iter;
    return LPR_OK_OPTIMAL;
}

}  // namespace lpr
