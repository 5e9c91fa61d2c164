// block_kernels.hip -- K consecutive primal pivots per sweep of the tableau.
// (reference: LPR_381_Group_V22/Simplex/PrimalSimplexSolver.cs:102-211)
//
// A pivot costs one read + one write of the whole tableau (2*8*R*C bytes) and the rank-1 update
// kernel already moves them at the speed of the memory system.  The only way to go faster is to
// move fewer bytes per pivot: this path decides K pivots FIRST, from O(R + C) data each, and then
// applies all K to every element in ONE sweep -- the element goes through the same K
// multiply-round-subtract-round steps it would go through in K separate sweeps, in registers, so
// the stored bits are identical to the C#'s.
//
// What pivot q of a block needs of the tableau T^(q-1) (T^(0) = the tableau in memory):
//   * its Z row, to pick the entering column (:152-167)       -> zrow, carried analytically:
//         Z^(q) = Z^(q-1) - (f_q[0] * p_q)                       (the expression :208 stores)
//   * column e_q and the RHS column, for the ratio test (:169-191) and as factors f_q (:206):
//         column: gathered from T^(0) and taken through the q-1 earlier pivots (k_blk_gather);
//         RHS:    b^(q) = b^(q-1) - (f_q * p_q[rhs]), b^(q)[r_q] = p_q[rhs]
//   * row r_q, normalised (:198-199): read from T^(0), taken through the q-1 earlier pivots.
// "Taken through pivot s": x -> x - (f_s[i] * p_s[j]), or p_s[j] itself on the pivot row r_s --
// exactly what the sweep does to every element.  The first pivot of a block uses the dense column
// / RHS that the previous sweep dumped (next_col / next_rhs), like the one-pivot path.
//
// Kernels per block: k_blk_head(1), then k_blk_gather(q) + k_blk_head(q) for q = 2..K, then
// k_blk_update.  A head that meets the end of the solve (optimal, unbounded, pivot limit) records
// it in `pending`; the pivots staged before it are still applied by the sweep, later heads of the
// block fall through, and the first head of the next block (or the host) publishes the status.
#include "engine_common.hpp"
#include "select_common.hpp"

#include <new>

#pragma clang fp contract(off)

namespace lpr {

constexpr int kBlkMax = 8;       // pivots per sweep, upper bound (register budget of the sweep)
constexpr int kBlkHeadNT = 512;  // threads per head workgroup

struct BlockState {
    int32_t status;        // kRunning or the final lpr_status
    int32_t pending;       // kRunning, or the status that ends the solve after this block's sweep
    int32_t kdone;         // pivots staged by the heads of the current block
    int32_t sweep;         // parity of the sweep direction
    int32_t r[kBlkMax];    // staged pivot rows
    int32_t e[kBlkMax];    // staged pivot columns
    int64_t iter;          // pivots applied; committed by the sweep, never read by it
    int64_t max_iter;      // <= 0: no limit
    int64_t log_cap;
    int64_t iter_pending;  // iter + kdone (written by the heads' lead workgroup)
};

// ------------------------------------------------------------------------------------------
// Primes the loop like k_bootstrap: entering column of the tableau in memory -> partial 0 of the
// current bank; that column and the RHS column, densely.
__global__ __launch_bounds__(1024) void k_blk_bootstrap(const double* __restrict__ T, int ld,
                                                        int R, int C,
                                                        double* __restrict__ next_col,
                                                        double* __restrict__ next_rhs,
                                                        BlockState* st,
                                                        ZPart* __restrict__ zparts, int G) {
    __shared__ double lds_v[16];
    __shared__ int lds_i[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    Cand c;
    c.v = 0.0;
    c.i = -1;
    for (int j = tid; j < C - 1; j += nt) {
        const double v = T[j];
        if (v < c.v) {
            c.v = v;
            c.i = j;
        }
    }
    c = block_cand_min(c, lds_v, lds_i);
    const int e = c.i;
    ZPart* bank = zparts + (st->iter & 1) * kMaxHeadGroups;
    if (tid < G) {
        bank[tid].v = (tid == 0) ? c.v : 0.0;
        bank[tid].i = (tid == 0) ? e : -1;
    }
    if (tid == 0) {
        st->iter_pending = st->iter;
        st->kdone = 0;
    }
    if (e < 0) return;
    const int rhs = C - 1;
    for (int i = tid; i < R; i += nt) {
        next_col[i] = T[(size_t)i * ld + e];
        next_rhs[i] = T[(size_t)i * ld + rhs];
    }
}

// ------------------------------------------------------------------------------------------
// Column e_q of T^(q-1), q >= 2: the strided gather from the tableau in memory, spread over many
// workgroups (one CU cannot walk 4097 rows x 98 KB stride quickly), then through pivots 1..q-1.
// Two dependent memory trips: everything whose address is known up front (control block, both
// partial banks, this row's factors) is requested first, then T[i, e] and p_s[e].
__global__ __launch_bounds__(128) void k_blk_gather(const double* __restrict__ T, int ld, int R,
                                                    int q, const double* __restrict__ prow,
                                                    double* __restrict__ fcol, int Rp,
                                                    const BlockState* st,
                                                    const ZPart* __restrict__ zparts, int G) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t status = st->status;
    const int32_t pending = st->pending;
    const int64_t it0 = st->iter;
    const Cand e0 = reduce_zparts(zparts, G);
    const Cand e1 = reduce_zparts(zparts + kMaxHeadGroups, G);
    int rs[kBlkMax];
    double f[kBlkMax];
#pragma unroll
    for (int s = 0; s < kBlkMax; ++s) {
        rs[s] = (s < q - 1) ? st->r[s] : -1;
        f[s] = (s < q - 1 && i < R) ? fcol[(size_t)s * Rp + i] : 0.0;
    }
    if (status != kRunning || pending != kRunning) return;
    const int e = (((it0 + q - 1) & 1) ? e1 : e0).i;
    if (e < 0 || i >= R) return;
    double c = T[(size_t)i * ld + e];
    double pe[kBlkMax];
#pragma unroll
    for (int s = 0; s < kBlkMax; ++s) pe[s] = (s < q - 1) ? prow[(size_t)s * ld + e] : 0.0;
#pragma unroll
    for (int s = 0; s < kBlkMax; ++s) {
        if (s < q - 1) {
            if (i == rs[s]) {
                c = pe[s];
            } else {
                const double prod = f[s] * pe[s];
                c = c - prod;
            }
        }
    }
    fcol[(size_t)(q - 1) * Rp + i] = c;
}

// ------------------------------------------------------------------------------------------
// Loop head of pivot q of the block (Solve :107-142): G workgroups.
//   1. e_q from the G partials of bank (pivot index & 1); none -> optimal (:110-126).
//   2. ratio test (:169-191) on the dense column / RHS of T^(q-1); every workgroup does the whole
//      scan itself and all arrive at the same r_q; none -> unbounded (:129-135).
//   3. by column slices: row r_q of T^(q-1) (memory row through pivots 1..q-1), normalised (:199)
//      -> prow[q-1]; Z^(q) -> zrow and this slice's partial arg-min -> the other bank.
//   4. by row slices: b^(q) -> bvec[q & 1]; the factor column of pivot 1 is copied out of
//      next_col (the sweep overwrites next_col while it still needs the factors).
// Only workgroup 0 writes the control block, and only fields no workgroup of this launch reads.
// Memory trips: (1) control block, both banks, the dense column / RHS, this lane's slices of the
// earlier pivot rows and of the Z row -- all addresses known up front; (2) what depends on r_q:
// the row itself and the factors f_s[r_q].
constexpr int kBlkHeadU = 9;  // dense-vector elements preloaded per lane (9 * 512 >= 4097)

__global__ __launch_bounds__(kBlkHeadNT) void k_blk_head(
    const double* __restrict__ T, int ld, int R, int C, int q, double* __restrict__ prow,
    double* __restrict__ fcol, int Rp, const double* __restrict__ next_col,
    const double* __restrict__ next_rhs, double* __restrict__ zrow, double* __restrict__ bvec,
    int32_t* __restrict__ basis, int32_t* __restrict__ log, BlockState* st,
    ZPart* __restrict__ zparts, int G) {
    __shared__ double lds_v[16];
    __shared__ int lds_i[16];
    __shared__ double lds_p[2];  // [0] = T^(q-1)[r, e], [1] = T^(q-1)[0, e]
    const int tid = threadIdx.x, nt = blockDim.x;
    const int g = blockIdx.x;
    const bool lead = (g == 0);
    const int ld2 = ld >> 1;
    const int rhs = C - 1;
    const double2* __restrict__ T2 = reinterpret_cast<const double2*>(T);
    double2* __restrict__ prow2 = reinterpret_cast<double2*>(prow);
    double2* __restrict__ zrow2 = reinterpret_cast<double2*>(zrow);
    const double* __restrict__ col = (q == 1) ? next_col : fcol + (size_t)(q - 1) * Rp;
    const double* __restrict__ bprev = (q == 1) ? next_rhs : bvec + (size_t)((q - 1) & 1) * Rp;
    double* __restrict__ bnew = bvec + (size_t)(q & 1) * Rp;

    // ---- trip 1 ----
    const int32_t status = st->status;
    const int32_t pending = st->pending;
    const int64_t it0 = st->iter;
    const int64_t mx = st->max_iter;
    const Cand e0 = reduce_zparts(zparts, G);
    const Cand e1 = reduce_zparts(zparts + kMaxHeadGroups, G);
    double a0[kBlkHeadU], b0[kBlkHeadU];
#pragma unroll
    for (int u = 0; u < kBlkHeadU; ++u) {
        const int i = tid + u * nt;
        a0[u] = (i < R) ? col[i] : 0.0;
        b0[u] = (i < R) ? bprev[i] : 0.0;
    }
    const int c2_first = g * nt + tid;
    const bool have_c2 = c2_first < ld2;
    int rs[kBlkMax];
    double prs[kBlkMax];
    double2 ps0[kBlkMax];
#pragma unroll
    for (int s = 0; s < kBlkMax; ++s) {
        rs[s] = (s < q - 1) ? st->r[s] : -1;
        prs[s] = (s < q - 1) ? prow[(size_t)s * ld + rhs] : 0.0;
        ps0[s] = (s < q - 1 && have_c2) ? prow2[(size_t)s * ld2 + c2_first]
                                        : make_double2(0.0, 0.0);
    }
    double2 z0 = make_double2(0.0, 0.0);
    if (have_c2) z0 = (q == 1) ? T2[c2_first] : zrow2[c2_first];

    if (status != kRunning) return;
    if (pending != kRunning) {
        if (q == 1 && lead && tid == 0) st->status = pending;  // the block before ended the solve
        return;
    }
    const int64_t pidx = it0 + q - 1;  // index of this pivot in the solve
    ZPart* __restrict__ bank_out = zparts + ((pidx + 1) & 1) * kMaxHeadGroups;
    const int e = ((pidx & 1) ? e1 : e0).i;
    if (e < 0) {
        if (lead && tid == 0) {
            st->pending = LPR_OK_OPTIMAL;
            if (q == 1) st->kdone = 0;
        }
        return;
    }

    // ---- FindLeavingVariable on the dense vectors ----
    Cand c;
    c.v = DBL_MAX;
    c.i = -1;
    double a_of_best = 0.0;
    for (int i0 = tid; i0 < R; i0 += kBlkHeadU * nt) {
        double a[kBlkHeadU], b[kBlkHeadU];
#pragma unroll
        for (int u = 0; u < kBlkHeadU; ++u) {
            const int i = i0 + u * nt;
            if (i0 == tid) {
                a[u] = a0[u];
                b[u] = b0[u];
            } else {
                a[u] = (i < R) ? col[i] : 0.0;
                b[u] = (i < R) ? bprev[i] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < kBlkHeadU; ++u) {
            const int i = i0 + u * nt;
            if (i < R) {
                if (i == 0) lds_p[1] = a[u];
                if (i >= 1 && a[u] > 1e-9) {
                    const double ratio = ieee_div(b[u], a[u]);
                    if (ratio >= 0 && ratio < c.v) {
                        c.v = ratio;
                        c.i = i;
                        a_of_best = a[u];
                    }
                }
            }
        }
    }
    const int my_best = c.i;
    c = block_cand_min(c, lds_v, lds_i);
    const int r = c.i;
    if (r < 0) {
        if (lead && tid == 0) {
            st->pending = LPR_UNBOUNDED;
            if (q == 1) st->kdone = 0;
        }
        return;
    }
    if (mx > 0 && pidx >= mx) {
        if (lead && tid == 0) {
            st->pending = LPR_PIVOT_LIMIT;
            if (q == 1) st->kdone = 0;
        }
        return;
    }
    if (my_best == r) lds_p[0] = a_of_best;  // exactly one lane owns row r

    // ---- trip 2: what the earlier pivots of the block do to row r, and the row itself ----
    double fr[kBlkMax];
#pragma unroll
    for (int s = 0; s < kBlkMax; ++s) fr[s] = (s < q - 1) ? fcol[(size_t)s * Rp + r] : 0.0;
    double wr = T[(size_t)r * ld + rhs];
    double2 w0 = have_c2 ? T2[(size_t)r * ld2 + c2_first] : make_double2(0.0, 0.0);
    __syncthreads();
    const double p = lds_p[0];
    const double f0 = lds_p[1];

    // ---- row r through pivots 1..q-1, normalise, next Z row, partial arg-min ----
    Cand n;
    n.v = 0.0;
    n.i = -1;
    for (int c2 = c2_first; c2 < ld2; c2 += G * nt) {
        double2 w, z;
        if (c2 == c2_first) {
            w = w0;
            z = z0;
        } else {
            w = T2[(size_t)r * ld2 + c2];
            z = (q == 1) ? T2[c2] : zrow2[c2];
        }
#pragma unroll
        for (int s = 0; s < kBlkMax; ++s) {
            if (s < q - 1) {
                const double2 ps = (c2 == c2_first) ? ps0[s] : prow2[(size_t)s * ld2 + c2];
                if (r == rs[s]) {
                    w = ps;
                } else {
                    const double px = fr[s] * ps.x;
                    const double py = fr[s] * ps.y;
                    w.x = w.x - px;
                    w.y = w.y - py;
                }
            }
        }
        const int j = 2 * c2;
        double2 pq;
        pq.x = (j < C) ? ieee_div(w.x, p) : 0.0;  // :199 true division
        pq.y = (j + 1 < C) ? ieee_div(w.y, p) : 0.0;
        prow2[(size_t)(q - 1) * ld2 + c2] = pq;
        const double mxp = f0 * pq.x;  // :208 product rounded, then the difference
        const double myp = f0 * pq.y;
        z.x = z.x - mxp;
        z.y = z.y - myp;
        zrow2[c2] = z;
        if (j < C - 1 && z.x < n.v) {
            n.v = z.x;
            n.i = j;
        }
        if (j + 1 < C - 1 && z.y < n.v) {
            n.v = z.y;
            n.i = j + 1;
        }
    }
    n = block_cand_min(n, lds_v, lds_i);

    // ---- RHS column after this pivot (and the factor column of pivot 1) ----
#pragma unroll
    for (int s = 0; s < kBlkMax; ++s) {
        if (s < q - 1) {
            if (r == rs[s]) {
                wr = prs[s];
            } else {
                const double prod = fr[s] * prs[s];
                wr = wr - prod;
            }
        }
    }
    const double prhs = ieee_div(wr, p);
    for (int i = g * nt + tid; i < R; i += G * nt) {
        const double a = col[i];
        if (q == 1) fcol[i] = a;
        const double prod = a * prhs;
        bnew[i] = (i == r) ? prhs : bprev[i] - prod;
    }

    if (tid == 0) {
        bank_out[g].v = n.v;
        bank_out[g].i = n.i;
        if (lead) {
            st->r[q - 1] = r;
            st->e[q - 1] = e;
            st->kdone = q;
            st->iter_pending = pidx + 1;
            if (q == 1) st->sweep ^= 1;
            basis[r - 1] = e;  // :142
            if (pidx < st->log_cap) {
                log[2 * pidx] = r;
                log[2 * pidx + 1] = e;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The sweep: every element goes through the kdone staged pivots (:202-210 each), in registers.
// Tile = TR rows x 256 double2 columns per 256-thread workgroup, all loads of a lane issued before
// its first store, serpentine order over the grid on alternate blocks.  The lanes that own the
// next entering column / the RHS column dump their final values densely for the next head.
template <int TR>
__global__ __launch_bounds__(256) void k_blk_update(double* __restrict__ T, int ld, int R, int C,
                                                    const double* __restrict__ prow,
                                                    const double* __restrict__ fcol, int Rp,
                                                    double* __restrict__ next_col,
                                                    double* __restrict__ next_rhs, BlockState* st,
                                                    const ZPart* __restrict__ zparts, int G) {
    if (st->status != kRunning) return;
    const int K = st->kdone;
    if (K <= 0) return;
    const int64_t itp = st->iter_pending;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) st->iter = itp;
    const int ne = (st->pending == kRunning)
                       ? reduce_zparts(zparts + (itp & 1) * kMaxHeadGroups, G).i
                       : -1;
    const int ld2 = ld >> 1;
    int ct = blockIdx.x, rt = blockIdx.y;
    if (st->sweep & 1) {
        ct = gridDim.x - 1 - ct;
        rt = gridDim.y - 1 - rt;
    }
    const int c2 = ct * 256 + threadIdx.x;
    if (c2 >= ld2) return;
    const int i0 = rt * TR;
    int rr[kBlkMax];
    double2 p[kBlkMax];
    const double2* __restrict__ prow2 = reinterpret_cast<const double2*>(prow);
#pragma unroll
    for (int s = 0; s < kBlkMax; ++s) {
        rr[s] = (s < K) ? st->r[s] : -1;
        p[s] = (s < K) ? prow2[(size_t)s * ld2 + c2] : make_double2(0.0, 0.0);
    }
    double2* __restrict__ T2 = reinterpret_cast<double2*>(T);
    const int rhs = C - 1;
    const bool own_rhs = (c2 == (rhs >> 1));
    const bool own_ne = (ne >= 0 && c2 == (ne >> 1));
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
            double2 t = x[k];
#pragma unroll
            for (int s = 0; s < kBlkMax; ++s) {
                if (s < K) {
                    if (i == rr[s]) {
                        t = p[s];  // the pivot row keeps the normalised values (:199)
                    } else {
                        const double f = fcol[(size_t)s * Rp + i];
                        const double px = f * p[s].x;  // product rounded ...
                        const double py = f * p[s].y;
                        t.x = t.x - px;                // ... then the difference (:208)
                        t.y = t.y - py;
                    }
                }
            }
            T2[(size_t)i * ld2 + c2] = t;
            if (own_ne) next_col[i] = (ne & 1) ? t.y : t.x;
            if (own_rhs) next_rhs[i] = (rhs & 1) ? t.y : t.x;
        }
    }
}

}  // namespace lpr

// ---------------------------------------------------------------------------------------------
// host side (the driver loop lives in lpr_engine.hip)

struct lpr_block_ctx {
    int rows = 0, ld = 0, Rp = 0;
    double* prow = nullptr;  // kBlkMax x ld
    double* fcol = nullptr;  // kBlkMax x Rp
    double* zrow = nullptr;  // ld
    double* bvec = nullptr;  // 2 x Rp
    lpr::BlockState* state = nullptr;
    lpr::BlockState* h_state = nullptr;
};

namespace lpr {

int blk_max_pivots() { return kBlkMax; }

void blk_release(lpr_tableau* t) {
    lpr_block_ctx* c = static_cast<lpr_block_ctx*>(t->blk);
    if (!c) return;
    if (t->graph) {  // a captured batch of this path holds the scratch pointers freed below
        hipGraphExecDestroy(t->graph);
        t->graph = nullptr;
        t->graph_batch = 0;
        t->graph_variant = -1;
        t->graph_key = lpr_tableau::GraphKey();
    }
    hipFree(c->prow);
    hipFree(c->fcol);
    hipFree(c->zrow);
    hipFree(c->bvec);
    hipFree(c->state);
    if (c->h_state) hipHostFree(c->h_state);
    delete c;
    t->blk = nullptr;
}

int blk_ensure(lpr_tableau* t) {
    lpr_block_ctx* c = static_cast<lpr_block_ctx*>(t->blk);
    if (c && c->rows == t->rows && c->ld == t->ld) return LPR_OK_OPTIMAL;
    blk_release(t);
    c = new (std::nothrow) lpr_block_ctx();
    if (!c) return LPR_OUT_OF_MEMORY;
    c->rows = t->rows;
    c->ld = t->ld;
    c->Rp = align_up(t->rows, 16);
    hipError_t err = hipSuccess;
    auto chk = [&](hipError_t x) { if (err == hipSuccess) err = x; };
    const size_t D = sizeof(double);
    chk(hipMalloc(&c->prow, (size_t)kBlkMax * c->ld * D));
    chk(hipMalloc(&c->fcol, (size_t)kBlkMax * c->Rp * D));
    chk(hipMalloc(&c->zrow, (size_t)c->ld * D));
    chk(hipMalloc(&c->bvec, (size_t)2 * c->Rp * D));
    chk(hipMalloc(&c->state, sizeof(BlockState)));
    chk(hipHostMalloc(&c->h_state, sizeof(BlockState)));
    t->blk = c;
    if (err != hipSuccess) {
        set_error("blocked-pivot scratch allocation failed: %s", hipGetErrorString(err));
        blk_release(t);
        return err == hipErrorOutOfMemory ? LPR_OUT_OF_MEMORY : LPR_DEVICE_ERROR;
    }
    hipStream_t s = t->eng->stream;
    LPR_HIP(hipMemsetAsync(c->prow, 0, (size_t)kBlkMax * c->ld * D, s));
    LPR_HIP(hipMemsetAsync(c->fcol, 0, (size_t)kBlkMax * c->Rp * D, s));
    LPR_HIP(hipMemsetAsync(c->zrow, 0, (size_t)c->ld * D, s));
    LPR_HIP(hipMemsetAsync(c->bvec, 0, (size_t)2 * c->Rp * D, s));
    std::memset(c->h_state, 0, sizeof(BlockState));
    return LPR_OK_OPTIMAL;
}

static int blk_groups(const lpr_tableau* t) {
    int g = (t->ld / 2 + kBlkHeadNT - 1) / kBlkHeadNT;
    if (g < 1) g = 1;
    if (g > kMaxHeadGroups) g = kMaxHeadGroups;
    return g;
}

// host mirror of the control block: (status, pending, iter)
void* blk_host_state(lpr_tableau* t) { return static_cast<lpr_block_ctx*>(t->blk)->h_state; }

int blk_upload_state(lpr_tableau* t, int64_t iter, int64_t max_iter) {
    lpr_block_ctx* c = static_cast<lpr_block_ctx*>(t->blk);
    BlockState* hs = c->h_state;
    std::memset(hs, 0, sizeof(BlockState));
    hs->status = kRunning;
    hs->pending = kRunning;
    hs->iter = iter;
    hs->iter_pending = iter;
    hs->max_iter = max_iter;
    hs->log_cap = t->log_cap;
    LPR_HIP(hipMemcpyAsync(c->state, hs, sizeof(BlockState), hipMemcpyHostToDevice,
                           t->eng->stream));
    return LPR_OK_OPTIMAL;
}

int blk_set_log_cap(lpr_tableau* t) {
    lpr_block_ctx* c = static_cast<lpr_block_ctx*>(t->blk);
    c->h_state->log_cap = t->log_cap;
    LPR_HIP(hipMemcpyAsync(&c->state->log_cap, &c->h_state->log_cap, sizeof(int64_t),
                           hipMemcpyHostToDevice, t->eng->stream));
    return LPR_OK_OPTIMAL;
}

// reads the control block back; returns status / pending / iter through the pointers
int blk_poll(lpr_tableau* t, int32_t* status, int32_t* pending, int64_t* iter) {
    lpr_block_ctx* c = static_cast<lpr_block_ctx*>(t->blk);
    hipStream_t s = t->eng->stream;
    LPR_HIP(hipMemcpyAsync(c->h_state, c->state, sizeof(BlockState), hipMemcpyDeviceToHost, s));
    LPR_HIP(hipStreamSynchronize(s));
    *status = c->h_state->status;
    *pending = c->h_state->pending;
    *iter = c->h_state->iter;
    return LPR_OK_OPTIMAL;
}

void blk_launch_bootstrap(lpr_tableau* t) {
    lpr_block_ctx* c = static_cast<lpr_block_ctx*>(t->blk);
    hipLaunchKernelGGL(k_blk_bootstrap, dim3(1), dim3(1024), 0, t->eng->stream, t->T, t->ld,
                       t->rows, t->cols, t->next_col, t->next_rhs, c->state,
                       reinterpret_cast<ZPart*>(t->zparts), blk_groups(t));
}

// the heads of one block of K pivots
void blk_launch_heads(lpr_tableau* t, int K) {
    lpr_block_ctx* c = static_cast<lpr_block_ctx*>(t->blk);
    hipStream_t s = t->eng->stream;
    const int G = blk_groups(t);
    ZPart* zp = reinterpret_cast<ZPart*>(t->zparts);
    for (int q = 1; q <= K; ++q) {
        if (q > 1)
            hipLaunchKernelGGL(k_blk_gather, dim3((t->rows + 127) / 128), dim3(128), 0, s, t->T,
                               t->ld, t->rows, q, c->prow, c->fcol, c->Rp, c->state, zp, G);
        hipLaunchKernelGGL(k_blk_head, dim3(G), dim3(kBlkHeadNT), 0, s, t->T, t->ld, t->rows,
                           t->cols, q, c->prow, c->fcol, c->Rp, t->next_col, t->next_rhs, c->zrow,
                           c->bvec, t->basis, t->log, c->state, zp, G);
    }
}

void blk_launch_update(lpr_tableau* t, int tr) {
    lpr_block_ctx* c = static_cast<lpr_block_ctx*>(t->blk);
    hipStream_t s = t->eng->stream;
    const int G = blk_groups(t);
    const ZPart* zp = reinterpret_cast<const ZPart*>(t->zparts);
    const int ld2 = t->ld / 2;
    if (tr >= 32) {
        dim3 grid((ld2 + 255) / 256, (t->rows + 31) / 32);
        hipLaunchKernelGGL((k_blk_update<32>), grid, dim3(256), 0, s, t->T, t->ld, t->rows, t->cols,
                           c->prow, c->fcol, c->Rp, t->next_col, t->next_rhs, c->state, zp, G);
    } else if (tr >= 16) {
        dim3 grid((ld2 + 255) / 256, (t->rows + 15) / 16);
        hipLaunchKernelGGL((k_blk_update<16>), grid, dim3(256), 0, s, t->T, t->ld, t->rows, t->cols,
                           c->prow, c->fcol, c->Rp, t->next_col, t->next_rhs, c->state, zp, G);
    } else {
        dim3 grid((ld2 + 255) / 256, (t->rows + 7) / 8);
        hipLaunchKernelGGL((k_blk_update<8>), grid, dim3(256), 0, s, t->T, t->ld, t->rows, t->cols,
                           c->prow, c->fcol, c->Rp, t->next_col, t->next_rhs, c->state, zp, G);
    }
}

}  // namespace lpr
