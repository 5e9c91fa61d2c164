// primal_kernels.hip -- gfx950 kernels of the full-tableau primal simplex pivot
// (reference: LPR_381_Group_V22/Simplex/PrimalSimplexSolver.cs).
//
// Data layout in HBM: the (m+1) x (n+m+1) fp64 tableau is row-major like the C# double[,], with
// the leading dimension padded to 16 doubles so every row starts on a 128-byte line and the
// streaming kernel can use 16 B/lane loads and stores.  Padding columns are kept at 0.
//
//   k_select  one workgroup: FindEnteringVariable (:152-167) as a wave64 shuffle arg-min,
//             FindLeavingVariable (:169-191) over the two strided columns, then the pivot-row
//             normalise (:198-199) into a dense scratch row and the pivot column into a dense
//             scratch column, basis + pivot-log update (:142).  Sets the device status word.
//   k_update  the roofline kernel: rank-1 row elimination (:202-210), every element read once
//             and written once (2*8*R*C algorithmic bytes), HBM-bound.
//   k_extract ExtractSolution (:213-252), one lane per decision column.
//   k_build*  the constructor (:27-87) and the synthetic benchmark LP.
#include "engine_common.hpp"
#include "select_common.hpp"

#pragma clang fp contract(off)

namespace lpr {

enum : int { kSelEnter = 1, kSelLeave = 2, kSelCommit = 4, kSelFull = 7 };

// One workgroup of 1024 threads.  In kSelFull mode this is "one C# loop head": it either ends the
// solve (status word) or leaves (cur_r, cur_e, rowbuf, colbuf) ready for k_update.
__global__ __launch_bounds__(1024) void k_select(double* __restrict__ T, int ld, int R, int C,
                                                 double* __restrict__ rowbuf,
                                                 double* __restrict__ colbuf,
                                                 int32_t* __restrict__ basis,
                                                 int32_t* __restrict__ log, PivotState* st,
                                                 int mode, int e_in, int r_in,
                                                 int32_t* __restrict__ out_i) {
    __shared__ double lds_v[16];
    __shared__ int lds_i[16];
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    const bool full = (mode == kSelFull);

    if (full && st->status != kRunning) return;

    int e = e_in;
    int r = r_in;

    // ---- FindEnteringVariable  PrimalSimplexSolver.cs:152-167 ----
    if (mode & kSelEnter) {
        Cand c;
        c.v = 0.0;  // mostNegative = 0
        c.i = -1;
        // 4 independent loads in flight per lane; candidates are folded in ascending j
        for (int j0 = tid; j0 < C - 1; j0 += 4 * nt) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * nt;
                v[u] = (j < C - 1) ? T[j] : 0.0;  // 0.0 is never < mostNegative
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (v[u] < c.v) {  // strict: ties keep the lower index, NaN and -0 never enter
                    c.v = v[u];
                    c.i = j0 + u * nt;
                }
            }
        }
        c = block_cand_min(c, lds_v, lds_i);
        e = c.i;
        if (tid == 0 && out_i) out_i[0] = e;
        if (e < 0) {
            if (full && tid == 0) st->status = LPR_OK_OPTIMAL;  // :110-126
            return;
        }
    }

    // ---- FindLeavingVariable  PrimalSimplexSolver.cs:169-191 ----
    // The same pass copies column e (all rows, row 0 included) into colbuf: the factors
    // `tableau[i, pivotCol]` that Pivot reads before it overwrites the row (:206).
    const bool gather_in_leave = (mode & kSelLeave) != 0;
    if (mode & kSelLeave) {
        Cand c;
        c.v = DBL_MAX;  // double.MaxValue
        c.i = -1;
        const int rhs = C - 1;
        for (int i0 = tid; i0 < R; i0 += 4 * nt) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // 8 strided loads in flight per lane
                const int i = i0 + u * nt;
                const bool in = i < R;
                a[u] = in ? T[(size_t)i * ld + e] : 0.0;
                b[u] = in ? T[(size_t)i * ld + rhs] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * nt;
                if (i < R) {
                    colbuf[i] = a[u];
                    if (i >= 1 && a[u] > 1e-9) {
                        const double ratio = ieee_div(b[u], a[u]);  // IEEE division
                        if (ratio >= 0 && ratio < c.v) {
                            c.v = ratio;
                            c.i = i;
                        }
                    }
                }
            }
        }
        c = block_cand_min(c, lds_v, lds_i);
        r = c.i;
        if (tid == 0 && out_i) out_i[1] = r;
        if (r < 0) {
            if (full && tid == 0) st->status = LPR_UNBOUNDED;  // :129-135
            return;
        }
    }

    if (!(mode & kSelCommit)) return;

    if (full) {
        const int64_t it = st->iter;
        const int64_t mx = st->max_iter;
        if (mx > 0 && it >= mx) {
            if (tid == 0) st->status = LPR_PIVOT_LIMIT;
            return;
        }
    }

    // ---- Pivot, first half  PrimalSimplexSolver.cs:195-199 ----
    // rowbuf[j] = T[r, j] / T[r, e] (true division).  The row itself is rewritten by k_update.
    const double p = T[(size_t)r * ld + e];
    const double* prow = T + (size_t)r * ld;
    for (int j0 = tid; j0 < ld; j0 += 4 * nt) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * nt;
            v[u] = (j < C) ? prow[j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * nt;
            if (j < ld) rowbuf[j] = (j < C) ? ieee_div(v[u], p) : 0.0;
        }
    }
    if (!gather_in_leave) {
        for (int i = tid; i < R; i += nt) colbuf[i] = T[(size_t)i * ld + e];
    }

    if (tid == 0) {
        st->cur_r = r;
        st->cur_e = e;
        basis[r - 1] = e;  // :142
        const int64_t it = st->iter;
        if (it < st->log_cap) {
            log[2 * it] = r;
            log[2 * it + 1] = e;
        }
        st->iter = it + 1;  // :138 ++iteration
        st->sweep ^= 1;
    }
}

// ------------------------------------------------------------------------------------------
// k_bootstrap: primes the pipelined loop head.  FindEnteringVariable (:152-167) on the current Z
// row -> partial 0 of the current bank, and the strided gather of that column and of the RHS column
// into the dense next_col / next_rhs vectors.  Runs once per lpr_primal_solve call; afterwards
// k_update keeps those vectors current as a by-product of its sweep.
__global__ __launch_bounds__(1024) void k_bootstrap(const double* __restrict__ T, int ld, int R,
                                                    int C, double* __restrict__ next_col,
                                                    double* __restrict__ next_rhs,
                                                    PivotState* st, ZPart* __restrict__ zparts,
                                                    int G) {
    __shared__ double lds_v[16];
    __shared__ int lds_i[16];
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    Cand c;
    c.v = 0.0;
    c.i = -1;
    for (int j0 = tid; j0 < C - 1; j0 += 4 * nt) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * nt;
            v[u] = (j < C - 1) ? T[j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (v[u] < c.v) {
                c.v = v[u];
                c.i = j0 + u * nt;
            }
        }
    }
    c = block_cand_min(c, lds_v, lds_i);
    const int e = c.i;
    ZPart* bank = zparts + (st->iter & 1) * kMaxHeadGroups;
    if (tid < G) {
        bank[tid].v = (tid == 0) ? c.v : 0.0;
        bank[tid].i = (tid == 0) ? e : -1;
    }
    if (tid == 0) st->iter_pending = st->iter;
    if (e < 0) return;
    const int rhs = C - 1;
    for (int i0 = tid; i0 < R; i0 += 4 * nt) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * nt;
            const bool in = i < R;
            a[u] = in ? T[(size_t)i * ld + e] : 0.0;
            b[u] = in ? T[(size_t)i * ld + rhs] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * nt;
            if (i < R) {
                next_col[i] = a[u];
                next_rhs[i] = b[u];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_pivot_head: one C# loop head (Solve :107-142), G workgroups, two memory round trips each.
//   1. e = arg-min over the G partials of the current bank (-1 -> optimal, :110-126).
//   2. FindLeavingVariable (:169-191) over the DENSE next_col / next_rhs vectors that the previous
//      k_update wrote while it streamed the tableau (coalesced, no strided gather).  Every
//      workgroup does this 64 KB scan itself -- cheaper than any cross-workgroup hand-off -- and
//      all arrive at the same r.  next_col is also this pivot's factor column (colbuf, copied by
//      workgroup 0).
//   3. Pivot, first half (:195-199), split by columns over the workgroups (one CU can only pull
//      ~50 GB/s, the row is 98 KB): rowbuf = T[r, :] / T[r, e]; at the same time the Z row of the
//      tableau AFTER this pivot is formed in registers, z'[j] = T[0, j] - (f0 * rowbuf[j]) -- the
//      very expression k_update will store -- and FindEnteringVariable (:152-167) runs on it; each
//      workgroup publishes its partial arg-min, k_update and the next head reduce the G partials.
//      So the NEXT entering column is known before this pivot's update starts and k_update can
//      dump that column on the fly.
// Workgroup 0 alone writes the control block; the pivot counter itself is committed by k_update
// (st->iter is read by every head workgroup and must not change under them).
__global__ __launch_bounds__(1024) void k_pivot_head(const double* __restrict__ T, int ld, int R,
                                                     int C, double* __restrict__ rowbuf,
                                                     double* __restrict__ colbuf,
                                                     const double* __restrict__ next_col,
                                                     const double* __restrict__ next_rhs,
                                                     int32_t* __restrict__ basis,
                                                     int32_t* __restrict__ log, PivotState* st,
                                                     ZPart* __restrict__ zparts, int G) {
    __shared__ double lds_v[16];
    __shared__ int lds_i[16];
    __shared__ double lds_p[2];  // [0] = T[r, e], [1] = T[0, e]
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    const int g = blockIdx.x;
    const bool lead = (g == 0);

    // Everything whose ADDRESS is known up front is requested before the first branch, so the
    // control block, both partial banks, this lane's share of the dense vectors and its Z-row
    // slice come back in ONE memory round trip (the caches were just flushed by the previous
    // kernel's end: each dependent trip costs ~2 us here).
    const int ld2 = ld >> 1;
    const double2* __restrict__ zrow2 = reinterpret_cast<const double2*>(T);
    double2* __restrict__ out2 = reinterpret_cast<double2*>(rowbuf);
    const int stride2 = G * nt;          // double2 chunks covered per trip by all workgroups
    const int c2_first = g * nt + tid;   // this lane's chunk in the first trip
    const int32_t status = st->status;
    const int64_t it = st->iter;
    const int64_t mx = st->max_iter;
    Cand e0 = reduce_zparts(zparts, G);                   // bank 0
    Cand e1 = reduce_zparts(zparts + kMaxHeadGroups, G);  // bank 1
    double a0[8], b0[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = tid + u * nt;
        const bool in = i < R;
        a0[u] = in ? next_col[i] : 0.0;
        b0[u] = in ? next_rhs[i] : 0.0;
    }
    double2 zv = (c2_first < ld2) ? zrow2[c2_first] : make_double2(0.0, 0.0);

    if (status != kRunning) return;
    ZPart* __restrict__ bank_out = zparts + ((it + 1) & 1) * kMaxHeadGroups;
    const int e = ((it & 1) ? e1 : e0).i;
    if (e < 0) {
        if (lead && tid == 0) st->status = LPR_OK_OPTIMAL;
        return;
    }

    // ---- FindLeavingVariable on the dense vectors ----
    Cand c;
    c.v = DBL_MAX;
    c.i = -1;
    double a_of_best = 0.0;  // next_col[c.i], carried so that p needs no second global read
    for (int i0 = tid; i0 < R; i0 += 8 * nt) {
        double a[8], b[8];
        if (i0 == tid) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = a0[u];
                b[u] = b0[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * nt;
                const bool in = i < R;
                a[u] = in ? next_col[i] : 0.0;
                b[u] = in ? next_rhs[i] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * nt;
            if (i < R) {
                if (lead) colbuf[i] = a[u];
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
        if (lead && tid == 0) st->status = LPR_UNBOUNDED;
        return;
    }
    if (mx > 0 && it >= mx) {
        if (lead && tid == 0) st->status = LPR_PIVOT_LIMIT;
        return;
    }
    if (my_best == r) lds_p[0] = a_of_best;  // exactly one lane owns row r
    __syncthreads();

    // ---- normalise this workgroup's slice of row r, form the next Z row, partial arg-min ----
    const double p = lds_p[0];   // T[r, e]
    const double f0 = lds_p[1];  // T[0, e]
    const double2* __restrict__ prow2 = reinterpret_cast<const double2*>(T + (size_t)r * ld);
    Cand n;
    n.v = 0.0;
    n.i = -1;
    for (int c2 = c2_first; c2 < ld2; c2 += stride2) {  // one trip unless ld > 2048 * G
        const double2 pv = prow2[c2];
        if (c2 != c2_first) zv = zrow2[c2];
        const int j = 2 * c2;
        double2 q;
        q.x = (j < C) ? ieee_div(pv.x, p) : 0.0;      // :199 true division
        q.y = (j + 1 < C) ? ieee_div(pv.y, p) : 0.0;
        out2[c2] = q;
        const double mx = f0 * q.x;          // :208 product rounded ...
        const double my = f0 * q.y;
        const double zx = zv.x - mx;         // ... then the difference
        const double zy = zv.y - my;
        if (j < C - 1 && zx < n.v) {
            n.v = zx;
            n.i = j;
        }
        if (j + 1 < C - 1 && zy < n.v) {
            n.v = zy;
            n.i = j + 1;
        }
    }
    n = block_cand_min(n, lds_v, lds_i);

    if (tid == 0) {
        bank_out[g].v = n.v;
        bank_out[g].i = n.i;
        if (lead) {
            st->cur_r = r;
            st->cur_e = e;
            basis[r - 1] = e;  // :142
            if (it < st->log_cap) {
                log[2 * it] = r;
                log[2 * it + 1] = e;
            }
            st->iter_pending = it + 1;  // :138, committed to st->iter by k_update
            st->sweep ^= 1;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Rank-1 row elimination  PrimalSimplexSolver.cs:202-210
//   T[i, j] = T[i, j] - (f_i * prow[j])   for i != r      (product rounded, then difference)
//   T[r, j] = prow[j]                                       (the normalised row, :199)
// Tile = TR rows x (256*VPT) double2 columns per 256-thread workgroup.  Each lane keeps its slice
// of the normalised pivot row in registers for the whole tile, the per-row factor f_i is a
// wave-uniform scalar load, and all TR*VPT 16-byte loads of a lane are issued before the first
// store so a wave has TR*VPT KiB in flight.
typedef double v2d_t __attribute__((ext_vector_type(2)));

// NT bit 0: non-temporal loads of the tableau, bit 1: non-temporal stores (tuning variants)
template <int NT>
__device__ __forceinline__ double2 ld_tab(const double2* p) {
    if (NT & 1) {
        const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(p));
        return make_double2(v.x, v.y);
    }
    return *p;
}
template <int NT>
__device__ __forceinline__ void st_tab(double2* p, double2 o) {
    if (NT & 2) {
        v2d_t v;
        v.x = o.x;
        v.y = o.y;
        __builtin_nontemporal_store(v, reinterpret_cast<v2d_t*>(p));
    } else {
        *p = o;
    }
}

template <int TR, int VPT, bool FULL, int NT>
__device__ __forceinline__ void update_tile(double2* __restrict__ T2, int ld2, int R, int r,
                                            const double2* __restrict__ prow2,
                                            const double* __restrict__ colbuf, int i0,
                                            int c2base, const ZPart* __restrict__ zbank, int G,
                                            int rhs, double* __restrict__ next_col,
                                            double* __restrict__ next_rhs) {
    double2 pr[VPT];
    bool ok[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int c2 = c2base + v * 256;
        ok[v] = FULL || c2 < ld2;
        pr[v] = ok[v] ? prow2[c2] : make_double2(0.0, 0.0);
    }
    double2 x[TR][VPT];
#pragma unroll
    for (int k = 0; k < TR; ++k) {
        const int i = i0 + k;
        if (FULL || i < R) {
#pragma unroll
            for (int v = 0; v < VPT; ++v)
                if (ok[v]) x[k][v] = ld_tab<NT>(&T2[(size_t)i * ld2 + c2base + v * 256]);
        }
    }
    // While the tile is in flight: which lanes own the next entering column / the RHS column?
    // They copy their new values into the dense vectors read by the next k_pivot_head
    // (zbank == nullptr: single-step pivot, nothing to dump).
    int dump_e = -1, dump_rhs = -1, ne = -1;
    if (zbank) {
        ne = reduce_zparts(zbank, G).i;
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int c2 = c2base + v * 256;
            if (ne >= 0 && c2 == (ne >> 1)) dump_e = v;
            if (ne >= 0 && c2 == (rhs >> 1)) dump_rhs = v;
        }
    }
#pragma unroll
    for (int k = 0; k < TR; ++k) {
        const int i = i0 + k;
        if (FULL || i < R) {
            const double f = colbuf[i];
            const bool is_r = (i == r);
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                if (ok[v]) {
                    double2 o;
                    const double px = f * pr[v].x;  // product rounded ...
                    const double py = f * pr[v].y;
                    o.x = x[k][v].x - px;           // ... then the difference (:208)
                    o.y = x[k][v].y - py;
                    if (is_r) o = pr[v];            // the pivot row keeps the normalised values
                    st_tab<NT>(&T2[(size_t)i * ld2 + c2base + v * 256], o);
                    if (v == dump_e) next_col[i] = (ne & 1) ? o.y : o.x;
                    if (v == dump_rhs) next_rhs[i] = (rhs & 1) ? o.y : o.x;
                }
            }
        }
    }
}

template <int TR, int VPT, int NT>
__global__ __launch_bounds__(256) void k_update(double* __restrict__ T, int ld, int R, int C,
                                                const double* __restrict__ rowbuf,
                                                const double* __restrict__ colbuf,
                                                double* __restrict__ next_col,
                                                double* __restrict__ next_rhs, PivotState* st,
                                                const ZPart* __restrict__ zparts, int G,
                                                int check_status, int serpentine, int dump_next) {
    if (check_status && st->status != kRunning) return;
    const int r = st->cur_r;
    const int64_t itp = st->iter_pending;
    if (dump_next && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        st->iter = itp;  // commit the pivot counter (no k_update workgroup reads st->iter)
    const ZPart* zbank = dump_next ? zparts + (itp & 1) * kMaxHeadGroups : nullptr;
    const int ld2 = ld >> 1;
    int ct = blockIdx.x, rt = blockIdx.y;
    if (serpentine && (st->sweep & 1)) {  // reverse the sweep on alternate pivots
        ct = gridDim.x - 1 - ct;
        rt = gridDim.y - 1 - rt;
    }
    const int c2base = ct * (256 * VPT) + threadIdx.x;
    const double2* __restrict__ prow2 = reinterpret_cast<const double2*>(rowbuf);
    double2* __restrict__ T2 = reinterpret_cast<double2*>(T);
    const int i0 = rt * TR;
    // interior tiles (the common case) run without any per-element guard
    const bool full = (i0 + TR <= R) && ((ct + 1) * (256 * VPT) <= ld2);
    if (full)
        update_tile<TR, VPT, true, NT>(T2, ld2, R, r, prow2, colbuf, i0, c2base, zbank, G, C - 1,
                                   next_col, next_rhs);
    else
        update_tile<TR, VPT, false, NT>(T2, ld2, R, r, prow2, colbuf, i0, c2base, zbank, G, C - 1,
                                    next_col, next_rhs);
}

// ------------------------------------------------------------------------------------------
// k_pivot_fused: one whole pivot (Solve :107-142 + Pivot :193-211) in ONE launch, for tableaux
// small enough to live in L2 / Infinity Cache, where the cost of a pivot is launch latency, not
// bytes.  Out of place (Tin -> Tout, the two buffers alternate) so that no workgroup can overwrite
// what another still has to read; therefore EVERY workgroup can afford to redo the selection
// itself -- Z-row arg-min (:152-167), ratio test over the two strided columns (:169-191) -- from
// the read-only input (a few KB out of L2), and then updates its own TR x 512 tile with the
// normalised pivot row formed on the fly (:199, :208).  Workgroup (0,0) alone records the pivot.
template <int TR>
__global__ __launch_bounds__(256) void k_pivot_fused(const double* __restrict__ Tin,
                                                     double* __restrict__ Tout, int ld, int R,
                                                     int C, int32_t* __restrict__ basis,
                                                     int32_t* __restrict__ log, PivotState* st) {
    __shared__ double lds_v[16];
    __shared__ int lds_i[16];
    if (st->status != kRunning) return;
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    const bool lead = (blockIdx.x == 0 && blockIdx.y == 0);

    Cand c;  // FindEnteringVariable
    c.v = 0.0;
    c.i = -1;
    for (int j = tid; j < C - 1; j += nt) {
        const double v = Tin[j];
        if (v < c.v) {
            c.v = v;
            c.i = j;
        }
    }
    c = block_cand_min(c, lds_v, lds_i);
    const int e = c.i;
    if (e < 0) {
        if (lead && tid == 0) st->status = LPR_OK_OPTIMAL;
        return;
    }
    Cand q;  // FindLeavingVariable
    q.v = DBL_MAX;
    q.i = -1;
    const int rhs = C - 1;
    for (int i = 1 + tid; i < R; i += nt) {
        const double a = Tin[(size_t)i * ld + e];
        if (a > 1e-9) {
            const double ratio = ieee_div(Tin[(size_t)i * ld + rhs], a);
            if (ratio >= 0 && ratio < q.v) {
                q.v = ratio;
                q.i = i;
            }
        }
    }
    q = block_cand_min(q, lds_v, lds_i);
    const int r = q.i;
    if (r < 0) {
        if (lead && tid == 0) st->status = LPR_UNBOUNDED;
        return;
    }

    const int ld2 = ld >> 1;
    const int c2 = blockIdx.x * nt + tid;
    const int i0 = blockIdx.y * TR;
    if (c2 < ld2) {
        const double p = Tin[(size_t)r * ld + e];
        const double2* __restrict__ in2 = reinterpret_cast<const double2*>(Tin);
        double2* __restrict__ out2 = reinterpret_cast<double2*>(Tout);
        const double2 pv = in2[(size_t)r * ld2 + c2];
        double2 pr;
        pr.x = (2 * c2 < C) ? ieee_div(pv.x, p) : 0.0;      // :199 true division
        pr.y = (2 * c2 + 1 < C) ? ieee_div(pv.y, p) : 0.0;
        double2 x[TR];
        double f[TR];
#pragma unroll
        for (int k = 0; k < TR; ++k) {
            const int i = i0 + k;
            if (i < R) {
                x[k] = in2[(size_t)i * ld2 + c2];
                f[k] = Tin[(size_t)i * ld + e];  // :206, wave-uniform
            }
        }
#pragma unroll
        for (int k = 0; k < TR; ++k) {
            const int i = i0 + k;
            if (i < R) {
                double2 o;
                const double px = f[k] * pr.x;  // product rounded ...
                const double py = f[k] * pr.y;
                o.x = x[k].x - px;              // ... then the difference (:208)
                o.y = x[k].y - py;
                if (i == r) o = pr;
                out2[(size_t)i * ld2 + c2] = o;
            }
        }
    }
    if (lead && tid == 0) {
        const int64_t it = st->iter;
        basis[r - 1] = e;  // :142
        if (it < st->log_cap) {
            log[2 * it] = r;
            log[2 * it + 1] = e;
        }
        st->iter = it + 1;  // :138 (no other workgroup reads the counter)
        st->cur_r = r;
        st->cur_e = e;
    }
}

// ------------------------------------------------------------------------------------------
// ExtractSolution  PrimalSimplexSolver.cs:213-252.  One lane per decision column j < n (coalesced
// across j), serial over rows.  The C# early `break`s only stop the scan; the verdict
// "exactly one entry within 1e-9 of 1 and every other entry within 1e-9 of 0" does not depend on
// where the scan stops, so the full scan gives the same x_j.
__global__ __launch_bounds__(256) void k_extract(const double* __restrict__ T, int ld, int R,
                                                 int C, int n, double* __restrict__ x) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int basicRow = -1;
    bool isBasic = true;
    for (int i = 1; i < R; ++i) {
        const double v = T[(size_t)i * ld + j];
        if (fabs(v - 1.0) < 1e-9) {
            if (basicRow == -1) basicRow = i;
            else isBasic = false;
        } else if (fabs(v) > 1e-9) {
            isBasic = false;
        }
    }
    x[j] = (isBasic && basicRow != -1) ? T[(size_t)basicRow * ld + (C - 1)] : 0.0;
}

// ------------------------------------------------------------------------------------------
// Constructor  PrimalSimplexSolver.cs:27-87 (T is zero-filled before this kernel runs).
__global__ __launch_bounds__(256) void k_build_rows(double* __restrict__ T, int ld, int n, int m,
                                                    const double* __restrict__ A, int lda,
                                                    const int32_t* __restrict__ ncoef,
                                                    const int8_t* __restrict__ rel,
                                                    const double* __restrict__ rhs,
                                                    int32_t* __restrict__ basis) {
    const int i = blockIdx.y;  // constraint row
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const bool ge = rel && rel[i] == LPR_REL_GE;  // :36-41
    const int cnt = ncoef ? ncoef[i] : n;
    double* row = T + (size_t)(i + 1) * ld;
    if (j < n && j < cnt) {  // :68-72
        const double a = A[(size_t)i * lda + j];
        row[j] = ge ? -a : a;
    }
    if (j == 0) {
        const int C = n + m + 1;
        row[n + i] = 1.0;                   // :75-76
        basis[i] = n + i;                   // :78
        row[C - 1] = ge ? -rhs[i] : rhs[i]; // :82
    }
}

__global__ __launch_bounds__(256) void k_build_obj(double* __restrict__ T, int n,
                                                   const double* __restrict__ obj, int is_max) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) T[j] = is_max ? -obj[j] : obj[j];  // :61-62
}

// Synthetic dense LP (DESIGN.md "benchmark input"): same SplitMix64-keyed function as
// oracle/oracle_primal.c:orc_u01, evaluated per element on the device.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ double u01(uint64_t seed, uint64_t stream, uint64_t i, uint64_t j) {
    uint64_t k = splitmix64(seed ^ (stream * 0xD1B54A32D192ED03ULL));
    k = splitmix64(k + i);
    k = splitmix64(k + j);
    return (double)(k >> 11) * 0x1.0p-53;
}

__global__ __launch_bounds__(256) void k_synthetic(double* __restrict__ T, int ld, int m, int n,
                                                   uint64_t seed, int32_t* __restrict__ basis) {
    const int i = blockIdx.y;  // tableau row
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld) return;
    const int C = n + m + 1;
    double v = 0.0;
    if (i == 0) {
        if (j < n) v = -u01(seed, 2, 0, (uint64_t)j);
    } else {
        const int ci = i - 1;
        if (j < n) v = u01(seed, 0, (uint64_t)ci, (uint64_t)j);
        else if (j == n + ci) v = 1.0;
        else if (j == C - 1) {
            const double u = u01(seed, 1, (uint64_t)ci, 0);
            const double s = u * 0.1;
            const double t = 1.0 + s;
            v = ((double)n * 0.25) * t;  // b_i = (n/4) * (1 + 0.1 U)
        }
        if (j == 0) basis[ci] = n + ci;
    }
    T[(size_t)i * ld + j] = v;
}

// ------------------------------------------------------------------------------------------
// Host-side launchers (called from lpr_engine.hip).

void launch_select(lpr_tableau* t, int mode, int e_in, int r_in, int32_t* out_i) {
    hipLaunchKernelGGL(k_select, dim3(1), dim3(1024), 0, t->eng->stream, t->T, t->ld, t->rows,
                       t->cols, t->rowbuf, t->colbuf, t->basis, t->log, t->state, mode, e_in,
                       r_in, out_i);
}

// number of k_pivot_head workgroups: one per 1024 double2 of the row, capped
int head_groups(const lpr_tableau* t) {
    int g = (t->ld / 2 + 1023) / 1024;
    if (g < 1) g = 1;
    if (g > kMaxHeadGroups) g = kMaxHeadGroups;
    return g;
}

void launch_bootstrap(lpr_tableau* t) {
    hipLaunchKernelGGL(k_bootstrap, dim3(1), dim3(1024), 0, t->eng->stream, t->T, t->ld, t->rows,
                       t->cols, t->next_col, t->next_rhs, t->state,
                       reinterpret_cast<ZPart*>(t->zparts), head_groups(t));
}

void launch_pivot_head(lpr_tableau* t) {
    const int G = head_groups(t);
    hipLaunchKernelGGL(k_pivot_head, dim3(G), dim3(1024), 0, t->eng->stream, t->T, t->ld, t->rows,
                       t->cols, t->rowbuf, t->colbuf, t->next_col, t->next_rhs, t->basis, t->log,
                       t->state, reinterpret_cast<ZPart*>(t->zparts), G);
}

template <int TR, int VPT, int NT = 0>
static void launch_update_t(lpr_tableau* t, int check_status, int serpentine, int dump_next) {
    const int ld2 = t->ld / 2;
    dim3 grid((ld2 + 256 * VPT - 1) / (256 * VPT), (t->rows + TR - 1) / TR);
    hipLaunchKernelGGL((k_update<TR, VPT, NT>), grid, dim3(256), 0, t->eng->stream, t->T, t->ld,
                       t->rows, t->cols, t->rowbuf, t->colbuf, t->next_col, t->next_rhs, t->state,
                       reinterpret_cast<const ZPart*>(t->zparts), head_groups(t), check_status,
                       serpentine, dump_next);
}

// variant: low byte selects the tile shape, bit 8 turns the serpentine sweep on.
int num_update_variants() { return 11; }

void launch_update(lpr_tableau* t, int variant, int check_status, int dump_next) {
    const int serp = (variant >> 8) & 1;
    const int nt = (variant >> 9) & 3;  // tuning: non-temporal loads (bit 9) / stores (bit 10)
    if (nt && (variant & 0xff) == 5) {
        if (nt == 1) launch_update_t<32, 1, 1>(t, check_status, serp, dump_next);
        else if (nt == 2) launch_update_t<32, 1, 2>(t, check_status, serp, dump_next);
        else launch_update_t<32, 1, 3>(t, check_status, serp, dump_next);
        return;
    }
    switch (variant & 0xff) {
        default:
        case 0: launch_update_t<16, 1>(t, check_status, serp, dump_next); break;
        case 1: launch_update_t<8, 1>(t, check_status, serp, dump_next); break;
        case 2: launch_update_t<8, 2>(t, check_status, serp, dump_next); break;
        case 3: launch_update_t<4, 2>(t, check_status, serp, dump_next); break;
        case 4: launch_update_t<4, 4>(t, check_status, serp, dump_next); break;
        case 5: launch_update_t<32, 1>(t, check_status, serp, dump_next); break;
        case 6: launch_update_t<16, 2>(t, check_status, serp, dump_next); break;
        case 7: launch_update_t<2, 4>(t, check_status, serp, dump_next); break;
        case 8: launch_update_t<4, 1>(t, check_status, serp, dump_next); break;
        case 9: launch_update_t<2, 1>(t, check_status, serp, dump_next); break;
        case 10: launch_update_t<1, 1>(t, check_status, serp, dump_next); break;
    }
}

// one fused pivot, in -> out (both rows x ld)
void launch_pivot_fused(lpr_tableau* t, const double* in, double* out) {
    const int ld2 = t->ld / 2;
    const int rows = t->rows;
    // tallest row tile that still yields >= 256 workgroups
    const int ct = (ld2 + 255) / 256;
    int tr = 8;
    while (tr > 1 && ct * ((rows + tr - 1) / tr) < 256) tr >>= 1;
    dim3 grid(ct, (rows + tr - 1) / tr);
    hipStream_t s = t->eng->stream;
    switch (tr) {
        case 8: hipLaunchKernelGGL((k_pivot_fused<8>), grid, dim3(256), 0, s, in, out, t->ld, rows, t->cols, t->basis, t->log, t->state); break;
        case 4: hipLaunchKernelGGL((k_pivot_fused<4>), grid, dim3(256), 0, s, in, out, t->ld, rows, t->cols, t->basis, t->log, t->state); break;
        case 2: hipLaunchKernelGGL((k_pivot_fused<2>), grid, dim3(256), 0, s, in, out, t->ld, rows, t->cols, t->basis, t->log, t->state); break;
        default: hipLaunchKernelGGL((k_pivot_fused<1>), grid, dim3(256), 0, s, in, out, t->ld, rows, t->cols, t->basis, t->log, t->state); break;
    }
}

void launch_extract(lpr_tableau* t, int n, double* x) {
    hipLaunchKernelGGL(k_extract, dim3((n + 255) / 256), dim3(256), 0, t->eng->stream, t->T,
                       t->ld, t->rows, t->cols, n, x);
}

void launch_build(lpr_tableau* t, int n, int m, const double* d_obj, const double* d_A, int lda,
                  const int32_t* d_ncoef, const int8_t* d_rel, const double* d_rhs, int is_max) {
    hipStream_t s = t->eng->stream;
    if (n > 0)
        hipLaunchKernelGGL(k_build_obj, dim3((n + 255) / 256), dim3(256), 0, s, t->T, n, d_obj,
                           is_max);
    if (m > 0) {
        const int nx = n > 0 ? (n + 255) / 256 : 1;
        hipLaunchKernelGGL(k_build_rows, dim3(nx, m), dim3(256), 0, s, t->T, t->ld, n, m, d_A,
                           lda, d_ncoef, d_rel, d_rhs, t->basis);
    }
}

void launch_synthetic(lpr_tableau* t, int m, int n, uint64_t seed) {
    hipLaunchKernelGGL(k_synthetic, dim3((t->ld + 255) / 256, t->rows), dim3(256), 0,
                       t->eng->stream, t->T, t->ld, m, n, seed, t->basis);
}

}  // namespace lpr
