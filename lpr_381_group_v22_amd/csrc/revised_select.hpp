// revised_select.hpp -- the two selections of the revised simplex loop as workgroup-wide device
// functions, shared by the stand-alone kernels (revised_kernels.hip: k_rev_enter, k_rev_ratio) and
// by the tails of the fused kernels (revised_fused.hip), where the workgroup that finishes last
// runs them on values other workgroups of the SAME launch have just stored (SC1 = true: those
// values are read with agent-scope loads, past this CU's L1).  Not part of the ABI.
#pragma once

#include "engine_common.hpp"
#include "revised_common.hpp"
#include "fold_common.hpp"

#pragma clang fp contract(off)

namespace lpr {

constexpr double kRevEps = 1e-9;  // RevisedPrimalSimplexSolver.cs:12

template <bool SC1>
__device__ __forceinline__ double rev_ld(const double* p) {
    if constexpr (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

constexpr int kRatioLds = 4096;  // rows whose ratios one wave replays out of LDS

// ------------------------------------------------------------------------------------------
// Block-wide minimum of an int (smallest index that satisfies a predicate; INT_MAX = none).
__device__ __forceinline__ int block_min_int(int v, int* lds) {
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

// ------------------------------------------------------------------------------------------
// The entering fold from GROUP MINIMA (the fused path).  The kernels that produce the reduced
// costs leave, per group of consecutive candidates (32 structural columns of a k_rev_rc_enter
// workgroup, 16 slack columns of a k_rev_update_y workgroup), the minimum of v = -rc over the
// group's candidates (+inf: none).  A take of the C#'s fold is a strict prefix minimum of the whole
// sequence (fold_common.hpp), i.e. a strict prefix minimum WITHIN its group that is also below
// the minimum of every earlier group.  So: an exclusive prefix-min scan over the G group minima
// (G = n/32 + m/16: 512 values instead of n + m = 12 288) gives each group its threshold; only the
// ~ln G groups whose own minimum beats their threshold can hold a take; one wave reads just those
// groups' values (two groups per 64 lanes, all loads of up to 16 groups requested before the
// first is used) and replays the fold over their flagged lanes in index order.  Returns false
// (nothing decided) when there are more groups than kHierPer per thread or more than kHierAct
// active ones: the caller then runs the full fold.
constexpr int kHierPer = 4;
constexpr int kHierAct = 64;
template <bool SC1>
__device__ __forceinline__ bool rev_enter_hier(const double* __restrict__ wmin, int gr, int gy,
                                               const double* __restrict__ rcx,
                                               const double* __restrict__ y,
                                               const uint8_t* __restrict__ is_basic, int n, int m,
                                               int* cur_out, int* lds_i, double* lds_v) {
    __shared__ int a_id[kHierAct];
    __shared__ double a_thr[kHierAct];
    __shared__ int s_res;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (kWave - 1), wave = tid / kWave, nw = nt / kWave;
    const int G = gr + gy;
    if (G > kHierPer * nt) return false;  // (uniform)
    const int cpt = (G + nt - 1) / nt;
    const int g0 = tid * cpt;
    double wm[kHierPer];
#pragma unroll
    for (int k = 0; k < kHierPer; ++k) {
        const int g = g0 + k;
        const bool ok = k < cpt && g < G;
        const double t = rev_ld<SC1>(wmin + (ok ? g : 0));
        wm[k] = (ok && t == t) ? t : (double)INFINITY;
    }
    double lmin = INFINITY;
#pragma unroll
    for (int k = 0; k < kHierPer; ++k) lmin = (wm[k] < lmin) ? wm[k] : lmin;
    // exclusive prefix minimum of the per-thread minima over the block
    __syncthreads();
    double inc = lmin;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const double o = __shfl_up(inc, off, kWave);
        if (lane >= off && o < inc) inc = o;
    }
    double exc = __shfl_up(inc, 1, kWave);
    if (lane == 0) exc = INFINITY;
    if (lane == kWave - 1) lds_v[wave] = inc;
    __syncthreads();
    {
        const double wv = (lane < nw) ? lds_v[lane] : (double)INFINITY;
        for (int w = 0; w < wave; ++w) {
            const double x = readlane_f64(wv, w);
            if (x < exc) exc = x;
        }
    }
    // thresholds and active groups of this thread
    unsigned act = 0;
    double thr[kHierPer];
    double run = exc;
#pragma unroll
    for (int k = 0; k < kHierPer; ++k) {
        thr[k] = run;
        if (wm[k] < run) {
            act |= 1u << k;
            run = wm[k];
        }
    }
    // ordered compaction of the active groups
    const int cnt = __popc(act);
    int ci = cnt;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const int o = __shfl_up(ci, off, kWave);
        if (lane >= off) ci += o;
    }
    if (lane == kWave - 1) lds_i[wave] = ci;
    __syncthreads();
    int before = 0, total = 0;
    {
        const int wt = (lane < nw) ? lds_i[lane] : 0;
        for (int w = 0; w < nw; ++w) {
            const int x = __builtin_amdgcn_readlane(wt, w);
            if (w < wave) before += x;
            total += x;
        }
    }
    if (total > kHierAct) return false;  // (uniform)
    {
        int pos = before + ci - cnt;
#pragma unroll
        for (int k = 0; k < kHierPer; ++k) {
            if ((act >> k) & 1u) {
                a_id[pos] = g0 + k;
                a_thr[pos] = thr[k];
                ++pos;
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
        const int half = lane >> 5, li = lane & 31;
        double best = INFINITY;
        int cur = -1;
        constexpr int NB = 8;  // pairs of groups whose loads are in flight together
        for (int b0 = 0; b0 < total; b0 += 2 * NB) {
            double vv[NB], tt[NB];
            int kk[NB];
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                const int a = b0 + 2 * p + half;
                const bool have = a < total;
                const int g = have ? a_id[a] : 0;
                const bool xs = g < gr;
                const int k = xs ? g * 32 + li : n + (g - gr) * 16 + li;
                const bool ok = have && (xs ? k < n : (li < 16 && k < n + m));
                const int kc = ok ? k : 0;
                // one address, one load, selects afterwards (as cand_value of the full fold)
                const double t = rev_ld<SC1>(kc < n ? rcx + kc : y + (kc - n));
                const double rc = (kc < n) ? t : -t;  // rcS_k = -y_k (:100-102)
                const bool cand = ok && is_basic[kc] == 0 && rc > kRevEps;
                vv[p] = cand ? -rc : (double)NAN;
                tt[p] = have ? a_thr[a] : 0.0;
                kk[p] = k;
            }
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                if (b0 + 2 * p >= total) break;  // (uniform)
                // strict prefix minimum within the group (a 32-lane half), below its threshold
                double pin = (vv[p] == vv[p]) ? vv[p] : (double)INFINITY;
#pragma unroll
                for (int off = 1; off < 32; off <<= 1) {
                    const double o = __shfl_up(pin, off, 32);
                    if (li >= off && o < pin) pin = o;
                }
                double pex = __shfl_up(pin, 1, 32);
                if (li == 0) pex = INFINITY;
                const bool flag = vv[p] < pex && vv[p] < tt[p];
                const double x = flag ? vv[p] : (double)NAN;
                unsigned long long alive = ~0ull;
                for (;;) {
                    const unsigned long long hit = __ballot(x < best - kFoldEps) & alive;
                    if (hit == 0ull) break;
                    const int fl = __builtin_amdgcn_readfirstlane(__builtin_ctzll(hit));
                    best = readlane_f64(x, fl);
                    cur = __builtin_amdgcn_readlane(kk[p], fl);
                    alive = (fl == kWave - 1) ? 0ull : (~0ull << (fl + 1));
                }
            }
        }
        if (lane == 0) s_res = cur;
    }
    __syncthreads();
    *cur_out = s_res;
    return true;
}

// ------------------------------------------------------------------------------------------
// Entering variable, RevisedPrimalSimplexSolver.cs:105-121:
//     foreach vIdx in nonBasic ascending:  rc > EPS  and
//         (none yet  or  rc > best + EPS  or  (|rc - best| <= EPS and vIdx < enteringIdx))  -> take
// The third clause can never fire while indices are visited in ascending order.  The comparator is
// an EPS-band rule and not associative, so it is not reduced as a tree: the fold is replayed
// exactly by repeatedly searching, in parallel, for the FIRST index after the current one at
// which the C# would replace its running best ("next take"), until there is none.  The number of
// rounds is the number of replacements the sequential loop makes (O(log N) on random data).
// K: candidates per thread the fold keeps in registers (K * blockDim >= n + m for the fast path).
// At: A transposed (n x ldt), so that GetColumn(A, e) is the contiguous row e of At.
// s_val: optional LDS scratch of s_cap doubles (the fused kernel lends its ring).  When n + m fits,
// the candidate values are first staged there with COALESCED loads (thread t takes t, t + nt, ...);
// the fold's own access pattern -- a contiguous chunk per thread -- then walks LDS instead of
// pulling 64 different cache lines per wave-instruction through one CU's L1 (12.8 us -> ~3).
template <bool SC1, int K>
__device__ __forceinline__ void rev_enter_body(const double* __restrict__ rcx,
                                               const double* __restrict__ y,
                                               const uint8_t* __restrict__ is_basic, int n, int m,
                                               RevState* st, const double* __restrict__ At, int ldt,
                                               const double* __restrict__ Binv, int ldb,
                                               double* __restrict__ acol, double* __restrict__ u,
                                               unsigned long long* dbg = nullptr,
                                               double* s_val = nullptr, int s_cap = 0,
                                               const double* __restrict__ wmin = nullptr,
                                               int gr = 0, int gy = 0) {
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    // (the feasibility test of :90-91, "optimal" and the pivot limit are decided in k_rev_ratio,
    // once x_B -- computed together with u in one pass over B^-1 -- is there; the entering choice
    // itself does not read x_B, and nothing is modified before those tests either way)
    const int N = n + m;
    // "rc > EPS, and first or rc > best + EPS" over ascending non-basic indices (:105-121; the
    // equal-within-EPS clause needs a smaller index than the current one and can never fire in
    // ascending order).  With v = -rc this is eps_fold's "v < best - EPS" from best = +inf:
    // negation is exact, and fl(-b - EPS) = -fl(b + EPS).
    __shared__ int lds_i2[32];
    __shared__ double lds_v2[32];
    auto cand_value = [&](int k) {
        // no branch before a load: one address, one load, selects afterwards
        const bool xs = k < n;
        const double t = rev_ld<SC1>(xs ? rcx + k : y + (k - n));
        const double rc = xs ? t : -t;  // rcS_k = -y_k (:100-102)
        const bool cand = is_basic[k] == 0 && rc > kRevEps;
        return cand ? -rc : (double)NAN;
    };
    int cur;
    if (wmin != nullptr && rev_enter_hier<SC1>(wmin, gr, gy, rcx, y, is_basic, n, m, &cur, lds_i2,
                                               lds_v2)) {
        // (decided from the group minima the producing kernels left)
    } else if (s_val != nullptr && N <= s_cap) {
        constexpr int G = 8;
        for (int k0 = tid; k0 < N; k0 += G * nt) {
            double t[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int k = k0 + g * nt;
                t[g] = cand_value(k < N ? k : k0);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int k = k0 + g * nt;
                if (k < N) s_val[k] = t[g];
            }
        }
        __syncthreads();
        cur = eps_fold<K>(0, N, INFINITY, [&](int k) { return s_val[k]; }, lds_i2, lds_v2);
    } else {
        cur = eps_fold<K>(0, N, INFINITY, cand_value, lds_i2, lds_v2);
    }
    if (tid == 0) st->entering = cur;
    if (dbg && tid == 0) atomicMax(dbg + 8, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    // GetColumn (:390-396) rides the tail of this launch: structural e -> acol = A[:, e] = row e of
    // At (input of u = B^-1 a_e); slack e = n + k -> u = BInverse[:, k] directly (:151), one element
    // per row of B^-1, eight requested before the first is stored.  No entering variable:
    // k_rev_ratio reports the optimum.
    if (cur < 0) return;
    if (cur < n) {
        // (acol == nullptr: the caller's next kernel reads row `cur` of At itself -- the fused path)
        if (acol) {
            const double* __restrict__ src = At + (size_t)cur * ldt;
            for (int i = tid; i < m; i += nt) acol[i] = src[i];
        }
        return;
    }
    const double* __restrict__ src = Binv + (cur - n);
    constexpr int G = 8;
    for (int i0 = tid; i0 < m; i0 += G * nt) {
        double t[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = i0 + g * nt;
            t[g] = src[(size_t)(i < m ? i : i0) * ldb];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = i0 + g * nt;
            if (i < m) u[i] = t[g];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Ratio test, RevisedPrimalSimplexSolver.cs:154-176:
//     for i ascending, u_i > EPS, ratio = xB_i / u_i:
//         ratio < best - EPS  or  (|ratio - best| <= EPS and (none yet or basic[i] < basic[row]))
// `best` may move UP by up to EPS on a band take, so no prefix-minimum shortcut is valid; the
// same exact "next take" replay as k_rev_enter is used.  Then the bookkeeping of :181-212 and the
// column of the elementary matrix E (:266-272): fac[r] = 1/p, fac[i] = -u_i / p.  The old pivot
// row of B^-1 is copied to browbuf because k_rev_update works in place.
// s_rat / s_bvi: kRatioLds entries each of LDS scratch (the caller's: the fused kernel lends its ring).
template <bool SC1>
__device__ __forceinline__ void rev_ratio_body(const double* __restrict__ u,
                                               const double* __restrict__ xB,
                                               int32_t* __restrict__ basic,
                                               uint8_t* __restrict__ is_basic,
                                               double* __restrict__ cB, const double* __restrict__ c,
                                               const double* __restrict__ Binv, int ldb,
                                               double* __restrict__ browbuf,
                                               double* __restrict__ fac, int32_t* __restrict__ log,
                                               int n, int m, RevState* st, double* s_rat,
                                               int* s_bvi, double* s_u = nullptr,
                                               unsigned long long* dbg = nullptr) {
    __shared__ int lds[16];
    __shared__ double lds_best;
    __shared__ double s_bmin[kRatioLds / kWave];
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    const int e = st->entering;
    // what the bookkeeping at the end needs of the control block and of c: requested now
    const int64_t it0 = st->iter, max_iter0 = st->max_iter, log_cap0 = st->log_cap;
    const double ce = (e >= 0 && e < n) ? c[e] : 0.0;
    // u, x_B and the basis rows of this lane are requested up front, all at once (first
    // kCacheR * nt rows; rows beyond are re-read by the slow paths below): one memory round trip
    // for the whole tail instead of one per row and use.
    constexpr int kCacheR = 8;
    double uu[kCacheR], xx[kCacheR];
    double rat[kCacheR];
    int bvi[kCacheR];
#pragma unroll
    for (int q = 0; q < kCacheR; ++q) {
        const int i = tid + q * nt;
        const int ic = i < m ? i : 0;
        uu[q] = rev_ld<SC1>(u + ic);
        xx[q] = rev_ld<SC1>(xB + ic);
        bvi[q] = basic[ic];
    }
    {   // the loop head's exits, in the C#'s order: infeasible basis (:90-91), optimal (:124-146),
        // then (no C# counterpart) the caller's pivot limit
        int bad = INT_MAX;
#pragma unroll
        for (int q = kCacheR - 1; q >= 0; --q) {
            const int i = tid + q * nt;
            if (i < m && xx[q] < -kRevEps) bad = i;
        }
        if (bad == INT_MAX)
            for (int i = tid + kCacheR * nt; i < m; i += nt)
                if (rev_ld<SC1>(xB + i) < -kRevEps) { bad = i; break; }
        bad = block_min_int(bad, lds);
        int32_t out = kRunning;
        if (bad != INT_MAX) out = LPR_INFEASIBLE_BASIS;
        else if (e < 0) out = LPR_OK_OPTIMAL;
        else if (max_iter0 > 0 && it0 >= max_iter0) out = LPR_PIVOT_LIMIT;
        __syncthreads();  // every lane has read the state before lane 0 changes it
        if (out != kRunning) {
            if (tid == 0) st->status = out;
            return;
        }
    }

    // Ratios of this lane's rows: NaN marks "u_i <= EPS" (:161,:172-175).
#pragma unroll
    for (int q = 0; q < kCacheR; ++q) {
        const int i = tid + q * nt;
        rat[q] = (i < m && uu[q] > kRevEps) ? ieee_div(xx[q], uu[q]) : (double)NAN;
        if (!(i < m)) bvi[q] = 0;
    }
    const bool cached_all = m <= kCacheR * nt;
    int row = -1;           // leavingRow
    double best = DBL_MAX;  // bestRatio
    int cur = -1;           // last index examined by the replayed loop
    // Up to kRatioLds rows: the ratios go to LDS and ONE wave replays the C#'s loop as it is
    // written, 64 rows at a time -- a ballot finds the first row after the last take that the loop
    // would take next -- with no workgroup barrier per take (three barriers per take cost ~2 us
    // x ~9 takes of the 23 us this kernel took at m = 4096).
    __shared__ int s_row;
    if (dbg && tid == 0) atomicMax(dbg + 9, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    const bool in_lds = m <= kRatioLds && cached_all;  // every row's ratio is in a register of its lane
    if (in_lds) {
#pragma unroll
        for (int q = 0; q < kCacheR; ++q) {
            const int i = tid + q * nt;
            if (i < m) {
                s_rat[i] = rat[q];
                s_bvi[i] = bvi[q];
                if (s_u) s_u[i] = uu[q];
            }
        }
        __syncthreads();
        // minimum ratio of every block of 64 rows (NaN: no candidate in it).  A row can only be
        // taken if ratio <= best + EPS; the replay below skips a block whose minimum is above
        // best + 2 EPS at the time it gets there (best only changes on a take) -- exact: above
        // that, ratio - best > EPS however the subtraction rounds, or ratio is at least one ulp
        // above a best whose ulp exceeds EPS.
        const int nblk = (m + kWave - 1) / kWave;
        // (a lane takes 8 consecutive rows, 8 adjacent lanes make a block of 64: three shuffles)
        for (int g0 = 0; g0 < nblk * 8; g0 += nt) {
            const int g = g0 + tid;  // group of 8 rows
            double v = (double)NAN;
            if (g < nblk * 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = g * 8 + k;
                    v = fmin(v, (i < m) ? s_rat[i] : (double)NAN);
                }
            }
            v = fmin(v, __shfl_xor(v, 1, kWave));
            v = fmin(v, __shfl_xor(v, 2, kWave));
            v = fmin(v, __shfl_xor(v, 4, kWave));
            if (g < nblk * 8 && (g & 7) == 0) s_bmin[g >> 3] = v;
        }
        __syncthreads();
        if (tid < kWave) {
            int brow = 0;
            const double bm = (tid < nblk) ? s_bmin[tid] : (double)NAN;
            int bnext = 0;
            for (;;) {
                unsigned long long cand = __ballot(bm <= best + 2.0 * kRevEps);
                cand &= (bnext >= kWave) ? 0ull : (~0ull << bnext);
                if (cand == 0ull) break;
                const int bq = __builtin_amdgcn_readfirstlane(__builtin_ctzll(cand));
                bnext = bq + 1;
                const int b0 = bq * kWave;
                const int i = b0 + tid;
                const double ratio = (i < m) ? s_rat[i] : NAN;
                const int bi = (i < m) ? s_bvi[i] : 0;
                unsigned long long alive = ~0ull;
                for (;;) {
                    const bool take = ratio < best - kRevEps ||
                                      (fabs(ratio - best) <= kRevEps && (row == -1 || bi < brow));
                    const unsigned long long hit = __ballot(take) & alive;
                    if (hit == 0ull) break;
                    const int fl = __builtin_amdgcn_readfirstlane(__builtin_ctzll(hit));
                    best = readlane_f64(ratio, fl);
                    brow = __builtin_amdgcn_readlane(bi, fl);
                    row = b0 + fl;
                    alive = (fl == kWave - 1) ? 0ull : (~0ull << (fl + 1));
                }
            }
            if (tid == 0) s_row = row;
        }
        __syncthreads();
        row = s_row;
    } else
    for (;;) {
        const int brow = (row >= 0) ? basic[row] : 0;
        int first = INT_MAX;
#pragma unroll
        for (int q = 0; q < kCacheR; ++q) {
            const int i = tid + q * nt;
            const double ratio = rat[q];  // NaN fails both tests below, as the C# skips the row
            if (first == INT_MAX && i < m && i > cur &&
                (ratio < best - kRevEps ||
                 (fabs(ratio - best) <= kRevEps && (row == -1 || bvi[q] < brow))))
                first = i;
        }
        if (first == INT_MAX && !cached_all) {
            for (int i = tid + kCacheR * nt; i < m; i += nt) {
                if (i <= cur) continue;
                const double ui = rev_ld<SC1>(u + i);
                if (!(ui > kRevEps)) continue;
                const double ratio = ieee_div(rev_ld<SC1>(xB + i), ui);
                if (ratio < best - kRevEps ||
                    (fabs(ratio - best) <= kRevEps && (row == -1 || basic[i] < brow))) {
                    first = i;
                    break;
                }
            }
        }
        const int mine = first;
        first = block_min_int(first, lds);
        if (first == INT_MAX) break;
        if (mine == first) {  // exactly one lane found it: publish its ratio
            double v = NAN;
#pragma unroll
            for (int q = 0; q < kCacheR; ++q)
                if (tid + q * nt == first) v = rat[q];
            if (first >= kCacheR * nt) v = ieee_div(rev_ld<SC1>(xB + first), rev_ld<SC1>(u + first));
            lds_best = v;
        }
        __syncthreads();
        cur = first;
        row = first;
        best = lds_best;
    }
    if (row < 0) {
        if (tid == 0) st->status = LPR_UNBOUNDED;  // :178-179
        return;
    }
    if (dbg && tid == 0) atomicMax(dbg + 10, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    const int leavingVar = in_lds ? s_bvi[row] : basic[row];
    if (leavingVar == e) {
        if (tid == 0) st->status = LPR_ENTERING_ALREADY_BASIC;  // :182-183
        return;
    }
    const double pivot = (in_lds && s_u) ? s_u[row] : rev_ld<SC1>(u + row);
    __syncthreads();  // every lane has read basic[row] before it is overwritten
    if (tid == 0) {
        const int64_t it = it0;
        if (it < log_cap0) {
            log[3 * it] = row;
            log[3 * it + 1] = e;
            log[3 * it + 2] = leavingVar;
        }
        basic[row] = e;               // :195
        is_basic[e] = 1;              // nonBasic.Remove(entering) :196
        is_basic[leavingVar] = 0;     // nonBasic.Add(leavingVar)  :197-198
        cB[row] = ce;                 // :205,211 (c[e] for a structural, 0 for a slack)
        st->leaving_row = row;
        if (fabs(pivot) < kRevEps) st->status = LPR_PIVOT_TOO_SMALL;  // :267 (after the bookkeeping)
        else st->iter = it + 1;  // :249
    }
    if (fabs(pivot) < kRevEps) return;
    {   // the old pivot row of B^-1 is requested first, the kCacheR quotients share one check
        double br[kCacheR], fn[kCacheR], fd[kCacheR], fq[kCacheR];
#pragma unroll
        for (int q = 0; q < kCacheR; ++q) {
            const int i = tid + q * nt;
            br[q] = Binv[(size_t)row * ldb + (i < m ? i : 0)];
            fn[q] = (i == row) ? 1.0 : -uu[q];  // :272
            fd[q] = pivot;
        }
        ieee_div_n<kCacheR>(fn, fd, fq);
#pragma unroll
        for (int q = 0; q < kCacheR; ++q) {
            const int i = tid + q * nt;
            if (i < m) {
                fac[i] = fq[q];
                browbuf[i] = br[q];
            }
        }
    }
    for (int i = tid + kCacheR * nt; i < m; i += nt) {
        fac[i] = (i == row) ? ieee_div(1.0, pivot) : ieee_div(-rev_ld<SC1>(u + i), pivot);  // :272
        browbuf[i] = Binv[(size_t)row * ldb + i];
    }
    for (int i = m + tid; i < ldb; i += nt) browbuf[i] = 0.0;
}


}  // namespace lpr
