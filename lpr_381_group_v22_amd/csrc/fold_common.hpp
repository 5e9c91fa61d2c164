// fold_common.hpp -- the sequential "better by more than EPS" fold of the reference's selection
// loops, replayed exactly on one workgroup (used by sens_engine.hip and revised_kernels.hip).
// Not part of the ABI.
#pragma once

#include "engine_common.hpp"

#pragma clang fp contract(off)

namespace lpr {

constexpr double kFoldEps = 1e-9;

// The C#'s selections are sequential folds "take idx when val(idx) < best - EPS" over ascending
// idx (NaN = not a candidate).  eps_fold replays one exactly and returns the last index taken
// (-1: none).  Two facts make it cheap:
//  (1) a take is always a strict prefix minimum of the sequence: everything before it was either
//      taken (>= the current best) or skipped (>= its best - EPS >= the current best - EPS), and
//      the take is below best - EPS.  So only the left-to-right minima -- about ln(n) of them on
//      unordered data -- can ever be taken; they are found with one block-wide prefix-min scan
//      (each thread owns a contiguous chunk, values stay in registers).
//  (2) the sequential loop is then replayed over those few flagged candidates only.  They are
//      compacted in index order into LDS (at most kFoldCap of them) and ONE wave walks them, 64 at
//      a time: a ballot finds the first lane after the last take with val < best - EPS -- what
//      the C# loop takes next -- with no workgroup barrier per take (a barrier per take was 1.2 us
//      x ~10 takes).  More candidates than kFoldCap (a long strictly decreasing input): the
//      next-take search over the whole block, one round per take, as before.
// lds_i / lds_v: 32 entries each (two banks of one slot per wave, alternating per round).
constexpr int kFoldCap = 1024;
__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

template <int K, class F>
__device__ __forceinline__ int eps_fold(int lo, int hi, double best, F val, int* lds_i,
                                         double* lds_v) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (kWave - 1), wave = tid / kWave, nw = nt / kWave;
    const int n = hi - lo;
    if (n <= 0) return -1;  // (uniform over the block)
    const int c = (n + nt - 1) / nt;  // candidates per thread, contiguous
    const bool cached = c <= K;
    const int base = lo + tid * c;
    double v[K];
    unsigned flags = 0;  // bit k: candidate base + k can still be taken
    __syncthreads();     // lds_i / lds_v may still be read by a previous fold
    if (cached) {
        // all K values are requested before the first is looked at: val() runs on an index that is
        // always valid (a lane's unused slots re-read `lo`), so its loads carry no branch and go
        // out back to back -- one memory round trip per thread instead of one per candidate
        // (measured in the revised entering fold: 24 us -> a few)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int idx = base + k;
            const bool ok = k < c && idx < hi;
            const double x = val(ok ? idx : lo);
            v[k] = ok ? x : (double)NAN;
        }
        double lmin = INFINITY;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (v[k] < lmin) {  // strict prefix minimum within the chunk
                flags |= 1u << k;
                lmin = v[k];
            }
        }
        // exclusive prefix minimum of the chunk minima over the block
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
        const double wv = (lane < nw) ? lds_v[lane] : INFINITY;
        for (int w = 0; w < wave; ++w) {
            const double x = readlane_f64(wv, w);
            if (x < exc) exc = x;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (((flags >> k) & 1u) && !(v[k] < exc)) flags &= ~(1u << k);
    }
    int cur = -1;
    if (cached) {
        __shared__ double cand_v[kFoldCap];
        __shared__ int cand_i[kFoldCap];
        __shared__ int s_res[2];
        // ordered compaction: exclusive prefix sum of the per-thread candidate counts
        const int cnt = __popc(flags);
        int inc = cnt;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const int o = __shfl_up(inc, off, kWave);
            if (lane >= off) inc += o;
        }
        if (lane == kWave - 1) lds_i[wave] = inc;
        __syncthreads();
        const int wt = (lane < nw) ? lds_i[lane] : 0;
        int before = 0, total = 0;
        for (int w = 0; w < nw; ++w) {
            const int x = __builtin_amdgcn_readlane(wt, w);
            if (w < wave) before += x;
            total += x;
        }
        __syncthreads();  // lds_i is reused by the fallback below
        if (total <= kFoldCap) {
            int pos = before + inc - cnt;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if ((flags >> k) & 1u) {
                    cand_v[pos] = v[k];
                    cand_i[pos] = base + k;
                    ++pos;
                }
            }
            __syncthreads();
            if (wave == 0) {
                for (int b0 = 0; b0 < total; b0 += kWave) {
                    const int k = b0 + lane;
                    const double x = (k < total) ? cand_v[k] : NAN;
                    const int xi = (k < total) ? cand_i[k] : 0;
                    unsigned long long alive = ~0ull;
                    for (;;) {
                        const unsigned long long hit = __ballot(x < best - kFoldEps) & alive;
                        if (hit == 0ull) break;
                        const int fl = __builtin_amdgcn_readfirstlane(__builtin_ctzll(hit));
                        best = readlane_f64(x, fl);
                        cur = __builtin_amdgcn_readlane(xi, fl);
                        alive = (fl == kWave - 1) ? 0ull : (~0ull << (fl + 1));
                    }
                }
                if (lane == 0) s_res[0] = cur;
            }
            __syncthreads();
            return s_res[0];
        }
    }
    for (int round = 0;; ++round) {
        int first = INT_MAX;
        double fv = 0.0;
        if (cached) {
            if (__ballot(flags != 0u) != 0ull) {
                int myfirst = INT_MAX;
                double myv = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if ((flags >> k) & 1u) {
                        const int idx = base + k;
                        if (idx > cur && v[k] < best - kFoldEps) {
                            if (myfirst == INT_MAX) {
                                myfirst = idx;
                                myv = v[k];
                            }
                        } else {
                            flags &= ~(1u << k);  // best only decreases: dead for good
                        }
                    }
                }
                // chunks are ordered by lane: the lowest lane with a hit holds the wave's first
                const unsigned long long mask = __ballot(myfirst != INT_MAX);
                if (mask != 0ull) {
                    const int fl = __builtin_amdgcn_readfirstlane(__builtin_ctzll(mask));
                    first = __builtin_amdgcn_readlane(myfirst, fl);
                    fv = readlane_f64(myv, fl);
                }
            }
        } else {
            for (int idx = lo + tid; idx < hi; idx += nt) {
                if (idx <= cur) continue;
                const double x = val(idx);
                if (x < best - kFoldEps) {
                    first = idx;
                    fv = x;
                    break;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const int oi = __shfl_xor(first, off, kWave);
                const double ov = __shfl_xor(fv, off, kWave);
                if (oi < first) {
                    first = oi;
                    fv = ov;
                }
            }
        }
        const int bank = (round & 1) * 16;
        if (lane == 0) {
            lds_i[bank + wave] = first;
            lds_v[bank + wave] = fv;
        }
        __syncthreads();
        const int ci = (lane < nw) ? lds_i[bank + lane] : INT_MAX;
        const double cv = (lane < nw) ? lds_v[bank + lane] : 0.0;
        first = INT_MAX;
        for (int w = 0; w < nw; ++w) {
            const int x = __builtin_amdgcn_readlane(ci, w);
            if (x < first) {
                first = x;
                fv = readlane_f64(cv, w);
            }
        }
        if (first == INT_MAX) break;
        cur = first;
        best = fv;
    }
    return cur;
}

}  // namespace lpr
