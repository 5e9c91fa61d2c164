// bb_kernels.hip -- gfx950 kernels of the Branch & Bound LP-relaxation path
// (reference: LPR_381_Group_V22/IntegerProgramming/BranchBoundSimplexSolver.cs).
//
// All kernels are batched: blockIdx.z (or .y / .x where noted) is the child sub-problem ("slot") of
// the current batch, so the two children of a DFS node -- or the whole frontier of a level -- are
// evaluated by the same launches.  Every node tableau lives in a rows_cap x ld buffer in HBM, ld a
// multiple of 16 doubles.
//
//   k_bb_child_init    AddConstraint, first half (:701-747): round, insert the slack column,
//                      append the branching row, round again
//   k_bb_basic_scan    IdentifyBasicVariables (:642-663): rounded column sums, row of the first 1.0
//   k_bb_eliminate     IdentifyBasicVariables ordering (:665-690) + the sequential eliminations of
//                      AddConstraint (:752-797)
//   k_bb_round(_clean) RoundTableau (:552-567) (+ the -0 -> +0 pass of DoDualSimplex :307-313)
//   k_bb_select        one DoDualSimplex loop head (:305-400): phase logic, PerformDualPivot
//                      (:115-160) / PerformPrimalPivot (:203-253) selection, pivot row normalise
//   k_bb_update        the row elimination of both pivots (:174-192, :255-271), in place, only the
//                      rows whose factor is not zero
//   k_bb_node_info     GetObjective (:892-897) + decision values (:807-827 / :899-921)
#include "bb_common.hpp"

#pragma clang fp contract(off)

namespace lpr {

constexpr double kBBEps = 1e-6;  // BranchAndBound.epsilon (:493)

// ------------------------------------------------------------------ .NET Framework rounding
// Math.Round(double) -- COMDouble::Round (round half to even through floor(x + 0.5)).
// (The runtime's own text is `if (x == (double)(long long)x) return x; t = x + 0.5; f = floor(t);
// if (f == t && fmod(t, 2.0) != 0) f -= 1.0; return copysign(f, x)`.  The two tests are restated
// without the 64-bit integer conversion and without fmod -- "x is integral" is floor(x) == x for
// every finite x (above 2^52 every double is), "t is odd" is "t / 2 is not integral", exact because
// halving is -- which is 4x fewer instructions on the device and bit-identical: checked against
// the literal form over 2e8 random and edge operands on the CPU.)
__device__ __forceinline__ double dn_round_int(double x) {
    if (isnan(x) || isinf(x)) return x;
    if (floor(x) == x) return x;
    const double t = x + 0.5;
    double f = floor(t);
    const double h = t * 0.5;
    if (f == t && floor(h) != h) f -= 1.0;
    return copysign(f, x);
}
// Math.Round(double, 4) -- Math.InternalRound: scale, round, unscale; identity for |x| >= 1e16.
__device__ __forceinline__ double dn_round4(double x) {
    if (fabs(x) < 1e16) {
        x = x * 10000.0;
        x = dn_round_int(x);
        x = ieee_div(x, 10000.0);
    }
    return x;
}

// RoundNumber(RoundNumber(x)): the C# rounds a value again wherever a rounded tableau is handed on
// (:702 then :655 / :747).  Below 1e11 the second call returns its argument (x * 1e4 is within half
// a unit of the integer it came from, so it rounds back to it; checked over 2e8 operands); only
// above that is it evaluated.
__device__ __forceinline__ double dn_round4_twice(double x) {
    const double r = dn_round4(x);
    return (fabs(r) < 1e11) ? r : dn_round4(r);
}

// x rounded three times, then DoDualSimplex's -0 -> +0 (what a row of the parent has been through
// when the child's dual simplex starts: :702, :747, :799, :307-313)
__device__ __forceinline__ double dn_round4_thrice_clean(double x) {
    double r = dn_round4(x);
    if (!(fabs(r) < 1e11)) r = dn_round4(dn_round4(r));
    if (r == 0.0) r = 0.0;
    return r;
}

// ------------------------------------------------------------------ reductions
struct Cand {
    double v;
    int i;
};
__device__ __forceinline__ Cand cand_min(Cand a, Cand b) {
    if (b.i < 0) return a;
    if (a.i < 0) return b;
    if (b.v < a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}
__device__ __forceinline__ Cand block_cand_min(Cand c, double* lds_v, int* lds_i) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int nwaves = blockDim.x / kWave;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Cand o;
        o.v = __shfl_xor(c.v, off, kWave);
        o.i = __shfl_xor(c.i, off, kWave);
        c = cand_min(c, o);
    }
    __syncthreads();
    if (lane == 0) {
        lds_v[wave] = c.v;
        lds_i[wave] = c.i;
    }
    __syncthreads();
    Cand r;
    r.v = lds_v[0];
    r.i = lds_i[0];
    for (int w = 1; w < nwaves; ++w) {
        Cand o;
        o.v = lds_v[w];
        o.i = lds_i[w];
        r = cand_min(r, o);
    }
    return r;
}
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

// ------------------------------------------------------------------ AddConstraint, part 1
// grid (ceil(ld/256), rows_max_child, nslots).  Child tableau = parent with a zero column inserted
// before the RHS (:716-719) and the branching row appended (:721-744); every value goes through
// RoundNumber as often as the C# applies it (working = Round(base) :702, updated = Round(updated)
// :747).
constexpr int kBBRowsPerThread = 8;  // rows a thread of the element-wise passes walks

// Per node, one byte per row behind its scores (row rows_cap + 1 of its buffer: 1 + nvars doubles
// are used, and 8 (ld - 1 - nvars) >= rows_cap bytes are left): 1 = k_bb_finish stored a -0.0 in
// that row (Math.Round of a tiny negative value).  The children of the node start from its rows
// with -0 made +0 (:307-313); a child that takes the buffer over in place only has to visit the
// flagged rows for that.
__device__ __forceinline__ uint8_t* bb_negz(double* T, int ld, int rows_cap, int nvars) {
    return reinterpret_cast<uint8_t*>(T + (size_t)(rows_cap + 1) * ld + 1 + nvars);
}

__global__ __launch_bounds__(256) void k_bb_child_init(const BBSlot* __restrict__ slots, int ld,
                                                       uint8_t* __restrict__ touched_all,
                                                       int rows_cap16, int rows_cap, int nvars) {
    const BBSlot& s = slots[blockIdx.z];
    if (s.inplace) return;  // (k_bb_child_inplace)
    const int Rc = s.rows, Cc = s.cols;
    const int R = Rc - 1, C = Cc - 1;  // parent shape
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i0 = blockIdx.y * kBBRowsPerThread;
    // every row starts as a rounded copy of the parent's: untouched -- except the appended row,
    // which k_bb_eliminate rewrites (:756-796); and no row holds a -0 yet (bb_negz)
    if (blockIdx.x == 0 && threadIdx.x < kBBRowsPerThread && i0 + (int)threadIdx.x < rows_cap16) {
        touched_all[(size_t)blockIdx.z * rows_cap16 + i0 + threadIdx.x] =
            (i0 + (int)threadIdx.x == Rc - 1) ? 1 : 0;
        if (i0 + (int)threadIdx.x < rows_cap) bb_negz(s.cur, ld, rows_cap, nvars)[i0 + threadIdx.x] = 0;
    }
    if (j >= ld) return;
    const double* __restrict__ P = s.parent;
#pragma unroll
    for (int d = 0; d < kBBRowsPerThread; ++d) {
        const int i = i0 + d;
        if (i >= Rc) break;
        double v = 0.0;
        if (i < R) {
            // rows of the parent are not touched by the eliminations of :756-796, so what
            // RoundTableau (:799) and the -0 pass of :307-313 would make of them is written here
            // (k_bb_eliminate reads them WITHOUT rounding again: this is the thrice-rounded value
            // the C# reads through RoundNumber(updated[..]); it does the same for the new row)
            if (j < C - 1) v = dn_round4_thrice_clean(P[(size_t)i * ld + j]);
            else if (j == C) v = dn_round4_thrice_clean(P[(size_t)i * ld + (C - 1)]);
            // j == C - 1: the inserted 0.0; j > C: padding
        } else {
            if (j == s.var) v = 1.0;                        // RoundNumber(1) twice is 1
            if (j == C) v = dn_round4_twice(s.bound);       // :732
            if (j == C - 1) v = s.reverse ? -1.0 : 1.0;     // slackPosition :734-742
        }
        s.cur[(size_t)i * ld + j] = v;
    }
}

// grid (nslots).  The second child of a parent takes the parent's buffer over (slot.inplace): the
// parent's rows are what k_bb_child_init would write -- they have been through RoundNumber (the
// parent is not `big`), so rounding them again changes nothing -- except that the right-hand side
// moves one column to the right, the inserted column is 0 (:716-719) and a -0.0 becomes +0.0
// (:307-313): only in the rows k_bb_finish flagged.  Then the branching row (:721-744).  Runs
// AFTER k_bb_child_init of the batch (the sibling copies the parent's rows first).
__global__ __launch_bounds__(256) void k_bb_child_inplace(const BBSlot* __restrict__ slots, int ld,
                                                          uint8_t* __restrict__ touched_all,
                                                          int rows_cap16, int rows_cap, int nvars) {
    const BBSlot& s = slots[blockIdx.x];
    if (!s.inplace) return;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int Rc = s.rows, Cc = s.cols;
    const int R = Rc - 1, C = Cc - 1;  // parent shape
    double* __restrict__ T = s.cur;
    uint8_t* __restrict__ negz = bb_negz(T, ld, rows_cap, nvars);
    uint8_t* __restrict__ touched = touched_all + (size_t)blockIdx.x * rows_cap16;
    __shared__ int s_nz, s_rows[256];
    if (tid == 0) s_nz = 0;
    __syncthreads();
    for (int i = tid; i < R; i += nt) {
        double rhs = T[(size_t)i * ld + (C - 1)];
        if (rhs == 0.0) rhs = 0.0;
        T[(size_t)i * ld + C] = rhs;
        T[(size_t)i * ld + (C - 1)] = 0.0;
        touched[i] = 0;
        if (negz[i]) {  // (rarely any: the flagged rows are listed, 256 at a time at most)
            const int pos = atomicAdd(&s_nz, 1);
            if (pos < 256) s_rows[pos] = i;
        }
    }
    __syncthreads();
    const int nz = s_nz;
    if (nz <= 256) {
        for (int q = 0; q < nz; ++q) {
            const int i = s_rows[q];
            for (int j = tid; j < C - 1; j += nt) {
                const double v = T[(size_t)i * ld + j];
                if (v == 0.0) T[(size_t)i * ld + j] = 0.0;
            }
        }
    } else {  // (more than the list holds: every flagged row, one after the other)
        for (int i = 0; i < R; ++i) {
            if (!negz[i]) continue;  // (uniform: every thread reads the same byte)
            for (int j = tid; j < C - 1; j += nt) {
                const double v = T[(size_t)i * ld + j];
                if (v == 0.0) T[(size_t)i * ld + j] = 0.0;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < rows_cap; i += nt) negz[i] = 0;
    for (int j = tid; j < ld; j += nt) {
        double v = 0.0;
        if (j == s.var) v = 1.0;                        // RoundNumber(1) twice is 1
        if (j == C) v = dn_round4_twice(s.bound);       // :732
        if (j == C - 1) v = s.reverse ? -1.0 : 1.0;     // slackPosition :734-742
        T[(size_t)R * ld + j] = v;
    }
    if (tid == 0) touched[R] = 1;
    for (int i = R + 1 + tid; i < rows_cap16; i += nt) touched[i] = 0;
}

// grid (ceil(C/64), nslots).  One lane per OLD column k: rounded values summed in row order
// (Enumerable.Sum), |Round(sum) - 1| <= eps marks it "basic" (:657-661); key = row of the first
// value that is exactly 1.0, else the row count (:680).
__global__ __launch_bounds__(64) void k_bb_basic_scan(const BBSlot* __restrict__ slots, int ld,
                                                     int32_t* __restrict__ bflag,
                                                     int32_t* __restrict__ bkey) {
    // one scan per distinct PARENT of the batch (its two children share it)
    const BBSlot& s = slots[slots[blockIdx.y].rep];
    const int R = s.rows - 1, C = s.cols - 1;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= C) return;
    const double* __restrict__ P = s.parent;
    double sum = 0.0;
    int key = R;
    constexpr int U = 8;  // rows in flight per lane (the sum itself stays in row order)
    for (int i0 = 0; i0 < R; i0 += U) {
        double x[U];
#pragma unroll
        for (int d = 0; d < U; ++d) x[d] = (i0 + d < R) ? P[(size_t)(i0 + d) * ld + k] : 0.0;
#pragma unroll
        for (int d = 0; d < U; ++d) {
            if (i0 + d < R) {
                // working = Round(base) :702, then RoundNumber inside Identify :655
                const double v = dn_round4_twice(x[d]);
                sum = sum + v;
                if (key == R && v == 1.0) key = i0 + d;
            }
        }
    }
    sum = dn_round4(sum);
    bflag[(size_t)blockIdx.y * ld + k] = (fabs(sum - 1.0) <= kBBEps) ? 1 : 0;
    bkey[(size_t)blockIdx.y * ld + k] = key;
}

// grid (nslots), one workgroup per child.  Orders the basic columns as
// OrderBy(first-1.0 row) does (stable: ties by column), then applies the eliminations of :756-796
// one basic column after the other -- each step reads the coefficient the previous steps left in
// the new row, so the steps are sequential; inside a step the row is updated in parallel.
// NOTE the C# indexes the widened tableau with the OLD column numbers (the RHS column index now
// names the inserted slack column); restated as is.
__global__ __launch_bounds__(1024) void k_bb_eliminate(const BBSlot* __restrict__ slots, int ld,
                                                       const int32_t* __restrict__ bflag,
                                                       const int32_t* __restrict__ bkey,
                                                       int32_t* __restrict__ blist, int side_row) {
    __shared__ int lds[16];
    __shared__ int s_count;
    const BBSlot& s = slots[blockIdx.x];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int Rc = s.rows, Cc = s.cols;
    const int R = Rc - 1, C = Cc - 1;
    const int crow = R;
    // side_row >= 0: the parent's IdentifyBasicVariables scan was left in row `side_row` of its own
    // buffer by k_bb_finish (flags, then keys); otherwise k_bb_basic_scan has just made it
    const int32_t* __restrict__ pside =
        reinterpret_cast<const int32_t*>(s.parent + (size_t)(side_row >= 0 ? side_row : 0) * ld);
    const int32_t* __restrict__ flag = side_row >= 0 ? pside : bflag + (size_t)s.pscan * ld;
    const int32_t* __restrict__ key = side_row >= 0 ? pside + ld : bkey + (size_t)s.pscan * ld;
    int32_t* __restrict__ list = blist + (size_t)blockIdx.x * ld;
    double* __restrict__ T = s.cur;
    if (tid == 0) s_count = 0;
    __syncthreads();
    // OrderBy(first-1.0 row), stable: rank = flagged columns with a smaller (key, column).  The
    // flags and keys of all columns are walked out of LDS (C / 1024 columns per lane x C steps).
    extern __shared__ int s_fk[];  // [C] key of a flagged column, INT_MAX otherwise
    for (int k = tid; k < C; k += nt) s_fk[k] = flag[k] ? key[k] : INT_MAX;
    __syncthreads();
    for (int k = tid; k < C; k += nt) {
        const int mykey = s_fk[k];
        if (mykey == INT_MAX) continue;
        int rank = 0;
        for (int k2 = 0; k2 < C; ++k2) {
            const int o = s_fk[k2];
            if (o < mykey || (o == mykey && k2 < k)) ++rank;
        }
        list[rank] = k;
        atomicAdd(&s_count, 1);
    }
    __syncthreads();
    const int count = s_count;
    const int reverse = s.reverse;
    // The C# walks the basic columns in order and eliminates where the new row's coefficient is
    // non-zero (:756-796); a step with a zero coefficient changes nothing, so the walk is replayed
    // as "find the next column whose coefficient is non-zero in the row AS IT IS NOW, eliminate,
    // continue behind it" -- a handful of rounds instead of one barrier per basic column.
    int q0 = 0;
    for (;;) {
        int firstq = INT_MAX;
        for (int q = q0 + tid; q < count; q += nt) {
            const double cf = dn_round4(T[(size_t)crow * ld + list[q]]);  // :758
            if (fabs(cf) > kBBEps) {
                firstq = q;
                break;
            }
        }
        const int q = block_min_int(firstq, lds);
        if (q == INT_MAX) break;
        const int colIndex = list[q];
        const double coefficient = dn_round4(T[(size_t)crow * ld + colIndex]);
        int first = INT_MAX;
        for (int row = tid; row < R; row += nt)
            if (fabs(T[(size_t)row * ld + colIndex] - 1.0) <= kBBEps) {  // (rounded by child_init)
                first = row;
                break;
            }
        const int pivotRow = block_min_int(first, lds);  // :763-770
        if (pivotRow != INT_MAX) {
            for (int col = tid; col < Cc; col += nt) {
                const double pivotVal = T[(size_t)pivotRow * ld + col];  // (rounded by child_init)
                const double constraintVal = dn_round4(T[(size_t)crow * ld + col]);
                double newVal;
                if (reverse) {
                    const double prod = coefficient * constraintVal;
                    newVal = pivotVal - prod;  // :785
                } else {
                    const double prod = coefficient * pivotVal;
                    newVal = constraintVal - prod;  // :789
                }
                T[(size_t)crow * ld + col] = dn_round4(newVal);  // :792
            }
        }
        q0 = q + 1;
        __syncthreads();  // the next coefficients are read from the row just rewritten
    }
    // RoundTableau (:799) and the -0 -> +0 of :307-313 for the new row (the other rows: child_init)
    for (int col = tid; col < Cc; col += nt) {
        double v = dn_round4(T[(size_t)crow * ld + col]);
        if (v == 0.0) v = 0.0;
        T[(size_t)crow * ld + col] = v;
    }
}

// grid (ceil(ld/256), rows_max, nslots).  RoundTableau over slot.cur; `clean` adds the
// `if (x == -0.0) x = 0.0` pass DoDualSimplex applies to the tableau it is handed (:307-313).
// slot.big is set when a rounded value is not below 1e11 (or not finite): only then can rounding the
// tableau AGAIN change it (dn_round4_twice), so the host may skip the next RoundAllTableaux.
__global__ __launch_bounds__(256) void k_bb_round(BBSlot* __restrict__ slots, int ld, int clean) {
    BBSlot& s = slots[blockIdx.z];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= s.cols) return;
    const int i0 = blockIdx.y * kBBRowsPerThread;
    bool big = false;
#pragma unroll
    for (int d = 0; d < kBBRowsPerThread; ++d) {
        const int i = i0 + d;
        if (i >= s.rows) break;
        double v = dn_round4(s.cur[(size_t)i * ld + j]);
        if (clean && v == 0.0) v = 0.0;
        big = big || !(fabs(v) < 1e11);
        s.cur[(size_t)i * ld + j] = v;
    }
    if (big) atomicOr(&s.big, 1);
}

// ------------------------------------------------------------------ DoDualSimplex loop head
// grid (nslots).  State machine of :305-400 for one child; see bb_common.hpp for the states.
__global__ __launch_bounds__(1024) void k_bb_select(BBSlot* slots, double* __restrict__ rowbuf_all,
                                                    double* __restrict__ colbuf_all, int ld,
                                                    int rows_cap, int32_t* __restrict__ trace_all,
                                                    int trace_cap, int32_t* running,
                                                    int step_no,
                                                    int32_t* __restrict__ rowlist_all,
                                                    uint8_t* __restrict__ touched_all,
                                                    int rows_cap16) {
    __shared__ double lds_v[16];
    __shared__ int lds_i[16];
    __shared__ int s_cnt;
    BBSlot* sp = &slots[blockIdx.x];
    const int tid = threadIdx.x, nt = blockDim.x;

    // snapshot the header (thread 0 rewrites it at the end)
    double* cur = sp->cur;
    int state = sp->state;
    int pivots = sp->pivots;
    int trace_n = sp->trace_n;
    const int had_flags = sp->do_update | sp->restore;
    const int R = sp->rows, C = sp->cols;
    __syncthreads();
    // (pivots are applied in place: `cur` is the child's tableau for its whole life, `nxt` only
    // ever holds the rows saved for a pivot that may be dropped)
    if (state >= kBBSolved) {
        if (tid == 0 && had_flags) {
            sp->do_update = 0;
            sp->restore = 0;
        }
        return;
    }
    const int old_state = state;
    double* __restrict__ rowbuf = rowbuf_all + (size_t)blockIdx.x * ld;
    double* __restrict__ colbuf = colbuf_all + (size_t)blockIdx.x * rows_cap;
    int do_update = 0, pr = -1, pc = -1;
    int backup = 0, restore = 0, nlist = sp->nlist;
    int tr_phase = -1;  // trace entry to append: 0 dual, 1 primal, 2 "last tableau dropped"
    const int rhs = C - 1;

    if (state == kBBDual) {
        int bad = 0;  // !rhsValues.All(num => num >= -1e-9)   (row 0 included, :315-320)
        for (int i = tid; i < R; i += nt)
            if (!(cur[(size_t)i * ld + rhs] >= -1e-9)) bad = 1;
        bad = __syncthreads_or(bad);
        if (!bad) {
            int notopt = 0;  // :345-348
            for (int j = tid; j < C - 1; j += nt)
                if (!(cur[j] >= 0)) notopt = 1;
            notopt = __syncthreads_or(notopt);
            state = notopt ? kBBPrimal : kBBSolved;  // optimal here skips the :351 block entirely
        } else {
            // ---- PerformDualPivot :115-160 ----
            Cand c;
            c.v = 0.0;
            c.i = -1;
            for (int i = tid; i < R; i += nt) {
                const double x = cur[(size_t)i * ld + rhs];
                if (x < 0 && (c.i < 0 || x < c.v)) {  // Min() of the negatives, first IndexOf
                    c.v = x;
                    c.i = i;
                }
            }
            c = block_cand_min(c, lds_v, lds_i);
            pr = c.i;
            if (pr < 0) {
                state = kBBInfeasible;  // `if (!negativeRhs.Any()) return (tableau, null)` :120
            } else {
                const double* __restrict__ prow = cur + (size_t)pr * ld;
                int non0inf = 0;
                Cand m;  // lexicographic min over theta > 0: the minimum and its first index
                m.v = 0.0;
                m.i = -1;
                for (int j = tid; j < C - 1; j += nt) {
                    const double a = prow[j];
                    const double th = (a < 0) ? fabs(ieee_div(cur[j], a)) : INFINITY;  // :126-143
                    if (!(th == 0 || th == INFINITY)) non0inf = 1;
                    if (th > 0 && (m.i < 0 || th < m.v)) {
                        m.v = th;
                        m.i = j;
                    }
                }
                non0inf = __syncthreads_or(non0inf);
                m = block_cand_min(m, lds_v, lds_i);
                if (!non0inf) {  // every theta is 0 or +inf -> minPositiveTheta = 0 (:146-147)
                    int first = INT_MAX;
                    for (int j = tid; j < C - 1; j += nt) {
                        const double a = prow[j];
                        const double th = (a < 0) ? fabs(ieee_div(cur[j], a)) : INFINITY;
                        if (th == 0) {
                            first = j;
                            break;
                        }
                    }
                    first = block_min_int(first, lds_i);
                    pc = (first == INT_MAX) ? -1 : first;
                } else {
                    pc = m.i;  // -1: Where(x > 0) empty -> +inf -> IndexOf misses (:148,:154)
                }
                if (pc < 0) {
                    state = kBBInfeasible;  // tableau[rowIndex][-1] throws -> (tableau, null)
                } else {
                    do_update = 1;
                    tr_phase = 0;
                }
            }
        }
    }

    if (state == kBBPrimal && !do_update) {
        int need_post = 0;
        int notopt = 0;  // :368-374
        for (int j = tid; j < C - 1; j += nt)
            if (!(cur[j] >= 0)) notopt = 1;
        notopt = __syncthreads_or(notopt);
        if (!notopt) {
            need_post = 1;
        } else {
            // ---- PerformPrimalPivot :203-253 (isMinimization == false) ----
            Cand c;
            c.v = 0.0;
            c.i = -1;
            for (int j = tid; j < C - 1; j += nt) {
                const double x = cur[j];
                if (x < 0 && x != 0 && (c.i < 0 || x < c.v)) {
                    c.v = x;
                    c.i = j;
                }
            }
            c = block_cand_min(c, lds_v, lds_i);
            pc = c.i;  // first occurrence of the minimum (IndexOf over the whole row, :220)
            int fail = (pc < 0);
            if (!fail) {
                int notneg = 0, anypos = 0, has0 = 0;
                Cand m;
                m.v = 0.0;
                m.i = -1;
                for (int i = 1 + tid; i < R; i += nt) {
                    const double a = cur[(size_t)i * ld + pc];
                    const double th = (a != 0) ? ieee_div(cur[(size_t)i * ld + rhs], a) : INFINITY;
                    if (!(th < 0)) notneg = 1;
                    if (th == 0) has0 = 1;
                    if (th > 0 && th != INFINITY) {
                        anypos = 1;
                        if (m.i < 0 || th < m.v) {
                            m.v = th;
                            m.i = i;
                        }
                    }
                }
                notneg = __syncthreads_or(notneg);
                anypos = __syncthreads_or(anypos);
                has0 = __syncthreads_or(has0);
                m = block_cand_min(m, lds_v, lds_i);
                if (!notneg) {
                    fail = 1;  // thetas.All(num => num < 0) (true for an empty list) :228-231
                } else if (!anypos) {
                    if (has0) {  // minTheta = 0.0 -> first theta equal to 0 (:236-237,:249)
                        int first = INT_MAX;
                        for (int i = 1 + tid; i < R; i += nt) {
                            const double a = cur[(size_t)i * ld + pc];
                            const double th = (a != 0) ? ieee_div(cur[(size_t)i * ld + rhs], a) : INFINITY;
                            if (th == 0) {
                                first = i;
                                break;
                            }
                        }
                        first = block_min_int(first, lds_i);
                        pr = first;
                    } else {
                        fail = 1;
                    }
                } else {
                    pr = m.i;
                }
                if (!fail && cur[(size_t)pr * ld + pc] == 0) fail = 1;  // :252
            }
            if (fail) {
                need_post = 1;  // thetaCol == null -> NullReferenceException -> catch -> break
                pr = pc = -1;
            } else {
                do_update = 1;
                tr_phase = 1;
            }
        }
        if (need_post) {  // :392-400
            int bad = 0;
            for (int i = tid; i < R; i += nt)
                if (!(cur[(size_t)i * ld + rhs] >= 0)) bad = 1;
            bad = __syncthreads_or(bad);
            if (bad) {
                if (pivots == 0) {
                    state = kBBFailed;  // pivotColumns.RemoveAt(-1) throws
                } else {
                    // tableaux.RemoveAt(Count - 1): back to the previous tableau -- the rows the
                    // last pivot changed were saved (backup below: this exit is only reachable
                    // behind a pivot that kept them) and k_bb_update copies them back
                    restore = 1;
                    pivots -= 1;
                    tr_phase = 2;
                    state = kBBSolved;
                }
            } else {
                state = kBBSolved;
            }
        }
    }

    if (do_update) {  // normalised pivot row (:174-178 / :257-261) and the factor column
        const double p = cur[(size_t)pr * ld + pc];
        const double* __restrict__ prow = cur + (size_t)pr * ld;
        int nonfinite = 0;
        for (int j = tid; j < ld; j += nt) {
            double v = (j < C) ? ieee_div(prow[j], p) : 0.0;
            if (v == 0.0) v = 0.0;  // `== -0.0` is true for both zeros
            rowbuf[j] = v;
            if (!(fabs(v) < INFINITY)) nonfinite = 1;
        }
        nonfinite = __syncthreads_or(nonfinite);
        double prhs = ieee_div(prow[rhs], p);
        if (prhs == 0.0) prhs = 0.0;
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        // The pivot changes row i only where its factor f_i = T[i, pc] is not zero: x - (0 * p_j)
        // is x for every finite p_j (the stored zeros are all +0, :334-340).  Those rows -- and the
        // pivot row -- go on the slot's row list; k_bb_update touches nothing else.  A B&B child
        // tableau is ~90 % zeros and so are its pivot columns: 13 % of the rows on the bench
        // instance.  (A non-finite entry in the pivot row: every row is listed, 0 * inf is NaN.)
        // cont: the tableau AFTER this pivot still has a right-hand side that fails `>= -1e-9`
        // (the value k_bb_update will store, formed the same way): the dual loop then goes on or
        // ends infeasible (:315-331) -- either way that tableau is never the one :395-400 drops,
        // so its old rows need not be kept.
        int32_t* __restrict__ list = rowlist_all + (size_t)blockIdx.x * rows_cap;
        int cont = 0;
        for (int i = tid; i < R; i += nt) {
            const double f = cur[(size_t)i * ld + pc];
            colbuf[i] = f;
            if (nonfinite || f != 0.0 || i == pr) {
                list[atomicAdd(&s_cnt, 1)] = i;
                touched_all[(size_t)blockIdx.x * rows_cap16 + i] = 1;  // (k_bb_finish)
            }
            double nr;
            if (i == pr) {
                nr = prhs;
            } else {
                const double prod = f * prhs;
                nr = cur[(size_t)i * ld + rhs] - prod;
            }
            if (nr == 0.0) nr = 0.0;
            if (!(nr >= -1e-9)) cont = 1;
        }
        cont = __syncthreads_or(cont);
        nlist = s_cnt;
        backup = (tr_phase == 0 && cont) ? 0 : 1;
        pivots += 1;
    }

    if (tid == 0) {
        if (tr_phase >= 0) {
            if (trace_n < trace_cap) {
                int32_t* tr = trace_all + ((size_t)blockIdx.x * trace_cap + trace_n) * 3;
                tr[0] = tr_phase;
                tr[1] = (tr_phase == 2) ? -1 : pr;
                tr[2] = (tr_phase == 2) ? -1 : pc;
            }
            trace_n += 1;
        }
        sp->state = state;
        sp->pivots = pivots;
        sp->pr = pr;
        sp->pc = pc;
        sp->do_update = do_update;
        sp->backup = backup;
        sp->restore = restore;
        sp->nlist = nlist;
        sp->trace_n = trace_n;
        if (old_state < kBBSolved && state >= kBBSolved) atomicSub(running, 1);
        // running[1]: the last step of this batch that found a child still at work (the host sizes
        // the first batch of the next level by it)
        if (old_state < kBBSolved) atomicMax(running + 1, step_no);
    }
}

__device__ __forceinline__ int align_up_dev(int x) { return (x + kLdAlign - 1) / kLdAlign * kLdAlign; }

// grid (1, ceil(rows_max/TR), nslots).  The pivot of :174-192 / :255-271 applied IN PLACE to the
// rows of the slot's row list (the rows whose factor is not zero, and the pivot row): TR listed
// rows per workgroup, all columns.  new = old - (f_i * p_j) -- product rounded, then the difference
// -- the pivot row replaced by the normalised one, every value through the -0 -> +0 rule the C#
// applies to each new tableau (:334-340, :355-361).  The C# builds a fresh tableau per pivot; the
// only use it ever makes of the previous one is :395-400 (drop the last tableau): when that can
// still happen the old rows are saved to slot.nxt first (backup), and a dropped pivot is undone by
// copying them back (restore).
// Algorithmic bytes per pivot: 2 * 8 * R * C; moved: 2 (3 with backup) * 8 * nlist * C.
template <int TR>
__global__ __launch_bounds__(256) void k_bb_update(const BBSlot* __restrict__ slots,
                                                   const double* __restrict__ rowbuf_all,
                                                   const double* __restrict__ colbuf_all,
                                                   const int32_t* __restrict__ rowlist_all, int ld,
                                                   int rows_cap) {
    const BBSlot& s = slots[blockIdx.z];
    if (!s.do_update && !s.restore) return;
    const int nl = s.nlist;
    const int k0 = blockIdx.y * TR;
    if (k0 >= nl) return;
    const int ld2 = ld >> 1;
    const int ncol2 = align_up_dev(s.cols) >> 1;
    const int32_t* __restrict__ list = rowlist_all + (size_t)blockIdx.z * rows_cap;
    int idx[TR];
#pragma unroll
    for (int k = 0; k < TR; ++k) idx[k] = (k0 + k < nl) ? list[k0 + k] : -1;
    double2* __restrict__ T2 = reinterpret_cast<double2*>(s.cur);
    double2* __restrict__ bak = reinterpret_cast<double2*>(s.nxt);
    if (s.restore) {
        for (int c2 = threadIdx.x; c2 < ncol2; c2 += blockDim.x)
#pragma unroll
            for (int k = 0; k < TR; ++k)
                if (idx[k] >= 0) T2[(size_t)idx[k] * ld2 + c2] = bak[(size_t)idx[k] * ld2 + c2];
        return;
    }
    const double2* __restrict__ prow2 = reinterpret_cast<const double2*>(rowbuf_all + (size_t)blockIdx.z * ld);
    const double* __restrict__ colbuf = colbuf_all + (size_t)blockIdx.z * rows_cap;
    const int r = s.pr;
    const bool keep = s.backup != 0;
    double f[TR];
#pragma unroll
    for (int k = 0; k < TR; ++k) f[k] = (idx[k] >= 0) ? colbuf[idx[k]] : 0.0;
    for (int c2 = threadIdx.x; c2 < ncol2; c2 += blockDim.x) {
        const double2 pr = prow2[c2];
        double2 x[TR];
#pragma unroll
        for (int k = 0; k < TR; ++k)
            if (idx[k] >= 0) x[k] = T2[(size_t)idx[k] * ld2 + c2];
#pragma unroll
        for (int k = 0; k < TR; ++k) {
            if (idx[k] >= 0) {
                if (keep) bak[(size_t)idx[k] * ld2 + c2] = x[k];
                const double px = f[k] * pr.x;
                const double py = f[k] * pr.y;
                double2 o;
                o.x = x[k].x - px;
                o.y = x[k].y - py;
                if (idx[k] == r) o = pr;
                if (o.x == 0.0) o.x = 0.0;
                if (o.y == 0.0) o.y = 0.0;
                T2[(size_t)idx[k] * ld2 + c2] = o;
            }
        }
    }
}

// grid (ceil(C/64), nslots), one lane per column.  What happens to a solved child between its last
// pivot and its own expansion, in ONE pass over it instead of three:
//   RoundAllTableaux(newTableaux) (:1124 / :1187): every entry rounded in place (k_bb_round);
//   GetObjective + the decision values (:892-897, :807-827) it is scored by when popped
//     (k_bb_node_info) -> row rows_cap + 1 of the child's own buffer: { z, x_0 .. };
//   IdentifyBasicVariables (:642-663) of it AS A PARENT (k_bb_basic_scan) -> row rows_cap: the
//     flags of all columns, then the keys.
// Valid while no rounded entry is >= 1e11 or non-finite (slot.big, as in k_bb_round): below that
// RoundNumber is idempotent, so the second rounding of the pop (:1047) and the roundings the
// consumers apply on top change nothing; a big node goes through the separate kernels.
// A row no pivot has written since k_bb_child_init stored it (touched == 0: ~2/3 of the rows of a
// child on the bench instance) holds values that have been through RoundNumber already, and below
// 2.2e11 RoundNumber gives such a value back unchanged (x = n / 1e4 rounded: x * 1e4 is within
// n * 2^-52 of the integer n, which the round-to-even step restores exactly): for those rows the
// three roundings per entry -- two divisions each, and this kernel is bound by them, not by bytes --
// are skipped and the entry itself is used.  (>= 1e11 sets slot.big as before.)
__global__ __launch_bounds__(64) void k_bb_finish(BBSlot* __restrict__ slots, int ld, int rows_cap,
                                                 int nvars, const uint8_t* __restrict__ touched_all,
                                                 int rows_cap16) {
    BBSlot& s = slots[blockIdx.y];
    if (s.state != kBBSolved) return;
    const uint8_t* __restrict__ touched = touched_all + (size_t)blockIdx.y * rows_cap16;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int R = s.rows, C = s.cols;
    if (k >= C) return;
    double* __restrict__ T = s.cur;
    uint8_t* __restrict__ negz = bb_negz(T, ld, rows_cap, nvars);
    double sum = 0.0;
    int key = R, frow = -1;
    bool big = false;
    double v0 = 0.0;
    constexpr int U = 16;  // rows in flight per lane (the sum itself stays in row order)
    for (int i0 = 0; i0 < R; i0 += U) {
        double x[U];
#pragma unroll
        for (int d = 0; d < U; ++d) x[d] = (i0 + d < R) ? T[(size_t)(i0 + d) * ld + k] : 0.0;
#pragma unroll
        for (int d = 0; d < U; ++d) {
            const int i = i0 + d;
            if (i < R && !touched[i]) {  // (the same for every lane: a scalar branch)
                const double v = x[d];
                big = big || !(fabs(v) < 1e11);
                if (i == 0) v0 = v;
                sum = sum + v;
                if (key == R && v == 1.0) key = i;
                if (frow < 0 && fabs(v - 1.0) <= kBBEps) frow = i;
            } else if (i < R) {
                const double v = dn_round4(x[d]);          // :1124 / :1187
                big = big || !(fabs(v) < 1e11);
                // (most entries are zeros or come from rows no pivot touched since child_init
                // rounded them: only a value whose bits change is written back)
                if (__double_as_longlong(v) != __double_as_longlong(x[d])) T[(size_t)i * ld + k] = v;
                if (__double_as_longlong(v) == (long long)0x8000000000000000ull) negz[i] = 1;
                if (i == 0) v0 = v;
                // working = Round(base) :702, Identify :655, :812-821 round v again: below 1e11
                // that gives v back (above, slot.big is set and none of this is used)
                sum = sum + v;
                if (key == R && v == 1.0) key = i;
                if (frow < 0 && fabs(v - 1.0) <= kBBEps) frow = i;
            }
        }
    }
    sum = dn_round4(sum);
    int32_t* __restrict__ side = reinterpret_cast<int32_t*>(T + (size_t)rows_cap * ld);
    side[k] = (fabs(sum - 1.0) <= kBBEps) ? 1 : 0;
    side[ld + k] = key;
    double* __restrict__ info = T + (size_t)(rows_cap + 1) * ld;
    if (k == C - 1) info[0] = dn_round4(v0);               // GetObjective :892-897
    if (k < nvars) {
        // the right-hand side of that row, rounded as the stored tableau and again by the reader
        // (the lane of column C - 1 may or may not have rounded it in place yet: the same value
        // either way below 1e11)
        info[1 + k] = (frow >= 0) ? dn_round4(dn_round4(T[(size_t)frow * ld + (C - 1)])) : 0.0;
    }
    if (big) atomicOr(&s.big, 1);
}

// grid (ceil((nvars + 1) / 256), count): the scores k_bb_finish left in each node's buffer, packed
__global__ __launch_bounds__(256) void k_bb_gather_info(const BBSlot* __restrict__ slots, int ld,
                                                        int rows_cap, int nvars,
                                                        double* __restrict__ info) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nvars) return;
    const BBSlot& s = slots[blockIdx.y];
    info[(size_t)blockIdx.y * (nvars + 1) + j] = s.cur[(size_t)(rows_cap + 1) * ld + j];
}

// grid (ceil(nvars/64), count).  info[slot] = { z, x_0 .. x_{nvars-1} }: GetObjective (:892-897)
// and the "first row whose rounded entry is 1" decision values (:807-827, :899-921; row 0 is
// part of the scan).
__global__ __launch_bounds__(64) void k_bb_node_info(const BBSlot* __restrict__ slots, int ld,
                                                    int nvars, double* __restrict__ info) {
    const BBSlot& s = slots[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double* __restrict__ T = s.cur;
    const int R = s.rows, C = s.cols;
    double* out = info + (size_t)blockIdx.y * (nvars + 1);
    if (i == 0) out[0] = dn_round4(T[C - 1]);
    if (i >= nvars) return;
    // the scan stops at the first row whose entry rounds to 1: a dependent load per row would cost a
    // memory round trip each (577 of them), so rows are fetched eight at a time and tested in order
    double val = 0.0;
    bool found = false;
    constexpr int U = 8;
    for (int j0 = 0; j0 < R && !found; j0 += U) {
        double v[U];
#pragma unroll
        for (int d = 0; d < U; ++d) v[d] = (j0 + d < R) ? T[(size_t)(j0 + d) * ld + i] : 0.0;
#pragma unroll
        for (int d = 0; d < U; ++d) {
            if (!found && j0 + d < R && fabs(dn_round4(v[d]) - 1.0) <= kBBEps) {
                val = dn_round4(T[(size_t)(j0 + d) * ld + (C - 1)]);
                found = true;
            }
        }
    }
    out[1 + i] = val;
}

// copies a (rows x cols, leading dimension src_ld) matrix into a node buffer (ld), zero padding
__global__ __launch_bounds__(256) void k_bb_copy_in(const double* __restrict__ src, int src_ld,
                                                    int rows, int cols, double* __restrict__ dst,
                                                    int ld) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows || j >= ld) return;
    dst[(size_t)i * ld + j] = (j < cols) ? src[(size_t)i * src_ld + j] : 0.0;
}

// ------------------------------------------------------------------ launchers

void bb_launch_copy_in(lpr_bb* b, const double* src, int src_ld, int rows, int cols, double* dst) {
    hipLaunchKernelGGL(k_bb_copy_in, dim3((b->ld + 255) / 256, rows), dim3(256), 0, b->eng->stream,
                       src, src_ld, rows, cols, dst, b->ld);
}

void bb_launch_round(lpr_bb* b, int nslots, int rows_max, int clean) {
    hipLaunchKernelGGL(k_bb_round,
                       dim3((b->ld + 255) / 256, (rows_max + kBBRowsPerThread - 1) / kBBRowsPerThread,
                            nslots),
                       dim3(256), 0, b->eng->stream, b->d_slots, b->ld, clean);
}

void bb_launch_node_info(lpr_bb* b, int count) {
    hipLaunchKernelGGL(k_bb_node_info, dim3((b->nvars + 63) / 64 > 0 ? (b->nvars + 63) / 64 : 1,
                                            count),
                       dim3(64), 0, b->eng->stream, b->d_slots, b->ld, b->nvars, b->info);
}

void bb_launch_finish(lpr_bb* b, int nslots, int cols_max) {
    hipLaunchKernelGGL(k_bb_finish, dim3((cols_max + 63) / 64, nslots), dim3(64), 0, b->eng->stream,
                       b->d_slots, b->ld, b->rows_cap, b->nvars, b->touched,
                       align_up(b->rows_cap, 16));
}

void bb_launch_gather_info(lpr_bb* b, int count) {
    hipLaunchKernelGGL(k_bb_gather_info, dim3((b->nvars + 1 + 255) / 256, count), dim3(256), 0,
                       b->eng->stream, b->d_slots, b->ld, b->rows_cap, b->nvars, b->info);
}

void bb_launch_add_constraint(lpr_bb* b, int nslots, int nparents, int rows_max, int cols_max,
                              bool side, bool inplace) {
    hipStream_t st = b->eng->stream;
    const dim3 egrid((b->ld + 255) / 256, (rows_max + kBBRowsPerThread - 1) / kBBRowsPerThread,
                     nslots);
    hipLaunchKernelGGL(k_bb_child_init, egrid, dim3(256), 0, st, b->d_slots, b->ld, b->touched,
                       align_up(b->rows_cap, 16), b->rows_cap, b->nvars);
    if (inplace)
        hipLaunchKernelGGL(k_bb_child_inplace, dim3(nslots), dim3(256), 0, st, b->d_slots, b->ld,
                           b->touched, align_up(b->rows_cap, 16), b->rows_cap, b->nvars);
    if (!side)  // (otherwise every parent carries its scan in its own buffer: k_bb_finish)
        hipLaunchKernelGGL(k_bb_basic_scan, dim3((cols_max + 63) / 64, nparents), dim3(64), 0, st,
                           b->d_slots, b->ld, b->bflag, b->bkey);
    // (:799 RoundTableau and the -0 pass of :307-313 are applied by child_init / eliminate as they
    // write: no separate pass over the children)
    const size_t elim_lds = (size_t)b->ld * sizeof(int);  // <= kBBEliminateLdsMax (lpr_bb_create)
    if (elim_lds > (48u << 10)) {
        static bool raised = false;  // one process = one GPU
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bb_eliminate),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kBBEliminateLdsMax);
            raised = true;
        }
    }
    hipLaunchKernelGGL(k_bb_eliminate, dim3(nslots), dim3(1024), elim_lds, st,
                       b->d_slots, b->ld, b->bflag, b->bkey, b->blist, side ? b->rows_cap : -1);
}

void bb_launch_pivot_step(lpr_bb* b, int nslots, int rows_max, int cols_max, int step_no) {
    hipStream_t st = b->eng->stream;
    const int threads = (rows_max > 256 || cols_max > 256) ? 1024 : 256;
    hipLaunchKernelGGL(k_bb_select, dim3(nslots), dim3(threads), 0, st, b->d_slots, b->rowbuf,
                       b->colbuf, b->ld, b->rows_cap, b->trace, b->trace_cap, b->d_running, step_no,
                       b->rowlist, b->touched, align_up(b->rows_cap, 16));
    constexpr int TR = 8;
    hipLaunchKernelGGL((k_bb_update<TR>), dim3(1, (rows_max + TR - 1) / TR, nslots), dim3(256), 0, st,
                       b->d_slots, b->rowbuf, b->colbuf, b->rowlist, b->ld, b->rows_cap);
}

}  // namespace lpr
