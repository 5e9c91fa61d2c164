// revised_kernels.hip -- gfx950 kernels of the revised primal simplex
// (reference: LPR_381_Group_V22/Simplex/RevisedPrimalSimplexSolver.cs).
//
// Every sum the C# forms is a sequential ascending-index loop `s += a * b` (product rounded, then
// the add); the pivot decisions depend on those bits, so the kernels keep that order: one lane
// owns one output element and walks its row / column serially, parallelism comes from the
// independent outputs.  The two selections are the C#'s EPS-band sequential folds, evaluated
// exactly by a "next take" search (see fold comments below).
//
//   k_rev_rowsum   MultiplyMatrixVector  :398-410   x_B = B^-1 b (:89), u = B^-1 a_e (:150)
//   k_rev_colsum   MultiplyVectorMatrix  :412-424   y = c_B B^-1 (:93); rc_j = c_j - y.A_j (:96-98)
//   k_rev_enter    entering fold (:105-121), then GetColumn(A, e) / GetColumn(BInverse, k)
//                  (:149-151, :390-396) in the same launch
//   k_rev_ratio    ratio-test fold (:154-176), bookkeeping (:194-212), eta factors (:266-272)
//   k_rev_update   UpdateBInverse = E * B^-1 (:264-275 via MultiplyMatrices :426-441), in place
//   k_rev_extract  ExtractSolution (:277-287)
//   k_rev_gemm     MultiplyMatrices(BInverse, A) (:360) on fp64 MFMA (mfma_f64_16x16x4)
#include "engine_common.hpp"
#include "revised_common.hpp"
#include "fold_common.hpp"
#include "revised_select.hpp"

#pragma clang fp contract(off)

namespace lpr {

constexpr double kEps = kRevEps;  // RevisedPrimalSimplexSolver.cs:12

// ------------------------------------------------------------------------------------------
// out[i] = sum_{j asc} M[i, j] * v[j], s starts at +0.0  (MultiplyMatrixVector :398-410).
// The sum of one row is a serial chain of m rounded multiply-adds -- that order is what the C#
// computes and what the pivot decisions depend on -- so its floor is the fp64 add latency times m.
// Everything else is made parallel: a 256-thread workgroup owns RB = 16 rows; ALL its lanes stream
// the rows in KC-column chunks with 16-byte coalesced loads into a double-buffered LDS tile while
// 16 lanes (one per row) walk the previous chunk out of LDS in order.  m/16 workgroups keep every
// CU busy pulling its own 16 rows (a CU can only pull ~50 GB/s; one lane per row left 3/4 of the
// chip idle).  `skip_if_slack`: the u = B^-1 a_e launch is a no-op when the entering variable is
// a slack (then u is a column of B^-1, written by k_rev_enter).
constexpr int kGRP = 16;   // operands fetched from LDS one group ahead of the add chain
constexpr int kRB = 16;    // rows per workgroup
constexpr int kKC = 256;   // columns per chunk (a chunk costs max(walk, one memory round trip): 128 was
                           // shorter than the round trip)

// Two products in ONE pass over M: lanes 0..15 walk the rows against v (x_B = B^-1 b), lanes
// 16..31 of the same wave walk the same rows against v2 (u = B^-1 a_e) -- the second walk rides in
// the SIMD lanes the first leaves idle, so B^-1 is read once per iteration for both (the C# calls
// MultiplyMatrixVector twice, :89 and :150; the sums and their order are the same).  v2 is skipped
// when there is no entering variable or it is a slack (u is then a column of B^-1, k_rev_enter).
__global__ __launch_bounds__(256) void k_rev_rowsum(const double* __restrict__ M, int ld, int m,
                                                    const double* __restrict__ v,
                                                    double* __restrict__ out,
                                                    const double* __restrict__ v2,
                                                    double* __restrict__ out2,
                                                    const RevState* __restrict__ st, int n) {
    // The tiles in LDS hold the PRODUCTS M[i, j] * v[j] (and M[i, j] * v2[j]), each rounded as the
    // C# rounds it before adding it (:406): the lanes that stage a chunk multiply as they store, the
    // walking lanes run only the chain of adds (see k_rev_colsum).  +2 doubles per row: rows stay
    // 16-byte aligned and the 16 consumer lanes hit disjoint banks (kGRP more so that the
    // one-group-ahead prefetch of the last group stays inside the array).
    extern __shared__ __attribute__((aligned(16))) double rev_dyn_lds[];
    typedef double Tile[kRB][kKC + kGRP + 2];
    Tile* sP = reinterpret_cast<Tile*>(rev_dyn_lds);  // [2 products][2 buffers]: sP[p * 2 + buf]
    if (st->status != kRunning) return;
    const bool two = v2 != nullptr && st->entering >= 0 && st->entering < n;
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * kRB;
    const int nchunk = (m + kKC - 1) / kKC;
    // staging map: kRB * kKC / 2 double2 per chunk, NQ per lane; a lane's double2 sit in the same
    // two columns of NQ rows, so it needs ONE double2 of v (and of v2).  (Leaving the walking wave
    // out of the staging, as k_rev_colsum does on B^-1, was slower here: 37 -> 41 us with 128
    // staging lanes.)
    constexpr int HK = kKC / 2;           // double2 per row of the chunk
    constexpr int NQ = kRB * HK / 256;    // per lane
    constexpr int RS = 256 / HK;          // rows covered by one pass of the 256 lanes
    double2 rm[NQ];
    double2 rv = make_double2(0.0, 0.0), rv2 = make_double2(0.0, 0.0);
    auto load_chunk = [&](int k0) {
        const int gk = k0 + (tid % HK) * 2;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int gi = row0 + tid / HK + q * RS;
            rm[q] = (gi < m && gk < ld)
                        ? *reinterpret_cast<const double2*>(M + (size_t)gi * ld + gk)
                        : make_double2(0.0, 0.0);
        }
        rv.x = (gk < m) ? v[gk] : 0.0;
        rv.y = (gk + 1 < m) ? v[gk + 1] : 0.0;
        if (two) {
            rv2.x = (gk < m) ? v2[gk] : 0.0;
            rv2.y = (gk + 1 < m) ? v2[gk + 1] : 0.0;
        }
    };
    auto store_chunk = [&](int buf) {
        const int k = (tid % HK) * 2;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int r = tid / HK + q * RS;
            sP[buf][r][k] = rm[q].x * rv.x;  // product rounded ...
            sP[buf][r][k + 1] = rm[q].y * rv.y;
            if (two) {
                sP[2 + buf][r][k] = rm[q].x * rv2.x;
                sP[2 + buf][r][k + 1] = rm[q].y * rv2.y;
            }
        }
    };
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    double s = 0.0;
    const bool walker = tid < kRB || (two && tid < 2 * kRB);
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) load_chunk((c + 1) * kKC);  // in flight under the serial walk below
        if (walker) {
            // columns past m hold products with zero-filled operands; the walk stops at kmax
            const int kmax = min(kKC, m - c * kKC);
            const double* __restrict__ row = sP[(tid / kRB) * 2 + buf][tid & (kRB - 1)];
            if (kmax == kKC) {
                double a[kGRP];
#pragma unroll
                for (int u = 0; u < kGRP; ++u) a[u] = row[u];
#pragma unroll
                for (int k0 = 0; k0 < kKC; k0 += kGRP) {
                    double na[kGRP];
#pragma unroll
                    for (int u = 0; u < kGRP; ++u) na[u] = row[k0 + kGRP + u];
                    __builtin_amdgcn_sched_barrier(0);  // next group in flight under the adds
#pragma unroll
                    for (int u = 0; u < kGRP; ++u) s = s + a[u];  // ... then added (:406)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < kGRP; ++u) a[u] = na[u];
                }
            } else {
                for (int k = 0; k < kmax; ++k) s = s + row[k];
            }
        }
        if (c + 1 < nchunk) store_chunk(buf ^ 1);
        __syncthreads();
    }
    if (tid < kRB && row0 + tid < m) out[row0 + tid] = s;
    if (two && tid >= kRB && tid < 2 * kRB && row0 + tid - kRB < m) out2[row0 + tid - kRB] = s;
}

// ------------------------------------------------------------------------------------------
// s_j = sum_{i asc} v[i] * M[i, j], s starts +0.0  (MultiplyVectorMatrix :412-424, and Dot(y,
// GetColumn(A, j)) :98 -- y[i] * col[i] is the same product, IEEE multiplication commutes).
//   mode 0: out[j] = s_j                (y = c_B B^-1)
//   mode 1: out[j] = c[j] - s_j         (rc_j = c_j - Dot(y, A_j))
// Same producer/consumer shape as k_rev_rowsum, transposed: a workgroup owns CB = 16 columns (one
// 128-byte line per row), all 256 lanes stream RC-row chunks into LDS, 16 lanes (one per column)
// walk the chunk in row order.  cols/16 workgroups (512 for A at n = 8192).
constexpr int kCB = 16;    // columns per workgroup on B^-1
constexpr int kCBA = 32;   // columns per workgroup on A
constexpr int kRCB = 512;  // rows per chunk on B^-1: a chunk costs max(walk, one memory round trip), and the
                           // add-only walk of 128 rows (0.45 us) is shorter than the round trip (~1.2 us)
constexpr int kRC = 128;   // rows per chunk

// CB columns per workgroup (16 per walking wave, CB / 16 waves walk side by side on their own SIMDs).
// On A (268 MB, from HBM) CB = 32: with one 128-byte line per row and workgroup every DRAM page that
// is opened serves 128 bytes, and the kernel ran at 3.3 TB/s with or without the walk (measured:
// deeper prefetch, 256-row chunks, padded strides, the walk removed -- all 79-89 us); 256 bytes per
// row halve the activations per byte.  On B^-1 (134 MB, Infinity-Cache resident) CB = 16 keeps twice
// the workgroups and the chain is the bound either way.
template <int RC, int CB>
__global__ __launch_bounds__(256) void k_rev_colsum(const double* __restrict__ M, int ld, int rows,
                                                    int cols, const double* __restrict__ v,
                                                    const double* __restrict__ c,
                                                    double* __restrict__ out, int mode,
                                                    const RevState* __restrict__ st) {
    // The tile in LDS holds the PRODUCTS v[i] * M[i, j] (each rounded, as the C# rounds it before it
    // adds it, :420): the 240 lanes that stage a chunk multiply as they store, so the walking lanes
    // run nothing but the chain of adds -- one LDS operand and one dependent v_add_f64 per step
    // (with the multiply in the walk a step cost ~25 cycles: 47 us for 4096 rows).
    extern __shared__ __attribute__((aligned(16))) double rev_dyn_lds[];
    typedef double Tile[RC + kGRP][CB];
    Tile* sP = reinterpret_cast<Tile*>(rev_dyn_lds);  // [2]
    if (st->status != kRunning) return;
    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * CB;
    const int nchunk = (rows + RC - 1) / RC;
    // staging map: RC * CB / 2 double2 per chunk, CB / 2 double2 per row, dealt over the NS staging
    // lanes.  With one walking wave (CB = 16, the chain-bound case) that wave does NOT stage: its
    // instruction stream is the chain, and the loads, multiplies and LDS writes of a chunk would
    // sit between two walks.
    constexpr int kStageBase = (CB == 16) ? 64 : 0;
    constexpr int NS = 256 - kStageBase;
    constexpr int H = CB / 2;
    constexpr int NQ = (RC * H + NS - 1) / NS;
    const int sid = tid - kStageBase;
    double2 rm[NQ];
    double rv[NQ];
    auto load_chunk = [&](int i0) {
        if (sid < 0) return;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = sid + q * NS;
            const int r = idx / H, cc = (idx % H) * 2;
            const int gi = i0 + r;
            // A (CB = 32) is read once per iteration and is larger than the Infinity Cache: its
            // stream is marked non-temporal so that it does not push B^-1 (134 MB, read by the two
            // passes either side of this one) out of the cache
            typedef double v2d_t __attribute__((ext_vector_type(2)));
            const double* gp = M + (size_t)gi * ld + j0 + cc;
            if (idx < RC * H && gi < rows && j0 + cc < ld) {
                if (CB == 32) {
                    const v2d_t t = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(gp));
                    rm[q] = make_double2(t.x, t.y);
                } else {
                    rm[q] = *reinterpret_cast<const double2*>(gp);
                }
            } else {
                rm[q] = make_double2(0.0, 0.0);
            }
            rv[q] = (idx < RC * H && gi < rows) ? v[gi] : 0.0;
        }
    };
    auto store_chunk = [&](int buf) {
        if (sid < 0) return;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = sid + q * NS;
            const int r = idx / H, cc = (idx % H) * 2;
            double2 p;
            p.x = rv[q] * rm[q].x;  // v[i] * M[i, j] rounded ...
            p.y = rv[q] * rm[q].y;
            if (idx < RC * H) *reinterpret_cast<double2*>(&sP[buf][r][cc]) = p;
        }
    };
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    double s = 0.0;
    const int wl = tid & 63;
    const bool walker = (tid >> 6) < CB / 16 && wl < 16;
    const int wc = (tid >> 6) * 16 + wl;  // this walker's column of the strip
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunk) load_chunk((ch + 1) * RC);
        if (walker) {
            // rows past the end of a ragged last chunk hold +0.0 * M = products of zero-filled
            // operands: the walk stops at imax
            const int imax = min(RC, rows - ch * RC);
            if (imax == RC) {
                double a[kGRP];
#pragma unroll
                for (int u = 0; u < kGRP; ++u) a[u] = sP[buf][u][wc];
#pragma unroll
                for (int i0 = 0; i0 < RC; i0 += kGRP) {
                    double na[kGRP];
#pragma unroll
                    for (int u = 0; u < kGRP; ++u) na[u] = sP[buf][i0 + kGRP + u][wc];
                    // the next group's LDS reads are issued BEFORE this group's adds (left alone
                    // the scheduler sinks them to their first use: an LDS round trip per group,
                    // 20 cycles per step instead of the 10 a dependent add takes)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < kGRP; ++u) s = s + a[u];  // ... then added (:420)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < kGRP; ++u) a[u] = na[u];
                }
            } else {
                for (int i = 0; i < imax; ++i) s = s + sP[buf][i][wc];
            }
        }
        if (ch + 1 < nchunk) store_chunk(buf ^ 1);
        __syncthreads();
    }
    if (walker && j0 + wc < cols) out[j0 + wc] = mode ? (c[j0 + wc] - s) : s;
}

// ------------------------------------------------------------------------------------------
// The two selections (revised_select.hpp) as launches of their own: lpr_revised_step's path.
__global__ __launch_bounds__(1024) void k_rev_enter(const double* __restrict__ rcx,
                                                    const double* __restrict__ y,
                                                    const uint8_t* __restrict__ is_basic, int n,
                                                    int m, RevState* st,
                                                    const double* __restrict__ A, int lda,
                                                    const double* __restrict__ Binv, int ldb,
                                                    double* __restrict__ acol,
                                                    double* __restrict__ u) {
    if (st->status != kRunning) return;
    rev_enter_body<false, 16>(rcx, y, is_basic, n, m, st, A, lda, Binv, ldb, acol, u);  // A = At here
}

__global__ __launch_bounds__(1024) void k_rev_ratio(const double* __restrict__ u,
                                                    const double* __restrict__ xB,
                                                    int32_t* __restrict__ basic,
                                                    uint8_t* __restrict__ is_basic,
                                                    double* __restrict__ cB,
                                                    const double* __restrict__ c,
                                                    const double* __restrict__ Binv, int ldb,
                                                    double* __restrict__ browbuf,
                                                    double* __restrict__ fac,
                                                    int32_t* __restrict__ log, int n, int m,
                                                    RevState* st) {
    __shared__ double s_rat[kRatioLds];
    __shared__ int s_bvi[kRatioLds];
    if (st->status != kRunning) return;
    rev_ratio_body<false>(u, xB, basic, is_basic, cB, c, Binv, ldb, browbuf, fac, log, n, m, st,
                          s_rat, s_bvi);
}

// ------------------------------------------------------------------------------------------
// UpdateBInverse (:264-275): BInverse = MultiplyMatrices(E, BInverse), E = I with column r
// replaced by fac.  MultiplyMatrices (:426-441) starts every R[i,j] at +0.0 and adds the terms of
// the non-skipped k in ascending order, so for row i the result is
//     i <  r : (0.0 + 1.0*B[i,j]) + fac_i*B[r,j]        (second term skipped if |fac_i| < EPS)
//     i >  r : (0.0 + fac_i*B[r,j]) + 1.0*B[i,j]        (first term skipped if |fac_i| < EPS)
//     i == r :  0.0 + fac_r*B[r,j]                      (0.0 if |fac_r| < EPS)
// Same streaming shape as the tableau update: 16 B/lane, pivot-row slice in registers.
template <int TR>
__global__ __launch_bounds__(256) void k_rev_update(double* __restrict__ Binv, int ld, int m,
                                                    const double* __restrict__ browbuf,
                                                    const double* __restrict__ fac,
                                                    const RevState* __restrict__ st) {
    if (st->status != kRunning) return;
    const int r = st->leaving_row;
    const int ld2 = ld >> 1;
    const int c2 = blockIdx.x * 256 + threadIdx.x;
    if (c2 >= ld2) return;
    const double2 br = reinterpret_cast<const double2*>(browbuf)[c2];
    double2* __restrict__ B2 = reinterpret_cast<double2*>(Binv);
    const int i0 = blockIdx.y * TR;
    double2 x[TR];
#pragma unroll
    for (int k = 0; k < TR; ++k)
        if (i0 + k < m) x[k] = B2[(size_t)(i0 + k) * ld2 + c2];
#pragma unroll
    for (int k = 0; k < TR; ++k) {
        const int i = i0 + k;
        if (i < m) {
            const double f = fac[i];
            const bool use = !(fabs(f) < kEps);
            const double px = f * br.x;
            const double py = f * br.y;
            double2 o;
            if (i == r) {
                o.x = use ? 0.0 + px : 0.0;
                o.y = use ? 0.0 + py : 0.0;
            } else if (i < r) {
                const double tx = 0.0 + x[k].x;  // 1.0 * B[i,j] is exact
                const double ty = 0.0 + x[k].y;
                o.x = use ? tx + px : tx;
                o.y = use ? ty + py : ty;
            } else {
                const double tx = use ? 0.0 + px : 0.0;
                const double ty = use ? 0.0 + py : 0.0;
                o.x = tx + x[k].x;
                o.y = ty + x[k].y;
            }
            // padding columns (j >= m) stay 0: browbuf is 0 there and x is 0
            B2[(size_t)i * ld2 + c2] = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// ExtractSolution (:277-287): x[basic[i]] = Math.Max(0.0, xB[i]) for structural basics,
// finalZ = Dot(cOrig, x) -- a sequential sum, done by one lane.
__global__ __launch_bounds__(256) void k_rev_extract(const int32_t* __restrict__ basic,
                                                     const double* __restrict__ xB,
                                                     const double* __restrict__ cOrig, int n,
                                                     int m, double* __restrict__ x,
                                                     double* __restrict__ z) {
    const int tid = threadIdx.x;
    for (int j = tid; j < n; j += blockDim.x) x[j] = 0.0;
    __syncthreads();
    for (int i = tid; i < m; i += blockDim.x) {
        const int v = basic[i];
        if (v < n) {
            const double xb = xB[i];
            x[v] = (0.0 > xb) ? 0.0 : xb;  // .NET Framework Math.Max(0.0, xb)
        }
    }
    __syncthreads();
    __threadfence_block();
    if (tid == 0) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) {
            const double p = cOrig[j] * x[j];
            s = s + p;
        }
        *z = s;
    }
}

// ------------------------------------------------------------------------------------------
// What CaptureSnapshot (:294-387) shows of the PRE-pivot state, kept before k_rev_ratio changes the
// basis: ratios_pre[i] = xB_i / u_i where u_i > EPS, else +inf (:159-175); the pre-pivot basis
// (:186); the entering variable's reduced cost (:189-191).
__global__ __launch_bounds__(256) void k_rev_snap_pre(const double* __restrict__ u,
                                                      const double* __restrict__ xB,
                                                      const int32_t* __restrict__ basic,
                                                      const double* __restrict__ rcx,
                                                      const double* __restrict__ y, int n, int m,
                                                      double* __restrict__ ratios,
                                                      int32_t* __restrict__ basis_pre,
                                                      double* __restrict__ scal,
                                                      const RevState* __restrict__ st) {
    if (st->status != kRunning || st->entering < 0) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        const int e = st->entering;
        scal[0] = (e < n) ? rcx[e] : -y[e - n];
    }
    if (i >= m) return;
    const double ui = u[i];
    ratios[i] = (ui > kEps) ? ieee_div(xB[i], ui) : (double)INFINITY;
    basis_pre[i] = basic[i];
}

// zWorking = Dot(cB, xB) (:245, :141): a sequential sum, one lane.
__global__ void k_rev_dot(const double* __restrict__ a, const double* __restrict__ b, int len,
                          double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < len; ++i) {
        const double p = a[i] * b[i];
        s = s + p;
    }
    *out = s;
}

// MultiplyMatrices(BInverse, A) (:360, :426-441) in the C#'s own order, for the PRINTED table of
// small models: R[i,j] starts at +0.0 and receives aik * B[k,j] for k ascending, skipping
// |aik| < EPS; product rounded, then the add.  One lane per output element.  (The MFMA product
// k_rev_gemm associates differently; at 3 printed decimals that can flip a digit on a tie.)
__global__ __launch_bounds__(256) void k_rev_matmul_exact(const double* __restrict__ Binv, int ldb,
                                                          const double* __restrict__ A, int lda,
                                                          double* __restrict__ C, int ldc, int m,
                                                          int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n || i >= m) return;
    double s = 0.0;
    for (int k = 0; k < m; ++k) {
        const double aik = Binv[(size_t)i * ldb + k];
        if (fabs(aik) < kEps) continue;
        const double p = aik * A[(size_t)k * lda + j];
        s = s + p;
    }
    C[(size_t)i * ldc + j] = s;
}

// ------------------------------------------------------------------------------------------
// Synthetic dense LP for the benchmark (same generator as the tableau form).
__device__ __forceinline__ uint64_t rsplitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double ru01(uint64_t seed, uint64_t stream, uint64_t i, uint64_t j) {
    uint64_t k = rsplitmix64(seed ^ (stream * 0xD1B54A32D192ED03ULL));
    k = rsplitmix64(k + i);
    k = rsplitmix64(k + j);
    return (double)(k >> 11) * 0x1.0p-53;
}

__global__ __launch_bounds__(256) void k_rev_synthetic(double* __restrict__ A, int lda, int m,
                                                       int n, uint64_t seed,
                                                       double* __restrict__ b,
                                                       double* __restrict__ c,
                                                       double* __restrict__ cOrig) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < lda) A[(size_t)i * lda + j] = (j < n) ? ru01(seed, 0, (uint64_t)i, (uint64_t)j) : 0.0;
    if (j == 0) {
        const double u = ru01(seed, 1, (uint64_t)i, 0);
        const double s = u * 0.1;
        const double t = 1.0 + s;
        b[i] = ((double)n * 0.25) * t;
    }
    if (i == 0 && j < n) {
        const double v = ru01(seed, 2, 0, (uint64_t)j);
        c[j] = v;
        cOrig[j] = v;
    }
}

// At[j, i] = A[i, j] through a 32 x 32 LDS tile (once per solver, after A is in place).
__global__ __launch_bounds__(256) void k_rev_transpose(const double* __restrict__ A, int lda,
                                                       double* __restrict__ At, int ldt, int m,
                                                       int n) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int i = i0 + ty + k, j = j0 + tx;
        tile[ty + k][tx] = (i < m && j < n) ? A[(size_t)i * lda + j] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int j = j0 + ty + k, i = i0 + tx;
        if (j < n && i < ldt) At[(size_t)j * ldt + i] = tile[tx][ty + k];
    }
}

__global__ __launch_bounds__(256) void k_rev_init(double* __restrict__ Binv, int ldb, int m, int n,
                                                  int32_t* __restrict__ basic,
                                                  uint8_t* __restrict__ is_basic) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // :72-79
    if (i < m) {
        Binv[(size_t)i * ldb + i] = 1.0;
        basic[i] = n + i;
    }
    if (i < n + m) is_basic[i] = (i >= n) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// B^-1 * A on the matrix cores:  C[m x n] = Bz[m x m] * A[m x n],  Bz = B^-1 with the entries the
// C# skips (|b_ik| < 1e-9, MultiplyMatrices :436) replaced by 0.  fp64 MFMA 16x16x4
// (v_mfma_f64_16x16x4_f64): per wave-instruction A-operand lane l holds Bz[i = l & 15][k = l >> 4],
// B-operand lane l holds A[k = l >> 4][j = l & 15], the 16x16 result has 4 f64 per lane at
// col = l & 15, row = (l >> 4) + 4 * reg  (cdna_hip_programming.md section 3, f64 map).
// Block = 256 threads = 4 waves, block tile 128 x 128, each wave 64 x 64 = 4 x 4 MFMA tiles,
// K stepped by 16 through LDS (double-buffered).  The sum over k is formed per 4-wide MFMA step
// with FMA accumulation, i.e. NOT the C#'s rounded-product sequential sum: this product feeds only
// the printed snapshot tableau (3 decimals), so its parity bar is a tolerance, not bits.
typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int GM = 128, GN = 128, GK = 16;
constexpr int LDA_S = GK + 1;   // Bz tile stored [row][k], padded: the A-operand read walks rows
constexpr int LDB_S = GN + 16;  // A tile stored [k][col]; +16 doubles = 32 banks: k and k+1 rows
                                // of one ds_read_b64 land on disjoint bank halves

// XCD-aware tile order (8 XCDs, each with its own L2): workgroups are dealt round-robin over the
// XCDs by the dispatcher, so the linear id is first split into (xcd, slot); each XCD then walks a
// contiguous share of the tile grid in "grouped" order (GROUP_M row tiles per column sweep) so
// that the B^-1 row panels and A column panels it touches stay resident in ITS 4 MiB L2.
__device__ __forceinline__ void gemm_tile_of_block(int& tm, int& tn, int gm, int gn) {
    const int nblk = gm * gn;
    int bid = blockIdx.x;
    constexpr int NXCD = 8;
    if (nblk % NXCD == 0) {
        const int per = nblk / NXCD;
        bid = (bid % NXCD) * per + bid / NXCD;
    }
    constexpr int GROUP_M = 4;
    const int width = GROUP_M * gn;
    const int group = bid / width;
    const int first_m = group * GROUP_M;
    const int gsz = min(gm - first_m, GROUP_M);
    tm = first_m + (bid % width) % gsz;
    tn = (bid % width) / gsz;
}

__global__ __launch_bounds__(256, 2) void k_rev_gemm(const double* __restrict__ Binv, int ldb,
                                                     const double* __restrict__ A, int lda,
                                                     double* __restrict__ Cout, int ldc, int m,
                                                     int n, int gm, int gn) {
    __shared__ __attribute__((aligned(16))) double sA[2][GM * LDA_S];
    __shared__ __attribute__((aligned(16))) double sB[2][GK * LDB_S];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = (wave >> 1) * 64;  // wave tile origin inside the block tile
    const int wn = (wave & 1) * 64;
    int tm, tn;
    gemm_tile_of_block(tm, tn, gm, gn);
    const int bm = tm * GM;
    const int bn = tn * GN;

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // Staging maps (16-byte global loads; rows are 128-byte aligned, ld % 16 == 0):
    //   B^-1 tile 128 x 16: 1024 double2, 4 per lane: row = idx >> 3, k pair = idx & 7
    //   A    tile  16 x 128: 1024 double2, 4 per lane: k   = idx >> 6, col pair = idx & 63
    double2 ra[4], rb[4];
    auto load_tile = [&](int k0) {  // global -> registers; no wait here
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + q * 256;
            const int r = idx >> 3, kk = (idx & 7) * 2;
            const int gi = bm + r, gk = k0 + kk;
            ra[q] = (gi < m && gk < ldb)
                        ? *reinterpret_cast<const double2*>(Binv + (size_t)gi * ldb + gk)
                        : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + q * 256;
            const int kk = idx >> 6, cc = (idx & 63) * 2;
            const int gk = k0 + kk, gj = bn + cc;
            rb[q] = (gk < m && gj < lda)
                        ? *reinterpret_cast<const double2*>(A + (size_t)gk * lda + gj)
                        : make_double2(0.0, 0.0);
        }
    };
    auto store_tile = [&](int buf, int k0) {  // registers -> LDS
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + q * 256;
            const int r = idx >> 3, kk = (idx & 7) * 2;
            double vx = ra[q].x, vy = ra[q].y;
            if (k0 + kk >= m || fabs(vx) < kEps) vx = 0.0;      // :436 zero-skip; k beyond m is 0
            if (k0 + kk + 1 >= m || fabs(vy) < kEps) vy = 0.0;
            sA[buf][r * LDA_S + kk] = vx;
            sA[buf][r * LDA_S + kk + 1] = vy;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + q * 256;
            const int kk = idx >> 6, cc = (idx & 63) * 2;
            double vx = rb[q].x, vy = rb[q].y;
            if (bn + cc >= n) vx = 0.0;      // padding columns of A are 0 anyway
            if (bn + cc + 1 >= n) vy = 0.0;
            *reinterpret_cast<double2*>(&sB[buf][kk * LDB_S + cc]) = make_double2(vx, vy);
        }
    };

    const int nk = (m + GK - 1) / GK;
    load_tile(0);
    store_tile(0, 0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        if (t + 1 < nk) load_tile((t + 1) * GK);  // in flight under the 64 MFMAs below
#pragma unroll
        for (int ks = 0; ks < GK; ks += 4) {
            double af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
                af[a] = sA[buf][(wm + a * 16 + (lane & 15)) * LDA_S + ks + (lane >> 4)];
#pragma unroll
            for (int b = 0; b < 4; ++b)
                bf[b] = sB[buf][(ks + (lane >> 4)) * LDB_S + wn + b * 16 + (lane & 15)];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0,
                                                                     0, 0);
        }
        if (t + 1 < nk) store_tile(buf ^ 1, (t + 1) * GK);
        __syncthreads();
    }

#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int gi = bm + wm + a * 16 + (lane >> 4) + 4 * rg;
                const int gj = bn + wn + b * 16 + (lane & 15);
                if (gi < m && gj < n) Cout[(size_t)gi * ldc + gj] = acc[a][b][rg];
            }
}

// ------------------------------------------------------------------------------------------
// launchers

// dynamic LDS of k_rev_rowsum: two product tiles x two buffers
// (the attribute belongs to the device the call is made on: asked once per device, not per process)
static bool rev_first_ask(unsigned long long (&mask)[4]) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 256) return true;
    const unsigned long long bit = 1ull << (dev & 63);
    if (mask[dev >> 6] & bit) return false;
    mask[dev >> 6] |= bit;
    return true;
}

static size_t rev_rowsum_lds() {
    constexpr size_t bytes = (size_t)4 * kRB * (kKC + kGRP + 2) * sizeof(double);
    static unsigned long long asked[4] = {0, 0, 0, 0};
    if (rev_first_ask(asked))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rev_rowsum),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return bytes;
}

// dynamic LDS of k_rev_colsum<kRC, CB> (CB = 32: above the 64 KB a kernel gets without asking)
template <int RC, int CB>
static size_t rev_colsum_lds() {
    constexpr size_t bytes = (size_t)(2 * (RC + kGRP) * CB) * sizeof(double);
    static unsigned long long asked[4] = {0, 0, 0, 0};
    if (rev_first_ask(asked))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rev_colsum<RC, CB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return bytes;
}

// x_B = B^-1 b, y = c_B B^-1, rc_j = c_j - y.A_j: the head of an iteration (:89-102) and, with the
// same operands after the pivot, the post-pivot quantities of the snapshot (:217-227).
void rev_launch_prices(lpr_revised* s) {
    hipStream_t st = s->eng->stream;
    const int m = s->m, n = s->n;
    // x_B = B^-1 b (:89)
    hipLaunchKernelGGL(k_rev_rowsum, dim3((m + kRB - 1) / kRB), dim3(256), rev_rowsum_lds(), st, s->Binv, s->ldb,
                       m, s->b, s->xB, (const double*)nullptr, (double*)nullptr, s->state, n);
    // y = c_B B^-1 (:93)
    hipLaunchKernelGGL((k_rev_colsum<kRCB, kCB>), dim3((m + kCB - 1) / kCB), dim3(256), (rev_colsum_lds<kRCB, kCB>()), st, s->Binv, s->ldb,
                       m, m, s->cB, (const double*)nullptr, s->y, 0, s->state);
    // rc_j = c_j - y.A_j (:96-98)
    hipLaunchKernelGGL((k_rev_colsum<kRC, kCBA>), dim3((n + kCBA - 1) / kCBA), dim3(256), (rev_colsum_lds<kRC, kCBA>()), st, s->A, s->lda, m,
                       n, s->y, s->c, s->rcx, 1, s->state);
}

void rev_launch_zworking(lpr_revised* s) {
    hipLaunchKernelGGL(k_rev_dot, dim3(1), dim3(64), 0, s->eng->stream, s->cB, s->xB, s->m,
                       s->snap_scal + 1);
}

void rev_launch_matmul_exact(lpr_revised* s, double* Cout, int ldc) {
    hipLaunchKernelGGL(k_rev_matmul_exact, dim3((s->n + 255) / 256, s->m), dim3(256), 0,
                       s->eng->stream, s->Binv, s->ldb, s->A, s->lda, Cout, ldc, s->m, s->n);
}

// One iteration of Solve() (:86-249).  `snapshot`: also keep the pre-pivot ratios / basis /
// entering reduced cost for CaptureSnapshot (s->snap_* must exist).
void rev_launch_iteration(lpr_revised* s, bool snapshot) {
    hipStream_t st = s->eng->stream;
    const int m = s->m, n = s->n;
    // y = c_B B^-1 (:93)
    hipLaunchKernelGGL((k_rev_colsum<kRCB, kCB>), dim3((m + kCB - 1) / kCB), dim3(256), (rev_colsum_lds<kRCB, kCB>()), st, s->Binv, s->ldb,
                       m, m, s->cB, (const double*)nullptr, s->y, 0, s->state);
    // rc_j = c_j - y.A_j (:96-98)
    hipLaunchKernelGGL((k_rev_colsum<kRC, kCBA>), dim3((n + kCBA - 1) / kCBA), dim3(256), (rev_colsum_lds<kRC, kCBA>()), st, s->A, s->lda, m,
                       n, s->y, s->c, s->rcx, 1, s->state);
    hipLaunchKernelGGL(k_rev_enter, dim3(1), dim3(1024), 0, st, s->rcx, s->y, s->is_basic, n, m,
                       s->state, s->At, s->ldb, s->Binv, s->ldb, s->acol, s->u);
    // x_B = B^-1 b (:89) and u = B^-1 a_e (:150; unless the entering variable is a slack) in one
    // pass over B^-1
    hipLaunchKernelGGL(k_rev_rowsum, dim3((m + kRB - 1) / kRB), dim3(256), rev_rowsum_lds(), st, s->Binv, s->ldb,
                       m, s->b, s->xB, s->acol, s->u, s->state, n);
    if (snapshot)
        hipLaunchKernelGGL(k_rev_snap_pre, dim3((m + 255) / 256), dim3(256), 0, st, s->u, s->xB,
                           s->basic, s->rcx, s->y, n, m, s->snap_ratios, s->snap_basis,
                           s->snap_scal, s->state);
    hipLaunchKernelGGL(k_rev_ratio, dim3(1), dim3(1024), 0, st, s->u, s->xB, s->basic,
                       s->is_basic, s->cB, s->c, s->Binv, s->ldb, s->browbuf, s->fac, s->log, n, m,
                       s->state);
    constexpr int TR = 8;
    hipLaunchKernelGGL((k_rev_update<TR>), dim3((s->ldb / 2 + 255) / 256, (m + TR - 1) / TR),
                       dim3(256), 0, st, s->Binv, s->ldb, m, s->browbuf, s->fac, s->state);
}

// revised_fused.hip
void rev_launch_y(lpr_revised* s);
void rev_launch_update_y(lpr_revised* s);
void rev_launch_rc_enter(lpr_revised* s);
void rev_launch_xu_ratio(lpr_revised* s);

// One iteration of lpr_revised_solve's batches.  Precondition: s->y = c_B B^-1 of the current state
// (rev_launch_y at the head of a call; afterwards the update pass of every pivot leaves it).
void rev_launch_iteration_batched(lpr_revised* s) {
    rev_launch_rc_enter(s);   // rc_j = c_j - y.A_j (:96-98) + entering (:105-121) + GetColumn (:149-151)
    rev_launch_xu_ratio(s);   // x_B (:89), u (:150) + exits, ratio test (:154-176), bookkeeping (:194-212)
    rev_launch_update_y(s);   // E * B^-1 (:264-275) + the next iteration's y (:93 = :219)
}

void rev_launch_transpose_a(lpr_revised* s) {
    hipLaunchKernelGGL(k_rev_transpose, dim3((s->n + 31) / 32, (s->ldb + 31) / 32), dim3(256), 0,
                       s->eng->stream, s->A, s->lda, s->At, s->ldb, s->m, s->n);
}

void rev_launch_extract(lpr_revised* s) {
    hipLaunchKernelGGL(k_rev_extract, dim3(1), dim3(256), 0, s->eng->stream, s->basic, s->xB,
                       s->cOrig, s->n, s->m, s->x, s->z);
}

void rev_launch_init(lpr_revised* s) {
    const int tot = s->n + s->m;
    hipLaunchKernelGGL(k_rev_init, dim3((tot + 255) / 256), dim3(256), 0, s->eng->stream, s->Binv,
                       s->ldb, s->m, s->n, s->basic, s->is_basic);
}

void rev_launch_synthetic(lpr_revised* s, uint64_t seed) {
    hipLaunchKernelGGL(k_rev_synthetic, dim3((s->lda + 255) / 256, s->m), dim3(256), 0,
                       s->eng->stream, s->A, s->lda, s->m, s->n, seed, s->b, s->c, s->cOrig);
}

void rev_launch_gemm(lpr_revised* s, double* Cout, int ldc) {
    const int gn = (s->n + GN - 1) / GN, gm = (s->m + GM - 1) / GM;
    hipLaunchKernelGGL(k_rev_gemm, dim3(gm * gn), dim3(256), 0, s->eng->stream, s->Binv, s->ldb,
                       s->A, s->lda, Cout, ldc, s->m, s->n, gm, gn);
}

}  // namespace lpr
