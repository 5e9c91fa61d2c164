// revised_fused.hip -- the batched iteration of the revised primal simplex (lpr_revised_solve) as
// THREE launches instead of six (reference: Simplex/RevisedPrimalSimplexSolver.cs:82-251):
//
//   k_rev_update_y   UpdateBInverse (:264-275: B^-1 <- E * B^-1, in place) and, in the SAME pass over
//                    B^-1, the next iteration's y = c_B B^-1 (:93 / :219, MultiplyVectorMatrix
//                    :412-424) -- every updated element is in a register anyway
//   k_rev_rc_enter   rc_j = c_j - Dot(y, A_j) (:96-98) and, in the workgroup that finishes last, the
//                    entering fold (:105-121) + GetColumn(A, e) / GetColumn(BInverse, k) (:149-151)
//   k_rev_xu_ratio   x_B = B^-1 b (:89) and u = B^-1 a_e (:150) in one pass over B^-1 and, in the
//                    workgroup that finishes last, the loop head's exits (:90-91, :124), the ratio
//                    fold (:154-176), the bookkeeping (:194-212) and the eta column (:266-272)
//
// Every sum keeps the C#'s order: one output = one serial chain of rounded adds of rounded
// products (s += a * b), so a chain costs (fp64 add latency) x m whatever else happens -- 4 096
// rows x ~4.3 ns = 18 us.  The kernels are built around that chain: ONE wave of a workgroup (the
// walker) does nothing but the adds of 16 (or 2 x 16) outputs, reading the products out of an LDS
// ring; the other waves (the stagers) stream the operands from memory, multiply (and, in
// k_rev_update_y, apply E and write B^-1 back) and fill the ring several slots ahead.  Walker and
// stagers meet through two LDS words per slot, never at a workgroup barrier, so the walker never
// waits for a memory round trip (the double-buffered form of revised_kernels.hip paid one per
// chunk: 30 us for the B^-1 column sums where the chain needs 18).
#include "engine_common.hpp"
#include "revised_common.hpp"
#include "revised_select.hpp"

#pragma clang fp contract(off)

namespace lpr {

// ---------------------------------------------------------------------------------------------
// The ring.  A slot holds the products of RC consecutive terms of NOUT outputs, one row of
// RC + kRingPad doubles per output: the walker lane of an output reads its row with ds_read_b128
// (two terms per read).  (RC + 18) * 2 dwords = 36 mod 64 for RC % 32 == 0: the 16 lanes of a
// wave that read 16 rows at the same offset hit 16 disjoint groups of four banks.
constexpr int kRingPad = 18;  // 16: the one-group-ahead prefetch of the last group stays inside

// ready[s]: stager waves that have filled slot s (monotonic: NSW per use of the slot);
// done: chunks the walker has finished (monotonic).
struct RingCtl {
    int ready[4];
    int done;
};

__device__ __forceinline__ int lds_load(const int* p) {
    return __atomic_load_n(p, __ATOMIC_RELAXED);
}

// The walker's side of the ring: one output per lane (`mine`), its products RC at a time.
// A full chunk is walked out of registers that already hold its first 16 products; while the LAST
// 16 are being added, the first 16 of the NEXT chunk are requested (the walker only starts a chunk
// once the one after it has been filled), so the chain never waits for an LDS round trip at a chunk
// boundary (32 boundaries x ~0.1 us in k_rev_xu_ratio).  A ragged chunk -- always the last -- is
// walked element by element.
template <int RC>
struct RingWalker {
    double2 a[8];
    bool primed = false;

    // row: this chunk's products; nrow: the next chunk's (nullptr: none)
    __device__ __forceinline__ double walk(const double* __restrict__ row,
                                           const double* __restrict__ nrow, int len, double s) {
        if (len != RC) {
            for (int k = 0; k < len; ++k) s = s + row[k];
            primed = false;
            return s;
        }
        const double2* __restrict__ r2 = reinterpret_cast<const double2*>(row);
        // (no next chunk: the reads of the last group go to the pad behind the row and are dropped)
        const double2* __restrict__ n2 =
            nrow ? reinterpret_cast<const double2*>(nrow) : r2 + RC / 2;
        if (!primed) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = r2[u];
        }
        // the group in a[] is requested in full before the pattern below starts: that is the
        // distance (8 reads = 16 adds) every later read keeps ahead of its use
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k0 = 0; k0 < RC; k0 += 16) {
            double2 na[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                na[u] = (k0 + 16 < RC) ? r2[(k0 + 16) / 2 + u] : n2[u];
                s = s + a[u].x;  // the C#'s `s += product` (:406, :420, :446)
                s = s + a[u].y;
            }
            // one LDS read between every two dependent adds: the reads issue in the bubbles of
            // the add chain instead of in front of it
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // 2 VALU
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = na[u];
        }
        primed = nrow != nullptr;
        return s;
    }
};

// The walker wave's loop over the chunks of one output per lane.  NOUT rows per slot.
template <int RC, int S, int NSW, int NOUT>
__device__ __forceinline__ double ring_consume(const double* ring, RingCtl& ctl, int nchunk,
                                               int total, int lane, bool mine,
                                               unsigned long long* wdbg = nullptr) {
    constexpr int ROW = RC + kRingPad;
    RingWalker<RC> wk;
    double s = 0.0;
    auto wait_ready = [&](int ch) {
        const int need = NSW * (ch / S + 1);
        while (lds_load(&ctl.ready[ch % S]) < need) __builtin_amdgcn_s_sleep(1);
    };
    if (nchunk > 0) wait_ready(0);
    for (int ch = 0; ch < nchunk; ++ch) {
        if (wdbg && lane == 0 && (ch == 8 || ch == 24))  // diagnostic: 16 chunks of one walker
            wdbg[ch == 8 ? 0 : 1] = __builtin_amdgcn_s_memrealtime();
        if (ch + 1 < nchunk) wait_ready(ch + 1);
        asm volatile("" ::: "memory");
        if (mine) {
            const double* row = ring + ((size_t)(ch % S) * NOUT + lane) * ROW;
            const double* nrow =
                (ch + 1 < nchunk) ? ring + ((size_t)((ch + 1) % S) * NOUT + lane) * ROW : nullptr;
            s = wk.walk(row, nrow, min(RC, total - ch * RC), s);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __atomic_store_n(&ctl.done, ch + 1, __ATOMIC_RELAXED);
    }
    return s;
}

// ---------------------------------------------------------------------------------------------
// k_rev_update_y.  A workgroup owns 16 columns of B^-1 (one 128-byte line per row) and takes all
// m rows through it in order, RC rows per ring slot.
//   stagers (NSW waves): a wave-instruction covers 8 rows x 8 double2; per chunk a lane holds NQ
//     double2 of one column pair (RC = NSW * 8 * NQ).  For each: new = E-row applied (the exact
//     expressions of k_rev_update), stored back in place, and cB[i] * new -- the product the C#
//     rounds before it adds it (:420) -- goes into the ring.
//   walker (wave 0, lanes 0..15): y_j = sum_i products, ascending i, s starts at +0.0.
// do_update == 0: no pivot has been made since y was last formed (first iteration of a call):
// plain y = c_B B^-1, nothing is stored.
// Algorithmic bytes: 2 * 8 * m^2 (B^-1 read once, written once).
template <int NSW, int NQ>
__global__ __launch_bounds__(64 * (NSW + 1)) void k_rev_update_y(
    double* __restrict__ Binv, int ld, int m, const double* __restrict__ browbuf,
    const double* __restrict__ fac, const double* __restrict__ cB, double* __restrict__ y,
    const RevState* __restrict__ st, int do_update, const uint8_t* __restrict__ is_basic, int n,
    double* __restrict__ wmin_y) {
    constexpr int RC = NSW * 8 * NQ;
    constexpr int ROW = RC + kRingPad;
    constexpr int S = 3;  // ring slots
    static_assert(RC % 32 == 0, "bank spreading assumes RC % 32 == 0");
    extern __shared__ __attribute__((aligned(16))) double rev_ring[];  // [S][16][ROW]
    __shared__ RingCtl ctl;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int j0 = blockIdx.x * 16;
    const int nchunk = (m + RC - 1) / RC;
    const int ld2 = ld >> 1;
    // the stagers' first chunk is requested BEFORE the control block is looked at (k_rev_xu_ratio)
    const int w = wave - 1;
    const int cp = lane & 7;       // column pair of the strip
    const int rs = lane >> 3;      // row of a wave-instruction
    const int c2 = (j0 >> 1) + cp;
    const bool col_ok = c2 < ld2;
    double2* __restrict__ B2 = reinterpret_cast<double2*>(Binv);
    double2 x[NQ], xn[NQ];
    double f[NQ], fn[NQ], cb[NQ], cbn[NQ];
    auto load_chunk = [&](int c, double2 (&xx)[NQ], double (&ff)[NQ], double (&cc)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = c * RC + (q * NSW + w) * 8 + rs;
            const bool ok = i < m && col_ok;
            xx[q] = ok ? B2[(size_t)i * ld2 + c2] : make_double2(0.0, 0.0);
            ff[q] = (ok && do_update) ? fac[i] : 0.0;
            cc[q] = ok ? cB[i] : 0.0;
        }
    };
    double2 br = make_double2(0.0, 0.0);
    if (wave != 0) {
        load_chunk(0, x, f, cb);
        if (do_update && col_ok) br = reinterpret_cast<const double2*>(browbuf)[c2];
    }
    const int32_t status0 = st->status;
    const int r = do_update ? st->leaving_row : -1;
    if (status0 != kRunning) return;
    if (tid < 4) ctl.ready[tid] = 0;
    if (tid == 4) ctl.done = 0;
    __syncthreads();

    if (wave == 0) {
        // ---- the walker ----
        __builtin_amdgcn_s_setprio(3);
        const double s = ring_consume<RC, S, NSW, 16>(rev_ring, ctl, nchunk, m, lane, lane < 16);
        // slack k = n + j: rcS_k = -y_j (:100-102), a candidate when it is > EPS; the group's
        // minimum of -rcS = y_j goes to the entering fold of the NEXT k_rev_rc_enter (rev_enter_hier)
        double vmin = INFINITY;
        if (lane < 16 && j0 + lane < m) {
            y[j0 + lane] = s;
            if (is_basic[n + j0 + lane] == 0 && -s > kRevEps) vmin = s;
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            const double o = __shfl_xor(vmin, off, kWave);
            vmin = (o < vmin) ? o : vmin;
        }
        if (lane == 0) wmin_y[blockIdx.x] = vmin;
        return;
    }

    // ---- the stagers ----
    for (int c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) load_chunk(c + 1, xn, fn, cbn);  // in flight under this chunk's work
        const int slot = c % S;
        // the slot was last used by chunk c - S: wait until the walker has finished it
        while (lds_load(&ctl.done) < c - S + 1) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        double* __restrict__ tile = rev_ring + (size_t)slot * 16 * ROW;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int rr = (q * NSW + w) * 8 + rs;  // row inside the chunk
            const int i = c * RC + rr;
            double2 o = x[q];
            if (do_update && i < m) {
                const double fi = f[q];
                const bool use = !(fabs(fi) < kRevEps);
                const double px = fi * br.x;
                const double py = fi * br.y;
                if (i == r) {                       // 0.0 + fac_r * B[r, j]  (0.0 if skipped)
                    o.x = use ? 0.0 + px : 0.0;
                    o.y = use ? 0.0 + py : 0.0;
                } else if (i < r) {                 // (0.0 + 1.0 * B[i, j]) + fac_i * B[r, j]
                    const double tx = 0.0 + x[q].x;
                    const double ty = 0.0 + x[q].y;
                    o.x = use ? tx + px : tx;
                    o.y = use ? ty + py : ty;
                } else {                            // (0.0 + fac_i * B[r, j]) + 1.0 * B[i, j]
                    const double tx = use ? 0.0 + px : 0.0;
                    const double ty = use ? 0.0 + py : 0.0;
                    o.x = tx + x[q].x;
                    o.y = ty + x[q].y;
                }
                if (col_ok) B2[(size_t)i * ld2 + c2] = o;
            }
            tile[(2 * cp) * ROW + rr] = cb[q] * o.x;      // c_B[i] * B^-1[i, j], rounded (:420)
            tile[(2 * cp + 1) * ROW + rr] = cb[q] * o.y;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&ctl.ready[slot], 1, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            x[q] = xn[q];
            f[q] = fn[q];
            cb[q] = cbn[q];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// "Last workgroup done": every workgroup stores its outputs with agent-scope (sc1, write-through)
// stores, waits for them (s_waitcnt vmcnt(0)) and adds 1 to a counter in the control block; the
// workgroup whose add returns nwg - 1 knows every other workgroup's stores have landed and reads
// them with agent-scope loads (MI355X_MICROARCH.md, hand-off table, first row).  Called by ONE
// lane after that wave's wait; the result goes through LDS to the rest of the workgroup.
__device__ __forceinline__ bool arrive_is_last(int32_t* ctr, int nwg) {
    const int old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return old == nwg - 1;
}
__device__ __forceinline__ void st_sc1(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

typedef double rv2d_t __attribute__((ext_vector_type(2)));

// Diagnostic stamps (LPR_REV_STAMPS=1: lpr_revised_solve prints them; nullptr otherwise).  Four
// 64-bit words per kernel: min entry, max end-of-walk, tail start, tail end (100 MHz ticks).
__device__ __forceinline__ void stamp_min(unsigned long long* d, int k) {
    if (d) atomicMin(d + k, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
__device__ __forceinline__ void stamp_max(unsigned long long* d, int k) {
    if (d) atomicMax(d + k, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

// ---------------------------------------------------------------------------------------------
// k_rev_rc_enter.  rc_j = c_j - sum_{i asc} y_i * A[i, j] (:96-98: Dot(y, GetColumn(A, j)), s starts
// at +0.0), 32 columns (256 bytes per row) per workgroup, all m rows in order through the ring;
// then, in the workgroup that arrives last, the entering fold and GetColumn (rev_enter_body).
//   stagers (NSW waves): a wave-instruction covers 4 rows x 16 double2; a lane holds NQ double2 of
//     one column pair per chunk (RC = NSW * 4 * NQ), loads run TWO chunks ahead of the ring fill
//     (A comes from HBM: 268 MB, non-temporal so that it does not evict B^-1 from the Infinity
//     Cache); the product y_i * A[i, j] is rounded as the C# rounds it (:446) before the add.
//   walker (wave 0, lanes 0..31): the 32 chains.
// Algorithmic bytes: 8 * m * n.
template <int NSW, int NQ>
__global__ __launch_bounds__(64 * (NSW + 1)) void k_rev_rc_enter(
    const double* __restrict__ A, int lda, int m, int n, const double* __restrict__ y,
    const double* __restrict__ c, double* __restrict__ rcx, const uint8_t* __restrict__ is_basic,
    const double* __restrict__ Binv, int ldb, double* __restrict__ u, RevState* st,
    unsigned long long* dbg, const double* __restrict__ At, double* __restrict__ wmin) {
    constexpr int RC = NSW * 4 * NQ;
    constexpr int ROW = RC + kRingPad;
    constexpr int S = 4;
    static_assert(RC % 32 == 0, "bank spreading assumes RC % 32 == 0");
    extern __shared__ __attribute__((aligned(16))) double rev_ring[];  // [S][32][ROW]
    __shared__ RingCtl ctl;
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int j0 = blockIdx.x * 32;
    const int nchunk = (m + RC - 1) / RC;
    // the stagers' first two chunks are requested BEFORE the control block is looked at
    // (k_rev_xu_ratio)
    const int w = wave - 1;
    const int cp = lane & 15;   // column pair of the strip
    const int rs = lane >> 4;   // row of a wave-instruction
    const int col = j0 + 2 * cp;
    const bool col_ok = col < lda;
    rv2d_t xb[3][NQ];
    double yb[3][NQ];
    auto load_chunk = [&](int ch, rv2d_t (&xx)[NQ], double (&yy)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = ch * RC + (q * NSW + w) * 4 + rs;
            const bool ok = i < m && col_ok && ch < nchunk;
            xx[q] = ok ? __builtin_nontemporal_load(
                             reinterpret_cast<const rv2d_t*>(A + (size_t)i * lda + col))
                       : (rv2d_t){0.0, 0.0};
            yy[q] = ok ? y[i] : 0.0;
        }
    };
    if (wave != 0) {
        load_chunk(0, xb[0], yb[0]);
        load_chunk(1, xb[1], yb[1]);
    }
    if (st->status != kRunning) return;
    if (tid < 4) ctl.ready[tid] = 0;
    if (tid == 4) ctl.done = 0;
    if (tid == 5) s_last = 0;
    __syncthreads();

    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        if (lane == 0) stamp_min(dbg, 0);
        const double s = ring_consume<RC, S, NSW, 32>(rev_ring, ctl, nchunk, m, lane, lane < 32);
        double vmin = INFINITY;  // min of -rc over this workgroup's candidates (rev_enter_hier)
        if (lane < 32 && j0 + lane < n) {
            const double rc = c[j0 + lane] - s;  // :97
            st_sc1(rcx + j0 + lane, rc);
            if (is_basic[j0 + lane] == 0 && rc > kRevEps) vmin = -rc;
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            const double o = __shfl_xor(vmin, off, kWave);
            vmin = (o < vmin) ? o : vmin;
        }
        if (lane == 0) st_sc1(wmin + blockIdx.x, vmin);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) stamp_max(dbg, 1);
        if (lane == 0 && arrive_is_last(&st->arrive_rc, (int)gridDim.x)) s_last = 1;
    } else {
        auto fill = [&](int ch, const rv2d_t (&xx)[NQ], const double (&yy)[NQ]) {
            const int slot = ch % S;
            while (lds_load(&ctl.done) < ch - S + 1) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            double* __restrict__ tile = rev_ring + (size_t)slot * 32 * ROW;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int rr = (q * NSW + w) * 4 + rs;
                tile[(2 * cp) * ROW + rr] = yy[q] * xx[q].x;      // y[i] * col[i], rounded (:446)
                tile[(2 * cp + 1) * ROW + rr] = yy[q] * xx[q].y;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&ctl.ready[slot], 1, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        for (int ch = 0; ch < nchunk; ch += 3) {   // register sets rotate with the chunk number
            load_chunk(ch + 2, xb[2], yb[2]);
            fill(ch, xb[0], yb[0]);
            if (ch + 1 < nchunk) {
                load_chunk(ch + 3, xb[0], yb[0]);
                fill(ch + 1, xb[1], yb[1]);
            }
            if (ch + 2 < nchunk) {
                load_chunk(ch + 4, xb[1], yb[1]);
                fill(ch + 2, xb[2], yb[2]);
            }
        }
    }
    __syncthreads();
    if (!s_last) return;
    // ---- the entering variable (:105-121) and its column (:149-151), by the last workgroup ----
    __threadfence_block();
    if (tid == 0) stamp_max(dbg, 2);
    // (a structural column is not copied out: k_rev_xu_ratio reads row e of At itself)
    rev_enter_body<true, 32>(rcx, y, is_basic, n, m, st, At, ldb, Binv, ldb, nullptr, u, dbg, rev_ring,
                             S * 32 * ROW, wmin, (n + 31) / 32, (m + 15) / 16);
    if (tid == 0) st->arrive_rc = 0;
    __syncthreads();
    if (tid == 0) stamp_max(dbg, 3);
}

// ---------------------------------------------------------------------------------------------
// k_rev_xu_ratio.  x_B = B^-1 b (:89) and, when the entering variable is structural, u = B^-1 a_e
// (:150) -- MultiplyMatrixVector :398-410, r_i = sum_{j asc} M[i, j] * v[j], s from +0.0 -- in ONE
// pass over B^-1: 16 rows per workgroup, lanes 0..15 of the walker chain the rows against b,
// lanes 16..31 the same rows against a_e.  Then, in the workgroup that arrives last, the loop
// head's exits, the ratio fold, the bookkeeping and the eta column (rev_ratio_body).
//   stagers (kXuNSW waves): a wave-instruction is one row's 128 columns of the chunk (1 KB
//     contiguous); wave w stages 16 / kXuNSW rows; loads run four chunks ahead of the ring fill.
// Algorithmic bytes: 8 * m^2.
constexpr int kXuRC = 128;
constexpr int kXuNSW = 8;   // stager waves (16 / kXuNSW rows each); the tail wants >= 512 threads at m = 4096
__global__ __launch_bounds__(64 * (kXuNSW + 1)) void k_rev_xu_ratio(
    const double* __restrict__ Binv, int ld, int m, int n, const double* __restrict__ b,
    double* __restrict__ xB, const double* __restrict__ At, double* __restrict__ u,
    int32_t* __restrict__ basic, uint8_t* __restrict__ is_basic, double* __restrict__ cB,
    const double* __restrict__ c, double* __restrict__ browbuf, double* __restrict__ fac,
    int32_t* __restrict__ log, RevState* st, unsigned long long* dbg) {
    constexpr int RC = kXuRC, NSW = kXuNSW;
    constexpr int ROW = RC + kRingPad;
    constexpr int S = 4;
    extern __shared__ __attribute__((aligned(16))) double rev_ring[];  // [S][32][ROW]
    __shared__ RingCtl ctl;
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int row0 = blockIdx.x * 16;
    const int nchunk = (m + RC - 1) / RC;
    constexpr int RPW = 16 / NSW;   // rows a stager wave stages: w * RPW ...
    struct Regs {
        double2 r[RPW];   // columns k0 + 2 lane, + 1 of those rows
        double2 bv, av;
    };
    constexpr int D = 4;   // chunks the loads run ahead of the ring fill (D + 1 register sets)
    Regs rg[D + 1];
    const int ld2 = ld >> 1;
    const double2* __restrict__ B2 = reinterpret_cast<const double2*>(Binv);
    // B^-1 and b of a chunk (nothing here depends on the control block)
    auto load_chunk_b = [&](int ch, Regs& g) {
        const int w = wave - 1;
        const int k = ch * RC + 2 * lane;   // first of this lane's two columns
        const bool ok = ch < nchunk && k < ld;
        const double2 z = make_double2(0.0, 0.0);
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int i = row0 + w * RPW + q;
            g.r[q] = (ok && i < m) ? B2[(size_t)i * ld2 + (k >> 1)] : z;
        }
        g.bv.x = (ok && k < m) ? b[k] : 0.0;
        g.bv.y = (ok && k + 1 < m) ? b[k + 1] : 0.0;
        g.av = z;
    };
    // The stagers ask for their first D chunks BEFORE the control block is looked at: the status /
    // entering words are a memory round trip of their own (~1.5 us with every CU starting up at
    // once) and the chain cannot start before the first chunk is in the ring.
    if (wave != 0) {
#pragma unroll
        for (int k = 0; k < D; ++k) load_chunk_b(k, rg[k]);
    }
    const int32_t status0 = st->status;
    const int e = st->entering;
    if (status0 != kRunning) return;
    if (tid < 4) ctl.ready[tid] = 0;
    if (tid == 4) ctl.done = 0;
    if (tid == 5) s_last = 0;
    __syncthreads();
    const bool two = e >= 0 && e < n;   // slack: u = BInverse[:, k], written by the entering tail
    // GetColumn(A, e) (:390-396) is row e of A transposed: contiguous, read in place
    const double* __restrict__ acol = At + (size_t)(two ? e : 0) * ld;

    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        if (lane == 0) stamp_min(dbg, 4);
        const bool mine = lane < 16 || (two && lane < 32);
        const double s = ring_consume<RC, S, NSW, 32>(rev_ring, ctl, nchunk, m, lane, mine,
                                                      (dbg && blockIdx.x == 100) ? dbg + 12 : nullptr);
        if (lane < 16 && row0 + lane < m) st_sc1(xB + row0 + lane, s);
        if (two && lane >= 16 && lane < 32 && row0 + lane - 16 < m) st_sc1(u + row0 + lane - 16, s);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) stamp_max(dbg, 5);
        if (lane == 0 && arrive_is_last(&st->arrive_xu, (int)gridDim.x)) s_last = 1;
    } else {
        const int w = wave - 1;
        auto load_chunk_a = [&](int ch, Regs& g) {   // a_e of a chunk (needs the entering variable)
            const int k = ch * RC + 2 * lane;
            const bool ok = ch < nchunk && k < ld;
            if (two) {
                g.av.x = (ok && k < m) ? acol[k] : 0.0;
                g.av.y = (ok && k + 1 < m) ? acol[k + 1] : 0.0;
            }
        };
        auto load_chunk = [&](int ch, Regs& g) {
            load_chunk_b(ch, g);
            load_chunk_a(ch, g);
        };
        auto fill = [&](int ch, const Regs& g) {
            const int slot = ch % S;
            while (lds_load(&ctl.done) < ch - S + 1) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            double* __restrict__ tile = rev_ring + (size_t)slot * 32 * ROW;
            // M[i, j] * v[j], rounded as the C# rounds it before the add (:406)
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                double2 p;
                p.x = g.r[q].x * g.bv.x; p.y = g.r[q].y * g.bv.y;
                *reinterpret_cast<double2*>(tile + (w * RPW + q) * ROW + 2 * lane) = p;
                if (two) {
                    p.x = g.r[q].x * g.av.x; p.y = g.r[q].y * g.av.y;
                    *reinterpret_cast<double2*>(tile + (16 + w * RPW + q) * ROW + 2 * lane) = p;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&ctl.ready[slot], 1, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        };
#pragma unroll
        for (int k = 0; k < D; ++k) load_chunk_a(k, rg[k]);   // (B^-1 and b were requested at the top)
        for (int ch = 0; ch < nchunk; ch += D + 1) {  // register sets rotate with the chunk number
#pragma unroll
            for (int k = 0; k <= D; ++k) {
                if (ch + k < nchunk) {
                    load_chunk(ch + k + D, rg[(k + D) % (D + 1)]);
                    fill(ch + k, rg[k]);
                }
            }
        }
    }
    __syncthreads();
    if (!s_last) return;
    // ---- exits of the loop head, ratio test, bookkeeping, eta column: the last workgroup ----
    __threadfence_block();
    if (tid == 0) stamp_max(dbg, 6);
    double* s_rat = rev_ring;                                       // kRatioLds doubles
    int* s_bvi = reinterpret_cast<int*>(rev_ring + kRatioLds);      // kRatioLds ints
    double* s_u = rev_ring + kRatioLds + kRatioLds / 2;             // kRatioLds doubles
    rev_ratio_body<true>(u, xB, basic, is_basic, cB, c, Binv, ld, browbuf, fac, log, n, m, st, s_rat,
                         s_bvi, s_u, dbg);
    __syncthreads();
    if (tid == 0) st->arrive_xu = 0;
    if (tid == 0) stamp_max(dbg, 7);
}

template <int NSW, int NQ>
static void launch_update_y(lpr_revised* s, int do_update) {
    constexpr int RC = NSW * 8 * NQ;
    constexpr size_t lds = (size_t)3 * 16 * (RC + kRingPad) * sizeof(double);
    static unsigned long long asked = 0;  // per device bit
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !(asked & (1ull << dev))) {
        asked |= 1ull << dev;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rev_update_y<NSW, NQ>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((k_rev_update_y<NSW, NQ>), dim3((s->m + 15) / 16), dim3(64 * (NSW + 1)), lds,
                       s->eng->stream, s->Binv, s->ldb, s->m, s->browbuf, s->fac, s->cB, s->y,
                       s->state, do_update, s->is_basic, s->n, s->wmin + (s->n + 31) / 32);
}

template <class K>
static void raise_dyn_lds(K kernel, size_t bytes, unsigned long long& asked) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !(asked & (1ull << dev))) {
        asked |= 1ull << dev;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    }
}

// rc = c - y A, then the entering variable and its column
void rev_launch_rc_enter(lpr_revised* s) {
    constexpr int NSW = 6, NQ = 4;
    constexpr size_t lds = (size_t)4 * 32 * (NSW * 4 * NQ + kRingPad) * sizeof(double);
    static unsigned long long asked = 0;
    raise_dyn_lds(&k_rev_rc_enter<NSW, NQ>, lds, asked);
    hipLaunchKernelGGL((k_rev_rc_enter<NSW, NQ>), dim3((s->n + 31) / 32), dim3(64 * (NSW + 1)), lds,
                       s->eng->stream, s->A, s->lda, s->m, s->n, s->y, s->c, s->rcx, s->is_basic,
                       s->Binv, s->ldb, s->u, s->state, s->dbg_stamps, s->At, s->wmin);
}

// x_B and u in one pass over B^-1, then the ratio test and the bookkeeping of the pivot
void rev_launch_xu_ratio(lpr_revised* s) {
    constexpr size_t lds = (size_t)4 * 32 * (kXuRC + kRingPad) * sizeof(double);
    static_assert(lds >= kRatioLds * (2 * sizeof(double) + sizeof(int)), "the tail borrows the ring");
    static unsigned long long asked = 0;
    raise_dyn_lds(&k_rev_xu_ratio, lds, asked);
    hipLaunchKernelGGL(k_rev_xu_ratio, dim3((s->m + 15) / 16), dim3(64 * (kXuNSW + 1)), lds,
                       s->eng->stream, s->Binv, s->ldb, s->m, s->n, s->b, s->xB, s->At, s->u,
                       s->basic, s->is_basic, s->cB, s->c, s->browbuf, s->fac, s->log, s->state,
                       s->dbg_stamps);
}

// y = c_B B^-1 of the current state (the head of a call: nothing to apply)
void rev_launch_y(lpr_revised* s) { launch_update_y<7, 4>(s, 0); }

// E * B^-1 of the pivot just chosen + the next iteration's y
void rev_launch_update_y(lpr_revised* s) { launch_update_y<7, 4>(s, 1); }

}  // namespace lpr
