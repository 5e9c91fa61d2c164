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

// Walk `len` (<= RC) products of one output, in order.  RC % 16 == 0.
template <int RC>
__device__ __forceinline__ double ring_walk(const double* __restrict__ row, int len, double s) {
    if (len == RC) {
        const double2* __restrict__ r2 = reinterpret_cast<const double2*>(row);
        double2 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = r2[u];
#pragma unroll
        for (int k0 = 0; k0 < RC; k0 += 16) {
            double2 na[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                na[u] = r2[(k0 + 16) / 2 + u];  // next group: in flight under this group's adds
                s = s + a[u].x;                 // the C#'s `s += product` (:406, :420)
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
    } else {
        for (int k = 0; k < len; ++k) s = s + row[k];
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
    const RevState* __restrict__ st, int do_update) {
    constexpr int RC = NSW * 8 * NQ;
    constexpr int ROW = RC + kRingPad;
    constexpr int S = 3;  // ring slots
    static_assert(RC % 32 == 0, "bank spreading assumes RC % 32 == 0");
    extern __shared__ __attribute__((aligned(16))) double rev_ring[];  // [S][16][ROW]
    __shared__ RingCtl ctl;
    if (st->status != kRunning) return;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    if (tid < 4) ctl.ready[tid] = 0;
    if (tid == 4) ctl.done = 0;
    __syncthreads();
    const int j0 = blockIdx.x * 16;
    const int nchunk = (m + RC - 1) / RC;
    const int r = do_update ? st->leaving_row : -1;
    const int ld2 = ld >> 1;

    if (wave == 0) {
        // ---- the walker ----
        __builtin_amdgcn_s_setprio(3);
        double s = 0.0;
        for (int c = 0; c < nchunk; ++c) {
            const int slot = c % S;
            const int need = NSW * (c / S + 1);
            while (lds_load(&ctl.ready[slot]) < need) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            if (lane < 16)
                s = ring_walk<RC>(rev_ring + ((size_t)slot * 16 + lane) * ROW, min(RC, m - c * RC), s);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __atomic_store_n(&ctl.done, c + 1, __ATOMIC_RELAXED);
        }
        if (lane < 16 && j0 + lane < m) y[j0 + lane] = s;
        return;
    }

    // ---- the stagers ----
    const int w = wave - 1;
    const int cp = lane & 7;       // column pair of the strip
    const int rs = lane >> 3;      // row of a wave-instruction
    const int c2 = (j0 >> 1) + cp;
    const bool col_ok = c2 < ld2;
    double2* __restrict__ B2 = reinterpret_cast<double2*>(Binv);
    const double2 br = (do_update && col_ok) ? reinterpret_cast<const double2*>(browbuf)[c2]
                                             : make_double2(0.0, 0.0);
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
    load_chunk(0, x, f, cb);
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
                       s->state, do_update);
}

// y = c_B B^-1 of the current state (the head of a call: nothing to apply)
void rev_launch_y(lpr_revised* s) { launch_update_y<7, 4>(s, 0); }

// E * B^-1 of the pivot just chosen + the next iteration's y
void rev_launch_update_y(lpr_revised* s) { launch_update_y<7, 4>(s, 1); }

}  // namespace lpr
