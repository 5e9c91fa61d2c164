// overlap_kernels.hip -- the primal path of tableaux above 1 MB: K pivots per sweep, decided by
// persistent "loop head" workgroups; on the largest tableaux the heads of the NEXT K pivots run
// while the sweep of the current K is running.
// (reference: LPR_381_Group_V22/Simplex/PrimalSimplexSolver.cs:102-211)
//
// block_kernels.hip explains why K pivots can be decided from O(R + C) data each and then applied
// to every element in one sweep with bit-identical results.  The pieces here:
//   ov_heads / ov_heads_rich   the K loop heads of a block as ONE persistent group of G workgroups
//                              with a spin barrier between the two phases of a head.
//   ov_tiles                   the sweep: 32-row x 256-double2 tiles.
// and three ways of putting them on the device (lpr_engine.hip picks by tableau size):
//   k_ov_heads + k_ov_sweep    heads, then the sweep in place (0x40tr; 1 MB .. 80 MB).
//   k_ov2_heads || k_ov2_sweep two kernels on two streams (0x30tr; above 80 MB): the sweep is out
//                              of place (tableau buffer `cur` -> `cur ^ 1`), so while it runs the
//                              buffer it reads is still the tableau BEFORE the block, and the heads
//                              of the next block work from that very buffer, taking every column /
//                              row they fetch through the block being swept plus their own earlier
//                              pivots.  A step lasts max(sweep, heads).
//   k_ov_step                  the same overlap in one launch (0x50tr): workgroups [0, G) are the
//                              heads -- dispatched first, so resident together -- the rest sweep.
//
// Control state that a launch both reads and updates exists twice (`OvCtl ctl[2]`): the launch
// with parity p reads ctl[p] and its two lead workgroups write ctl[p ^ 1] -- so no workgroup can
// see a field change under it.  Same for the barrier counter.
#include "engine_common.hpp"
#include "select_common.hpp"

#include <new>

#include <hip/hip_ext.h>

#pragma clang fp contract(off)

// tools/sweep_bench.hip compiles this file with LPR_OV_KERNELS_ONLY and LPR_OV_DIAG bits to take
// the sweep apart (1: no multiply-subtract chains, 2: no pivot-row loads, 4: half the chains,
// 8: plain instead of non-temporal loads / stores, 16: static grid with every XCD on a contiguous
// range of tiles, 32: tiles taken column-major, 64 / 128: write-through stores of system / agent
// scope, 256: no half tiles at the tail of the queue).  The library is built with neither.
#ifndef LPR_OV_DIAG
#define LPR_OV_DIAG 0
#endif

namespace lpr {

constexpr int kOvMax = 16;      // pivots per block, upper bound
constexpr int kOvNT = 256;      // threads per workgroup (heads and tiles)
constexpr int kOvGroups = 64;   // head workgroups, upper bound
constexpr unsigned kOvSpinMax = 1u << 22;
// B.sflag also holds kOvDoneCopies copies of the heads' completion count, one per 4 352 bytes (in
// different memory channels): the ~900 workgroups of a sweep that ask for it poll 32 different
// lines, not one (all on one word measured as a hot spot that slowed the heads' last gathers)
constexpr int kOvDoneCopies = 32;
constexpr int kOvDoneStride = 1088;  // in words
constexpr size_t kOvFlagWords = (size_t)(kOvDoneCopies + 1) * kOvDoneStride;
#ifndef LPR_OV_TILE_ROWS
#define LPR_OV_TILE_ROWS 32  // (tools/sweep_bench.hip overrides it)
#endif
constexpr int kOvTileRows = LPR_OV_TILE_ROWS;  // rows per sweep workgroup (TR in flight at a time)
// ov_heads_rich: the unit of a hand-off is the WAVE, not the workgroup -- every wave publishes its
// own partial once its own stores have drained and collects all G x kOvWPG of them itself: no LDS
// combine, no workgroup barrier in a head (false: one partial per workgroup, four barriers per head)
#ifndef LPR_OV_WAVE_HANDOFF
#define LPR_OV_WAVE_HANDOFF 1
#endif
constexpr bool kOvWaveHandoff = LPR_OV_WAVE_HANDOFF != 0;
constexpr int kOvWPG = kOvNT / 64;                  // waves per head workgroup
constexpr int kOvParts = kOvGroups * kOvWPG;        // partial slots per bank

struct OvCtl {
    int32_t status;    // kRunning or the final lpr_status
    int32_t pending;   // kRunning, or the status that ends the solve once the staged block is swept
    int32_t kdone;     // pivots of the block the reader's tiles must apply
    int32_t slot;      // staging slot holding that block (the heads stage into slot ^ 1)
    int32_t cur;       // buffer holding the tableau before that block
    int32_t sweep;     // parity of the sweep direction
    int32_t r[kOvMax]; // pivot rows of that block
    int32_t error;     // a grid barrier timed out
    int32_t head_xcc;  // XCD the loop heads of the previous launch shared (-1: none / spread)
    int64_t staged;    // pivots decided so far (index of the next one)
    int64_t applied;   // pivots swept into the tableau so far
    int64_t max_iter;  // <= 0: no limit
    int64_t log_cap;
};

struct OvBuffers {      // everything the step kernel touches, passed by value
    double* Tb[2];      // the two tableau buffers
    double* prow;       // [2][kOvMax][ld]  normalised pivot rows per staging slot
    double* fcol;       // [2][kOvMax][Rp]  factor columns per staging slot
    double* zrow;       // [ld]  Z row of the tableau after all staged pivots
    double* bvec;       // [2][Rp] RHS column after `staged` pivots in bvec[staged & 1]
    ZPart* zparts;      // [2][kOvGroups]
    double* rparts;     // [3][kOvGroups] ratio-test partials: ratio, pivot element, row (as double); + f0
    unsigned long long* gran;  // [3][kOvGroups][3] partials as {epoch, 32-bit value} granules:
                               // Z-row partials bank 0 / 1, ratio partials (ov_heads_rich)
    unsigned long long* xgran;  // [kOvGroups] {launch epoch, XCC id} of every head workgroup
    unsigned long long* hx;     // {launch epoch, 0x100 | XCC id}: the XCD this launch's heads share
    unsigned* tileq;            // [2] next tile of the sweep (work queue), by launch parity
    unsigned* sflag;            // [0] sweeps completed in this solve call (stored by the next sweep's
                                // start); [1] head launches completed (stored by the workgroup of
                                // a head launch that finishes last); [2] its arrival counter;
                                // [3] a sweep gave up waiting for [1]
    unsigned long long* dbg;    // diagnostic time stamps of the lead head workgroup (or null)
    OvCtl* ctl;         // [2]
    unsigned* bar;      // [2]
    int32_t* basis;
    int32_t* log;
};

constexpr int kOvStampsPerPivot = 12;   // diagnostic build of the heads (opts.variant bit 16)
constexpr int kOvStampPivots = 64;      // pivots kept (a ring over the launches)
// + xcc, mode, 6 spare; then per launch parity 8 words: heads entry / tableau ready / last head done /
// completion published, sweep (first workgroup) entry / heads' word seen
constexpr size_t kOvDbgLaunch = (size_t)kOvStampPivots * kOvStampsPerPivot + 8;
constexpr size_t kOvDbgWords = kOvDbgLaunch + 64;  // a ring over 8 steps

// ------------------------------------------------------------------------------------------
// Per solve call: Z row, RHS column and entering column of the tableau in memory.
__global__ __launch_bounds__(1024) void k_ov_prologue(const double* __restrict__ T, int ld, int R,
                                                      int C, double* __restrict__ zrow,
                                                      double* __restrict__ bvec0,
                                                      ZPart* __restrict__ bank, int G,
                                                      unsigned long long* gbank,
                                                      unsigned epoch,
                                                      unsigned long long* gran_all,
                                                      unsigned long long* __restrict__ xgran,
                                                      unsigned* __restrict__ bar,
                                                      unsigned* __restrict__ tileq,
                                                      unsigned long long* __restrict__ hx,
                                                      unsigned* __restrict__ sflag) {
    __shared__ double lds_v[16];
    __shared__ int lds_i[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    // the granule banks start from epoch 0 (the bank written below: its other workgroup slots)
    for (int k = tid; k < 9 * kOvParts; k += nt) gran_all[k] = 0ull;
    for (int k = tid; k < kOvGroups; k += nt) xgran[k] = 0ull;
    if (tid < 4) bar[tid] = 0u;
    if (tid < 2) tileq[tid] = 0u;
    if (tid == 0) *hx = 0ull;
    if (tid < 4) sflag[tid] = 0u;
    if (tid < kOvDoneCopies) sflag[(size_t)(tid + 1) * kOvDoneStride] = 0u;
    __syncthreads();
    Cand c;
    c.v = 0.0;
    c.i = -1;
    for (int j = tid; j < ld; j += nt) {
        const double v = T[j];
        zrow[j] = v;
        if (j < C - 1 && v < c.v) {
            c.v = v;
            c.i = j;
        }
    }
    c = block_cand_min(c, lds_v, lds_i);
    if (tid < G) {
        const double v = (tid == 0) ? c.v : 0.0;
        const int i = (tid == 0) ? c.i : -1;
        bank[tid].v = v;
        bank[tid].i = i;
    }
    // the same partials as granules (ov_heads_rich), see gr_publish: one per workgroup, or one per
    // wave of every workgroup
    const int nparts = kOvWaveHandoff ? G * kOvWPG : G;
    if (tid < nparts) {
        const double v = (tid == 0) ? c.v : 0.0;
        const int i = (tid == 0) ? c.i : -1;
        const unsigned long long e = (unsigned long long)epoch << 32;
        gbank[tid * 3 + 0] = e | (unsigned)__double2loint(v);
        gbank[tid * 3 + 1] = e | (unsigned)__double2hiint(v);
        gbank[tid * 3 + 2] = e | (unsigned)i;
    }
    for (int i = tid; i < R; i += nt) bvec0[i] = T[(size_t)i * ld + (C - 1)];
}

// Data that the head workgroups hand to each other INSIDE a launch (the gathered column, the
// normalised row, the RHS column, the partial arg-mins) is written and read with agent-scope
// relaxed atomics: on gfx950 these are sc1 accesses that are coherent at the memory side across
// the XCDs' L2s, so the barrier between the phases needs no L2 write-back / invalidate (which
// costs microseconds per barrier, and under a concurrently running sweep far more).  What a
// lane wrote itself, and everything written by an earlier launch, is read with plain loads.
__device__ __forceinline__ double xld(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xst(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int xld(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xst(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The same hand-off when every head workgroup of the launch sits on ONE XCD (checked per launch,
// see ov_heads_rich): the XCD's L2 is the coherence point, so the producer stores with workgroup
// scope (sc0: the line stays in that L2) and the consumer still loads with sc1 (past its own L1,
// served by the L2) -- a hand-off costs an L2 round trip instead of a trip to the memory side.
__device__ __forceinline__ void hst(double* p, double v, bool l2) {
    if (l2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void hst(unsigned long long* p, unsigned long long v, bool l2) {
    if (l2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// id (0..7) of the XCD this wave runs on
__device__ __forceinline__ unsigned ov_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xfu;
}
__device__ __forceinline__ unsigned long long ov_now() { return __builtin_amdgcn_s_memrealtime(); }

// value of `v` in lane `k` (k a compile-time constant) as a wave-uniform scalar
__device__ __forceinline__ double ov_rl(double v, int k) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}

// (value, index) lexicographic minimum over the 64 lanes of a wave, in every lane, with DPP moves
// (row-local permutes, then the two row broadcasts of gfx9) instead of six rounds of ds_bpermute:
// the operation is associative and commutative, so the order of combination does not matter.
// bit t of mask ? a : b on the bit patterns, a wave-uniform (it comes out of v_readlane): one
// v_bfe_i32 + two v_bfi_b32 with a read straight from its SGPRs, instead of an and, a compare, two
// moves of a into VGPRs and two v_cndmask (the chains are issue-bound: 12 -> 7 instructions per
// step).  Inline asm: written as C the bit operations are folded back into select(compare).
__device__ __forceinline__ double ov_bitsel(unsigned mask, int t, double a, double b) {
    int m, lo, hi;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(mask), "s"(t));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(m), "s"(__double2loint(a)), "v"(__double2loint(b)));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(m), "s"(__double2hiint(a)), "v"(__double2hiint(b)));
    return __hiloint2double(hi, lo);
}

// cand_min without branches (the early returns of cand_min compile to exec-masked branches around
// every one of the six DPP stages: ~25 instructions each, four reductions per head)
__device__ __forceinline__ Cand ov_cand_min_sel(Cand a, Cand b) {
    const bool take = (b.i >= 0) & ((a.i < 0) | (b.v < a.v) | ((b.v == a.v) & (b.i < a.i)));
    Cand r;
    r.v = take ? b.v : a.v;
    r.i = take ? b.i : a.i;
    return r;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Cand ov_dpp_min(Cand c) {
    Cand o;
    const int lo = __double2loint(c.v), hi = __double2hiint(c.v);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    o.i = __builtin_amdgcn_update_dpp(c.i, c.i, CTRL, ROW_MASK, 0xf, false);
    o.v = __hiloint2double(ohi, olo);
    return ov_cand_min_sel(c, o);
}
// Wave-wide lexicographic minimum of (value, index), index < 0 = no candidate; the result in every
// lane.  Two plain reductions instead of one on pairs (six stages of ~20 instructions each): the
// minimum VALUE over the candidates (two DPP moves + v_min_f64 per stage), then the minimum INDEX
// over the lanes that hold it (v_min_i32 with a DPP operand).  Candidates are never NaN (a ratio is
// finite and >= 0, a Z-row entry passed a `<` test), -0.0 == +0.0 ties go to the lower index as in
// cand_min, and the value returned when there is no candidate is not used by any caller.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double ov_dpp_fmin(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return fmin(v, __hiloint2double(ohi, olo));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int ov_dpp_imin(int i) {
    return min(i, __builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ Cand ov_wave_min(Cand c) {
    double v = (c.i >= 0) ? c.v : INFINITY;
    v = ov_dpp_fmin<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = ov_dpp_fmin<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = ov_dpp_fmin<0x141, 0xf>(v);  // row_half_mirror
    v = ov_dpp_fmin<0x140, 0xf>(v);  // row_mirror: every row of 16 holds its minimum
    v = ov_dpp_fmin<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
    v = ov_dpp_fmin<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3: lane 63 holds the minimum
    const double vmin = ov_rl(v, 63);
    int i = (c.i >= 0 && c.v == vmin) ? c.i : INT_MAX;
    i = ov_dpp_imin<0xB1, 0xf>(i);
    i = ov_dpp_imin<0x4E, 0xf>(i);
    i = ov_dpp_imin<0x141, 0xf>(i);
    i = ov_dpp_imin<0x140, 0xf>(i);
    i = ov_dpp_imin<0x142, 0xa>(i);
    i = ov_dpp_imin<0x143, 0xc>(i);
    const int imin = __builtin_amdgcn_readlane(i, 63);
    Cand r;
    r.v = vmin;
    r.i = (imin == INT_MAX) ? -1 : imin;
    return r;
}

// arg-min over the G partials, one per lane (G <= 64), every wave on its own
__device__ __forceinline__ Cand ov_reduce_zparts(const ZPart* bank, int G) {
    const int lane = threadIdx.x & (kWave - 1);
    Cand c;
    c.v = 0.0;
    c.i = -1;
    if (lane < G) {
        c.v = xld(&bank[lane].v);
        c.i = xld(&bank[lane].i);
    }
    return wave_cand_min(c);
}

// Barrier over the G head workgroups.  Returns false when it timed out.  No cache maintenance:
// the workgroup barrier waits for every lane's (sc1) stores to complete, the arrival and the poll
// are memory-side atomics, and what is read afterwards is read with sc1 loads (see xld).
__device__ __forceinline__ bool ov_barrier(unsigned* bar, unsigned target) {
    __shared__ int ok;
    __builtin_amdgcn_s_waitcnt(0);  // this lane's stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int good = 1;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kOvSpinMax) {
                good = 0;
                break;
            }
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// ------------------------------------------------------------------------------------------
// The heads of the next block (workgroups [0, G)).
template <int NT>
__device__ void ov_heads(const OvBuffers B, int ld, int R, int C, int Rp, int K, int G, int lp,
                         bool solo) {
    __shared__ double lds_v[16];
    __shared__ int lds_i[16];
    __shared__ double lds_p[2];
    __shared__ int s_r[kOvMax];   // rows of the pivots staged by this launch
    __shared__ int rA[kOvMax];    // rows of the pivots of the block being swept
    __shared__ double s_fa[kOvMax], s_fn[kOvMax];  // f_t[r] of the earlier pivots
    __shared__ double s_pa[kOvMax], s_pn[kOvMax];  // p_t[e], later p_t[rhs]
    const int tid = threadIdx.x, nt = blockDim.x;
    const int g = blockIdx.x;
    const bool lead = (g == 0);
    const OvCtl* ci = B.ctl + lp;
    OvCtl* co = B.ctl + (lp ^ 1);
    const int32_t status = ci->status;
    const int32_t pend_in = ci->pending;
    const int kb = solo ? 0 : ci->kdone;  // pivots of the block being swept right now
    const int sa = ci->slot;        // ... staged in this slot
    const int64_t staged0 = ci->staged;
    const int64_t mx = ci->max_iter;
    const int64_t log_cap = ci->log_cap;
    const double* __restrict__ Tin = B.Tb[ci->cur];
    const int ld2 = ld >> 1;
    const int rhs = C - 1;
    const size_t slotP = (size_t)kOvMax * ld, slotF = (size_t)kOvMax * Rp;
    // not __restrict__: other head workgroups write these between the barriers
    double* prowA = B.prow + (size_t)sa * slotP;
    double* fcolA = B.fcol + (size_t)sa * slotF;
    double* prowN = B.prow + (size_t)(sa ^ 1) * slotP;
    double* fcolN = B.fcol + (size_t)(sa ^ 1) * slotF;
    double* zrow = B.zrow;
    unsigned* bar = B.bar + lp;
    if (tid < kOvMax) rA[tid] = (tid < kb) ? ci->r[tid] : -1;
    __syncthreads();

    int32_t pend_out = pend_in;
    int32_t status_out = status;
    int count = 0;
    int err = 0;
    unsigned nbar = 0;

    if (status == kRunning && pend_in != kRunning) {
        status_out = pend_in;  // the block staged before is being swept by this very launch
    } else if (status == kRunning) {
        for (int q = 1; q <= K; ++q) {
            const int64_t pidx = staged0 + q - 1;
            const ZPart* bank_in = B.zparts + (pidx & 1) * kOvGroups;
            ZPart* bank_out = B.zparts + ((pidx + 1) & 1) * kOvGroups;
            const double* bprev = B.bvec + (size_t)(pidx & 1) * Rp;
            double* bnew = B.bvec + (size_t)((pidx + 1) & 1) * Rp;
            double* colq = fcolN + (size_t)(q - 1) * Rp;

            // ---- entering column (:152-167) from the partials of the previous head ----
            const int e = ov_reduce_zparts(bank_in, G).i;
            if (e < 0) {
                pend_out = LPR_OK_OPTIMAL;
                break;
            }
            // ---- column e of the tableau after all earlier pivots, this workgroup's rows ----
            if (tid < kb) s_pa[tid] = prowA[(size_t)tid * ld + e];
            if (tid >= 32 && tid - 32 < q - 1) s_pn[tid - 32] = xld(&prowN[(size_t)(tid - 32) * ld + e]);
            __syncthreads();
            Cand rc;
            rc.v = DBL_MAX;
            rc.i = -1;
            double a_of_best = 0.0;
            for (int i = g * nt + tid; i < R; i += G * nt) {
                double c = Tin[(size_t)i * ld + e];
                for (int t0 = 0; t0 < kb; t0 += kOvMax) {  // through the block being swept
                    double f[kOvMax];
#pragma unroll
                    for (int u = 0; u < kOvMax; ++u)
                        f[u] = (t0 + u < kb) ? fcolA[(size_t)(t0 + u) * Rp + i] : 0.0;
#pragma unroll
                    for (int u = 0; u < kOvMax; ++u) {
                        const int t = t0 + u;
                        if (t < kb) {
                            if (i == rA[t]) {
                                c = s_pa[t];
                            } else {
                                const double prod = f[u] * s_pa[t];
                                c = c - prod;
                            }
                        }
                    }
                }
                for (int t0 = 0; t0 < q - 1; t0 += 8) {  // through this block's earlier pivots
                    double f[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        f[u] = (t0 + u < q - 1) ? fcolN[(size_t)(t0 + u) * Rp + i] : 0.0;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int t = t0 + u;
                        if (t < q - 1) {
                            if (i == s_r[t]) {
                                c = s_pn[t];
                            } else {
                                const double prod = f[u] * s_pn[t];
                                c = c - prod;
                            }
                        }
                    }
                }
                xst(&colq[i], c);
                // FindLeavingVariable (:169-191) on this lane's rows: its RHS entries are its own
                if (i == 0) xst(&B.rparts[3 * kOvGroups], c);  // T[0, e], the Z row's factor
                if (i >= 1 && c > 1e-9) {
                    const double ratio = ieee_div(bprev[i], c);
                    if (ratio >= 0 && ratio < rc.v) {
                        rc.v = ratio;
                        rc.i = i;
                        a_of_best = c;
                    }
                }
            }
            {   // this workgroup's (ratio, row) minimum and its pivot element -> the partials
                const int my_best = rc.i;
                rc = block_cand_min(rc, lds_v, lds_i);
                if (rc.i >= 0 && my_best == rc.i) lds_p[0] = a_of_best;  // one lane owns that row
                __syncthreads();
                if (tid == 0) {
                    xst(&B.rparts[g], rc.v);
                    xst(&B.rparts[kOvGroups + g], rc.i >= 0 ? lds_p[0] : 0.0);
                    xst(&B.rparts[2 * kOvGroups + g], (double)rc.i);
                }
            }
            if (!ov_barrier(bar, (++nbar) * (unsigned)G)) {
                err = 1;
                break;
            }

            // ---- the leaving row: lexicographic minimum of the G partials (every wave on its own)
            double p, f0;
            int r;
            {
                const int lane = tid & (kWave - 1);
                Cand c;
                c.v = DBL_MAX;
                c.i = -1;
                double a = 0.0;
                if (lane < G) {
                    c.v = xld(&B.rparts[lane]);
                    a = xld(&B.rparts[kOvGroups + lane]);
                    c.i = (int)xld(&B.rparts[2 * kOvGroups + lane]);
                }
                f0 = xld(&B.rparts[3 * kOvGroups]);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    Cand o;
                    o.v = __shfl_xor(c.v, off, kWave);
                    o.i = __shfl_xor(c.i, off, kWave);
                    const double oa = __shfl_xor(a, off, kWave);
                    const Cand m = cand_min(c, o);
                    if (m.i != c.i || m.v != c.v) a = oa;  // the other side won
                    c = m;
                }
                r = c.i;
                p = a;
            }
            if (r < 0) {
                pend_out = LPR_UNBOUNDED;
                break;
            }
            if (mx > 0 && pidx >= mx) {
                pend_out = LPR_PIVOT_LIMIT;
                break;
            }
            if (tid == 0) s_r[q - 1] = r;

            // ---- row r after all earlier pivots, normalised (:199); next Z row; partial ----
            // f_t[r] of every earlier pivot, once per workgroup
            if (tid < kb) s_fa[tid] = fcolA[(size_t)tid * Rp + r];
            if (tid >= 32 && tid - 32 < q - 1) s_fn[tid - 32] = xld(&fcolN[(size_t)(tid - 32) * Rp + r]);
            if (tid >= 64 && tid - 64 < kb) s_pa[tid - 64] = prowA[(size_t)(tid - 64) * ld + rhs];
            if (tid >= 96 && tid - 96 < q - 1) s_pn[tid - 96] = xld(&prowN[(size_t)(tid - 96) * ld + rhs]);
            double wr = (tid == 128) ? Tin[(size_t)r * ld + rhs] : 0.0;
            __syncthreads();
            const double2* Tin2 = reinterpret_cast<const double2*>(Tin);
            const double2* prowA2 = reinterpret_cast<const double2*>(prowA);
            double2* prowN2 = reinterpret_cast<double2*>(prowN);
            double2* zrow2 = reinterpret_cast<double2*>(zrow);
            Cand n;
            n.v = 0.0;
            n.i = -1;
            for (int c2 = g * nt + tid; c2 < ld2; c2 += G * nt) {
                double2 w = Tin2[(size_t)r * ld2 + c2];
                double2 z = zrow2[c2];
                for (int t0 = 0; t0 < kb; t0 += kOvMax) {
                    double2 ps[kOvMax];
#pragma unroll
                    for (int u = 0; u < kOvMax; ++u)
                        ps[u] = (t0 + u < kb) ? prowA2[(size_t)(t0 + u) * ld2 + c2]
                                              : make_double2(0.0, 0.0);
#pragma unroll
                    for (int u = 0; u < kOvMax; ++u) {
                        const int t = t0 + u;
                        if (t < kb) {
                            if (r == rA[t]) {
                                w = ps[u];
                            } else {
                                const double f = s_fa[t];
                                const double px = f * ps[u].x;
                                const double py = f * ps[u].y;
                                w.x = w.x - px;
                                w.y = w.y - py;
                            }
                        }
                    }
                }
                for (int t0 = 0; t0 < q - 1; t0 += 8) {
                    double2 ps[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        ps[u] = (t0 + u < q - 1) ? prowN2[(size_t)(t0 + u) * ld2 + c2]
                                                 : make_double2(0.0, 0.0);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int t = t0 + u;
                        if (t < q - 1) {
                            if (r == s_r[t]) {
                                w = ps[u];
                            } else {
                                const double f = s_fn[t];
                                const double px = f * ps[u].x;
                                const double py = f * ps[u].y;
                                w.x = w.x - px;
                                w.y = w.y - py;
                            }
                        }
                    }
                }
                const int j = 2 * c2;
                double2 pq;
                pq.x = (j < C) ? ieee_div(w.x, p) : 0.0;  // :199 true division
                pq.y = (j + 1 < C) ? ieee_div(w.y, p) : 0.0;
                xst(&prowN[(size_t)(q - 1) * ld + 2 * c2], pq.x);
                xst(&prowN[(size_t)(q - 1) * ld + 2 * c2 + 1], pq.y);
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

            // ---- RHS column after this pivot ----
            if (tid == 128) {
                for (int t = 0; t < kb; ++t) {
                    if (r == rA[t]) {
                        wr = s_pa[t];
                    } else {
                        const double prod = s_fa[t] * s_pa[t];
                        wr = wr - prod;
                    }
                }
                for (int t = 0; t < q - 1; ++t) {
                    if (r == s_r[t]) {
                        wr = s_pn[t];
                    } else {
                        const double prod = s_fn[t] * s_pn[t];
                        wr = wr - prod;
                    }
                }
                lds_p[0] = ieee_div(wr, p);  // p was read by every lane before the arg-min barriers above
            }
            __syncthreads();
            const double prhs = lds_p[0];
            for (int i = g * nt + tid; i < R; i += G * nt) {
                const double prod = colq[i] * prhs;  // own rows: written by this lane above
                xst(&bnew[i], (i == r) ? prhs : bprev[i] - prod);
            }
            if (tid == 0) {
                xst(&bank_out[g].v, n.v);
                xst(&bank_out[g].i, n.i);
                if (lead) {
                    co->r[q - 1] = r;
                    B.basis[r - 1] = e;  // :142
                    if (pidx < log_cap) {
                        B.log[2 * pidx] = r;
                        B.log[2 * pidx + 1] = e;
                    }
                }
            }
            count = q;
            if (!ov_barrier(bar, (++nbar) * (unsigned)G)) {
                err = 1;
                break;
            }
        }
    }

    if (lead && tid == 0) {  // the next launch's view (fields owned by the heads)
        const bool staged_now = (status == kRunning && pend_in == kRunning);
        co->status = err ? LPR_DEVICE_ERROR : status_out;
        co->pending = pend_out;
        co->kdone = staged_now ? count : 0;
        co->slot = staged_now ? (sa ^ 1) : sa;
        co->staged = staged0 + (staged_now ? count : 0);
        co->max_iter = mx;
        co->log_cap = log_cap;
        co->error = err | ci->error;
        co->head_xcc = -1;
        B.bar[lp ^ 1] = 0u;  // nobody touches the other counter during this launch
        if (solo) {  // no sweep in this launch: its fields are carried over here
            co->applied = ci->applied;
            co->cur = ci->cur;
            co->sweep = ci->sweep;
        }
    }
}

// ---- partials as granules (the data is the flag) ------------------------------------------------
// A partial (double v, int i) of workgroup g is three 8-byte granules {epoch << 32 | 32 bits}, each
// written by one sc1 store; a reader re-reads them until all three carry the epoch it waits for.
// That is the barrier and the data transfer in one memory round trip: a workgroup publishes its
// partial after its other stores have drained (s_waitcnt + workgroup barrier), so a reader that
// holds all G partials of an epoch also knows that every workgroup's stores of that phase landed.
__device__ __forceinline__ void gr_publish(unsigned long long* g3, unsigned epoch, double v, int i,
                                           bool l2 = false) {
    const unsigned long long e = (unsigned long long)epoch << 32;
    hst(g3 + 0, e | (unsigned)__double2loint(v), l2);
    hst(g3 + 1, e | (unsigned)__double2hiint(v), l2);
    hst(g3 + 2, e | (unsigned)i, l2);
}

// One wave: lane l < G fetches workgroup l's partial of `epoch`; returns the lexicographic minimum
// (value, then index; index < 0 = no candidate) in every lane, or sets *fail after kOvSpinMax polls.
__device__ __forceinline__ Cand gr_collect(const unsigned long long* base, int G, unsigned epoch,
                                           double none, int* fail) {
    const int lane = threadIdx.x & (kWave - 1);
    Cand c;
    c.v = none;
    c.i = -1;
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
        if (lane < G) {
            const unsigned long long a = __hip_atomic_load(base + lane * 3 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long b = __hip_atomic_load(base + lane * 3 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long d = __hip_atomic_load(base + lane * 3 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (unsigned)(a >> 32) == epoch && (unsigned)(b >> 32) == epoch &&
                 (unsigned)(d >> 32) == epoch;
            c.v = __hiloint2double((int)(unsigned)b, (int)(unsigned)a);
            c.i = (int)(unsigned)d;
        }
        if (__all(ok)) break;
        if (spins > kOvSpinMax) {
            *fail = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    if (lane >= G) {
        c.v = none;
        c.i = -1;
    }
    return ov_wave_min(c);
}

// The same with one partial per WAVE (P = G x kOvWPG of them, P <= 256): every wave collects for
// itself, a lane fetches the partials lane, lane + 64, ...
__device__ __forceinline__ Cand gr_collect_waves(const unsigned long long* base, int P,
                                                 unsigned epoch, double none, int* fail) {
    const int lane = threadIdx.x & (kWave - 1);
    Cand c;
    c.v = none;
    c.i = -1;
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
        Cand acc;
        acc.v = none;
        acc.i = -1;
        for (int p = lane; p < P; p += kWave) {
            const unsigned long long a = __hip_atomic_load(base + p * 3 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long b = __hip_atomic_load(base + p * 3 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long d = __hip_atomic_load(base + p * 3 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (unsigned)(a >> 32) == epoch && (unsigned)(b >> 32) == epoch &&
                 (unsigned)(d >> 32) == epoch;
            Cand o;
            o.v = __hiloint2double((int)(unsigned)b, (int)(unsigned)a);
            o.i = (int)(unsigned)d;
            acc = ov_cand_min_sel(acc, o);
        }
        if (__all(ok)) {
            c = acc;
            break;
        }
        if (spins > kOvSpinMax) {
            *fail = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return ov_wave_min(c);
}

// ... and the wave's side of it: reduce, wait for the wave's OWN stores of the phase, publish.  A
// collector that has seen all P partials of an epoch knows that every wave's stores have landed.
__device__ __forceinline__ void ov_publish_wave_min(Cand c, unsigned long long* g3, unsigned epoch,
                                                    bool l2) {
    c = ov_wave_min(c);
    __builtin_amdgcn_s_waitcnt(0);
    if ((threadIdx.x & (kWave - 1)) == 0) gr_publish(g3, epoch, c.v, c.i, l2);
}

// One phase of a head ends: every wave reduces its lanes' candidates and leaves the result in its
// LDS slot, waits until its own stores of the phase have been acknowledged, and meets the others at
// ONE workgroup barrier; then lane 0 combines the slots and publishes the workgroup's partial.
__device__ __forceinline__ void ov_publish_min(Cand c, double* lds_v, int* lds_i,
                                               unsigned long long* g3, unsigned epoch, bool l2) {
    c = ov_wave_min(c);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) {
        lds_v[wave] = c.v;
        lds_i[wave] = c.i;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        Cand r;
        r.v = lds_v[0];
        r.i = lds_i[0];
        for (int w = 1; w < (int)blockDim.x / kWave; ++w) {
            Cand o;
            o.v = lds_v[w];
            o.i = lds_i[w];
            r = cand_min(r, o);
        }
        gr_publish(g3, epoch, r.v, r.i, l2);
    }
}

// ------------------------------------------------------------------------------------------
// The heads again, for the launches where they have a kernel (and so a register file) of their
// own (k_ov_heads, k_ov2_heads).  Same protocol and arithmetic as ov_heads; what differs is where
// the operands live.  A lane owns one row (gather / RHS) and one column pair (pivot row / Z row),
// and everything that belongs to them stays in registers for the whole launch:
//   fA / pA   its slices of the block being swept (the same for every pivot of the launch)
//   myf / myp its slices of this block's own pivots so far (it computed them itself)
//   myz, myb  its Z-row pair and RHS entry after the pivots staged so far
// so a head needs three dependent memory trips -- the partial arg-mins; T[i, e] with the p_t[e];
// after the barrier the ratio partials, then T[r, c] with the f_t[r] -- instead of about ten.
// Lanes of very tall / wide tableaux own further rows / column pairs; those go through memory
// (the loops marked "further").
//
// Placement: the launch has `spread` x G workgroups and only every spread-th one works (g =
// blockIdx.x / spread).  Workgroups are dealt round-robin over the 8 XCDs, so with spread = 8 the
// G working ones normally share an XCD -- that is not a guarantee, so every launch checks it: each
// workgroup publishes its XCC id through the memory side, all collect the G ids, and only when
// they are equal do the hand-offs of this launch go through that XCD's L2 (hst / gr_publish with
// l2 = true); otherwise they take the memory-side form.  Same bits either way.
// STAMP: diagnostic build, the lead workgroup leaves s_memrealtime stamps per phase in B.dbg.
// wait_sweeps >= 0 (two-stream form): this launch was queued right behind the previous heads,
// WITHOUT waiting for the sweep that writes the tableau buffer it reads; it does its start-up
// (placement check, register fills from the staging slots) and then waits until B.sflag -- stored
// by the first workgroup of every sweep as it STARTS: sweep k's start means sweeps 0..k-1 of the
// stream are complete and flushed -- says that `wait_sweeps` sweeps of this solve call have
// completed.  (The sweep it waits for the start of is the one of its own step, which only waits for
// the heads of the step before: no cycle.)  The cross-stream event that used to order the two (5-10 us of latency per
// step, on the critical path whenever the heads end last) is gone.  In this form the heads also own
// the control-block fields the sweep used to write (applied / cur / sweep parity): the sweep of the
// same step only reads the control block.
template <int NT, bool STAMP>
__device__ void ov_heads_rich(const OvBuffers B, int ld, int R, int C, int Rp, int K, int G, int lp,
                              bool solo, int spread, int no_l2, int wait_sweeps,
                              int heads_done = -1) {
    if ((int)blockIdx.x % spread != 0) return;
    __shared__ double lds_v[16];
    __shared__ int lds_i[16];
    __shared__ int s_pick[2];  // [0] the index a collect returned, [1] it timed out
    const int tid = threadIdx.x, nt = blockDim.x;
    const int g = (int)blockIdx.x / spread;
    const bool lead = (g == 0);
    const OvCtl* ci = B.ctl + lp;
    OvCtl* co = B.ctl + (lp ^ 1);
    const int32_t status = ci->status;
    const int32_t pend_in = ci->pending;
    const int kb = solo ? 0 : ci->kdone;
    const int sa = ci->slot;
    const int64_t staged0 = ci->staged;
    const bool staging = (status == kRunning && pend_in == kRunning);
    const unsigned my_xcc = ov_xcc_id();
    const unsigned xepoch = (unsigned)(staged0 + 1);  // unique per staging launch of a solve call
    if (staging && tid == 0)  // collected below, after the register fills have been requested
        __hip_atomic_store(B.xgran + g, ((unsigned long long)xepoch << 32) | my_xcc,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long* stamp = nullptr;
    if (STAMP && lead && tid == 0) stamp = B.dbg;
    if (STAMP && stamp) B.dbg[kOvDbgLaunch + 8 * ((staged0 / 16) & 7) + 0] = ov_now();
#define OV_STAMP(q_, k_)                                                                       \
    do {                                                                                       \
        if (STAMP && stamp)                                                                    \
            stamp[(size_t)(((staged0 + (q_)-1) % kOvStampPivots) * kOvStampsPerPivot) + (k_)] = \
                ov_now();                                                                      \
    } while (0)
    const int64_t mx = ci->max_iter;
    const int64_t log_cap = ci->log_cap;
    const double* __restrict__ Tin = B.Tb[ci->cur];
    const int ld2 = ld >> 1;
    const int rhs = C - 1;
    const size_t slotP = (size_t)kOvMax * ld, slotF = (size_t)kOvMax * Rp;
    double* prowA = B.prow + (size_t)sa * slotP;
    double* fcolA = B.fcol + (size_t)sa * slotF;
    double* prowN = B.prow + (size_t)(sa ^ 1) * slotP;
    double* fcolN = B.fcol + (size_t)(sa ^ 1) * slotF;
    const double2* Tin2 = reinterpret_cast<const double2*>(Tin);
    const double2* prowA2 = reinterpret_cast<const double2*>(prowA);
    double2* zrow2 = reinterpret_cast<double2*>(B.zrow);

    const int i_first = g * nt + tid, c2_first = g * nt + tid;
    const bool have_i = i_first < R, have_c = c2_first < ld2;
    double fA[kOvMax], myf[kOvMax];
    double2 pA[kOvMax], myp[kOvMax];
#pragma unroll
    for (int t = 0; t < kOvMax; ++t) {
        fA[t] = (t < kb && have_i) ? fcolA[(size_t)t * Rp + i_first] : 0.0;
        pA[t] = (t < kb && have_c) ? prowA2[(size_t)t * ld2 + c2_first] : make_double2(0.0, 0.0);
        myf[t] = 0.0;
        myp[t] = make_double2(0.0, 0.0);
    }
    double2 myz = have_c ? zrow2[c2_first] : make_double2(0.0, 0.0);
    double myb = have_i ? B.bvec[(size_t)(staged0 & 1) * Rp + i_first] : 0.0;
    // Workgroup-uniform operands are kept ONE PER LANE (the same in every wave) and read with
    // v_readlane where a chain needs them: lane t < 16 holds what belongs to pivot t of the block
    // being swept, lane 16 + t what belongs to pivot t of this block.  One load per wave fetches a
    // whole operand vector; the chains run on registers and scalars only (no LDS, no barrier).
    const int lane = tid & (kWave - 1);
    const int rAv = (lane < kb) ? ci->r[lane] : -1;                          // pivot rows, block A
    const double parhsAv = (lane < kb) ? prowA[(size_t)lane * ld + rhs] : 0.0;  // p_t[rhs], block A
    int rNv = -1;            // lane t: pivot row of this block's pivot t
    double prhsNv = 0.0;     // lane t: p_t[rhs] of this block's pivot t
    unsigned maskA = 0u, maskN = 0u;  // bit t: this lane's row is the pivot row of pivot t
#pragma unroll
    for (int t = 0; t < kOvMax; ++t) {
        const int rt = __builtin_amdgcn_readlane(rAv, t);
        maskA |= (have_i && rt == i_first) ? (1u << t) : 0u;
    }

    int32_t pend_out = pend_in;
    int32_t status_out = status;
    int count = 0;
    int err = 0;
    bool l2 = false;  // hand-offs through the XCD's L2 (all G workgroups share one XCD)

    if (status == kRunning && pend_in != kRunning) {
        status_out = pend_in;  // the block staged before is being swept by this very launch
    } else if (status == kRunning) {
        {   // where did the G workgroups land?  (memory-side exchange, once per launch)
            if (tid < kWave) {
                unsigned xo = my_xcc;
                bool same = true;
                int fail = 0;
                for (unsigned spins = 0;; ++spins) {
                    bool ok = true;
                    if (tid < G) {
                        const unsigned long long v = __hip_atomic_load(
                            B.xgran + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = (unsigned)(v >> 32) == xepoch;
                        xo = (unsigned)v;
                    }
                    if (__all(ok)) break;
                    if (spins > kOvSpinMax) {
                        fail = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                same = __all(tid >= G || xo == my_xcc);
                if (tid == 0) {
                    s_pick[0] = (same && !no_l2 && !fail) ? 1 : 0;
                    s_pick[1] = fail;
                }
            }
            __syncthreads();
            l2 = s_pick[0] != 0;
            if (s_pick[1]) err = 1;
            __syncthreads();
            // tell the sweep running beside this launch which XCD to leave to the heads
            if (lead && tid == 0 && l2 && !solo)
                __hip_atomic_store(B.hx, ((unsigned long long)xepoch << 32) | 0x100u | my_xcc,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (wait_sweeps > 0) {  // the tableau buffer this launch reads: has its sweep ended?
                if (tid == 0) {
                    int fail = 1;
                    for (unsigned spins = 0; spins < kOvSpinMax; ++spins) {
                        if (__hip_atomic_load(B.sflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >=
                            (unsigned)wait_sweeps) {
                            fail = 0;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    s_pick[1] = fail;
                }
                __syncthreads();
                if (s_pick[1]) err = 1;
                __syncthreads();
            }
            if (STAMP && stamp) {
                B.dbg[kOvDbgLaunch + 8 * ((staged0 / 16) & 7) + 1] = ov_now();
                B.dbg[(size_t)kOvStampPivots * kOvStampsPerPivot + 0] = my_xcc;
                B.dbg[(size_t)kOvStampPivots * kOvStampsPerPivot + 1] = l2 ? 1u : 0u;
            }
        }
        for (int q = 1; q <= K && !err; ++q) {
            const int64_t pidx = staged0 + q - 1;
            OV_STAMP(q, 0);
            const double* bprev = B.bvec + (size_t)(pidx & 1) * Rp;
            double* bnew = B.bvec + (size_t)((pidx + 1) & 1) * Rp;
            double* colq = fcolN + (size_t)(q - 1) * Rp;

            // ---- entering column (:152-167) from the partials of the previous head ----
            int e;
            if (kOvWaveHandoff) {  // every wave collects the G x kOvWPG partials (the barrier)
                int fail = 0;
                const Cand zc = gr_collect_waves(B.gran + (size_t)(pidx & 1) * 3 * kOvParts,
                                                 G * kOvWPG, (unsigned)(2 * pidx + 1), 0.0, &fail);
                e = zc.i;
                if (fail) {
                    err = 1;
                    break;
                }
            } else {
                if (tid < kWave) {  // wave 0 collects the G partials (this is also the barrier)
                    int fail = 0;
                    const Cand zc = gr_collect(B.gran + (size_t)(pidx & 1) * 3 * kOvParts, G,
                                               (unsigned)(2 * pidx + 1), 0.0, &fail);
                    if (tid == 0) {
                        s_pick[0] = zc.i;
                        s_pick[1] = fail;
                    }
                }
                __syncthreads();
                e = s_pick[0];
                if (s_pick[1]) {
                    err = 1;
                    break;
                }
            }
            OV_STAMP(q, 1);  // entering column known
            if (e < 0) {
                pend_out = LPR_OK_OPTIMAL;
                break;
            }
            // ---- column e after all earlier pivots, this lane's row(s); one trip ----
            // pv: lane t < kb: p_t[e] of the block being swept; lane 16 + t, t < q - 1: p_t[e] of
            // this block's earlier pivots; every other lane +0.0.  Steps of unused pivots are exact
            // no-ops (x - (+0 * +0) = x), so both chains are straight-line code of kOvMax steps.
            double cq = have_i ? Tin[(size_t)i_first * ld + e] : 0.0;
            double pv = 0.0;
            if (lane < kb) pv = prowA[(size_t)lane * ld + e];
            else if (lane >= kOvMax && lane - kOvMax < q - 1)
                pv = xld(&prowN[(size_t)(lane - kOvMax) * ld + e]);
            if (STAMP) {
                __builtin_amdgcn_s_waitcnt(0);
                OV_STAMP(q, 2);  // this wave's column gather has arrived
                OV_STAMP(q, 3);
            }
#pragma unroll
            for (int t = 0; t < kOvMax; ++t) {  // through the block being swept
                const double pt = ov_rl(pv, t);
                const double prod = fA[t] * pt;
                const double d = cq - prod;
                cq = ov_bitsel(maskA, t, pt, d);  // i_first == rA[t]: the pivot row keeps p_t[e]
            }
#pragma unroll
            for (int t = 0; t < kOvMax; ++t) {  // through this block's earlier pivots
                const double pt = ov_rl(pv, kOvMax + t);
                const double prod = myf[t] * pt;
                const double d = cq - prod;
                cq = ov_bitsel(maskN, t, pt, d);
            }
#pragma unroll
            for (int t = 0; t < kOvMax; ++t)
                if (t == q - 1) myf[t] = cq;
            // FindLeavingVariable (:169-191) on this lane's rows
            Cand rc;
            rc.v = DBL_MAX;
            rc.i = -1;
            if (have_i) {
                hst(&colq[i_first], cq, l2);
                if (i_first >= 1 && cq > 1e-9) {
                    const double ratio = ieee_div(myb, cq);
                    if (ratio >= 0 && ratio < DBL_MAX) {  // :184, minRatio starts at MaxValue
                        rc.v = ratio;
                        rc.i = i_first;
                    }
                }
            }
            for (int i = i_first + G * nt; i < R; i += G * nt) {  // further rows of this lane
                double c = Tin[(size_t)i * ld + e];
#pragma unroll
                for (int t = 0; t < kOvMax; ++t) {
                    if (t < kb) {
                        const double pt = ov_rl(pv, t);
                        if (i == __builtin_amdgcn_readlane(rAv, t)) {
                            c = pt;
                        } else {
                            const double prod = fcolA[(size_t)t * Rp + i] * pt;
                            c = c - prod;
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < kOvMax; ++t) {
                    if (t < q - 1) {
                        const double pt = ov_rl(pv, kOvMax + t);
                        if (i == __builtin_amdgcn_readlane(rNv, t)) {
                            c = pt;
                        } else {
                            const double prod = fcolN[(size_t)t * Rp + i] * pt;
                            c = c - prod;
                        }
                    }
                }
                hst(&colq[i], c, l2);
                if (c > 1e-9) {
                    const double ratio = ieee_div(bprev[i], c);
                    if (ratio >= 0 && ratio < rc.v) {
                        rc.v = ratio;
                        rc.i = i;
                    }
                }
            }
            // this workgroup's (ratio, row) minimum -> its partial, published once its column slice
            // has drained; then wave 0 collects the G partials: the leaving row (:169-191)
            OV_STAMP(q, 4);
            OV_STAMP(q, 5);
            int r;
            if (kOvWaveHandoff) {
                ov_publish_wave_min(rc, B.gran + ((size_t)2 * kOvParts + g * kOvWPG + tid / kWave) * 3,
                                    (unsigned)(2 * pidx + 2), l2);
                OV_STAMP(q, 6);  // column slice drained, partial published
                int fail = 0;
                const Cand rr = gr_collect_waves(B.gran + (size_t)2 * 3 * kOvParts, G * kOvWPG,
                                                 (unsigned)(2 * pidx + 2), DBL_MAX, &fail);
                r = rr.i;
                if (fail) {
                    err = 1;
                    break;
                }
            } else {
                ov_publish_min(rc, lds_v, lds_i, B.gran + (size_t)(2 * kOvParts + g) * 3,
                               (unsigned)(2 * pidx + 2), l2);
                OV_STAMP(q, 6);  // column slice drained, partial published
                if (tid < kWave) {
                    int fail = 0;
                    const Cand rr = gr_collect(B.gran + (size_t)2 * 3 * kOvParts, G,
                                               (unsigned)(2 * pidx + 2), DBL_MAX, &fail);
                    if (tid == 0) {
                        s_pick[0] = rr.i;
                        s_pick[1] = fail;
                    }
                }
                __syncthreads();
                r = s_pick[0];
                if (s_pick[1]) {
                    err = 1;
                    break;
                }
            }
            OV_STAMP(q, 7);  // leaving row known
            if (r < 0) {
                pend_out = LPR_UNBOUNDED;
                break;
            }
            if (mx > 0 && pidx >= mx) {
                pend_out = LPR_PIVOT_LIMIT;
                break;
            }

            // ---- row r after all earlier pivots, normalised (:199); next Z row; one trip ----
            // fv: lane t < kb: f_t[r] of the block being swept; lane 16 + t, t < q - 1: f_t[r] of
            // this block; lane 32: T[r, rhs]; lane 33: the pivot element T[r, e]; lane 34: the Z
            // row's factor T[0, e]; others +0.0.  The row's RHS entry goes through the same chain
            // as the lane's own column pair (independent chains, interleaved by the scheduler), so
            // p_q[rhs] needs neither a lane of its own nor a barrier.
            double2 w = have_c ? Tin2[(size_t)r * ld2 + c2_first] : make_double2(0.0, 0.0);
            double fv = 0.0;
            {
                const double* src = nullptr;
                if (lane < kb) src = fcolA + (size_t)lane * Rp + r;
                else if (lane >= kOvMax && lane - kOvMax < q - 1)
                    src = fcolN + (size_t)(lane - kOvMax) * Rp + r;
                else if (lane == 32) src = Tin + (size_t)r * ld + rhs;
                else if (lane == 33) src = colq + r;
                else if (lane == 34) src = colq;
                if (src) fv = xld(src);
            }
            if (STAMP) {
                __builtin_amdgcn_s_waitcnt(0);
                OV_STAMP(q, 8);  // this wave's row gather has arrived
            }
            double wr = ov_rl(fv, 32);
            const double p = ov_rl(fv, 33);   // the pivot element T[r, e] ...
            const double f0 = ov_rl(fv, 34);  // ... and the Z row's factor T[0, e]
            // r was the pivot row of earlier pivot t: bit t (wave-uniform)
            const unsigned ra = (unsigned)__ballot(lane < kb && rAv == r);
            const unsigned rn = (unsigned)__ballot(lane < q - 1 && rNv == r);
            if ((ra | rn) == 0u) {  // the common case: r has not been a pivot row in these blocks
#pragma unroll
                for (int t = 0; t < kOvMax; ++t) {  // through the block being swept
                    const double f = ov_rl(fv, t);
                    const double px = f * pA[t].x;
                    const double py = f * pA[t].y;
                    const double pr = f * ov_rl(parhsAv, t);
                    w.x = w.x - px;
                    w.y = w.y - py;
                    wr = wr - pr;
                }
#pragma unroll
                for (int t = 0; t < kOvMax; ++t) {  // through this block's earlier pivots
                    const double f = ov_rl(fv, kOvMax + t);
                    const double px = f * myp[t].x;
                    const double py = f * myp[t].y;
                    const double pr = f * ov_rl(prhsNv, t);
                    w.x = w.x - px;
                    w.y = w.y - py;
                    wr = wr - pr;
                }
            } else {
#pragma unroll
                for (int t = 0; t < kOvMax; ++t) {
                    if ((ra >> t) & 1u) {
                        w = pA[t];
                        wr = ov_rl(parhsAv, t);
                    } else {
                        const double f = ov_rl(fv, t);
                        const double px = f * pA[t].x;
                        const double py = f * pA[t].y;
                        const double pr = f * ov_rl(parhsAv, t);
                        w.x = w.x - px;
                        w.y = w.y - py;
                        wr = wr - pr;
                    }
                }
#pragma unroll
                for (int t = 0; t < kOvMax; ++t) {
                    if ((rn >> t) & 1u) {
                        w = myp[t];
                        wr = ov_rl(prhsNv, t);
                    } else {
                        const double f = ov_rl(fv, kOvMax + t);
                        const double px = f * myp[t].x;
                        const double py = f * myp[t].y;
                        const double pr = f * ov_rl(prhsNv, t);
                        w.x = w.x - px;
                        w.y = w.y - py;
                        wr = wr - pr;
                    }
                }
            }
            // p_q[rhs] (every lane works it out for itself) and this lane's pair of the normalised
            // row (:199 true division): three quotients behind one check
            double dq[3];
            {
                const double dn[3] = {wr, w.x, w.y}, dd[3] = {p, p, p};
                ieee_div_n<3>(dn, dd, dq);
            }
            const double prhs = dq[0];
            if (lane == q - 1) {
                prhsNv = prhs;
                rNv = r;
            }
            if (have_i && i_first == r) maskN |= 1u << (q - 1);
            Cand n;
            n.v = 0.0;
            n.i = -1;
            if (have_c) {
                const int j = 2 * c2_first;
                double2 pq;
                pq.x = (j < C) ? dq[1] : 0.0;
                pq.y = (j + 1 < C) ? dq[2] : 0.0;
#pragma unroll
                for (int t = 0; t < kOvMax; ++t)
                    if (t == q - 1) myp[t] = pq;
                hst(&prowN[(size_t)(q - 1) * ld + j], pq.x, l2);
                hst(&prowN[(size_t)(q - 1) * ld + j + 1], pq.y, l2);
                const double mxp = f0 * pq.x;  // :208 product rounded, then the difference
                const double myp2 = f0 * pq.y;
                myz.x = myz.x - mxp;
                myz.y = myz.y - myp2;
                zrow2[c2_first] = myz;  // read again by the next launch
                if (j < C - 1 && myz.x < n.v) {
                    n.v = myz.x;
                    n.i = j;
                }
                if (j + 1 < C - 1 && myz.y < n.v) {
                    n.v = myz.y;
                    n.i = j + 1;
                }
            }
            for (int c2 = c2_first + G * nt; c2 < ld2; c2 += G * nt) {  // further column pairs
                double2 ww = Tin2[(size_t)r * ld2 + c2];
                double2 z = zrow2[c2];
#pragma unroll
                for (int t = 0; t < kOvMax; ++t) {
                    if (t < kb) {
                        const double2 ps = prowA2[(size_t)t * ld2 + c2];
                        if ((ra >> t) & 1u) {
                            ww = ps;
                        } else {
                            const double f = ov_rl(fv, t);
                            const double px = f * ps.x;
                            const double py = f * ps.y;
                            ww.x = ww.x - px;
                            ww.y = ww.y - py;
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < kOvMax; ++t) {
                    if (t < q - 1) {
                        double2 ps;
                        ps.x = prowN[(size_t)t * ld + 2 * c2];
                        ps.y = prowN[(size_t)t * ld + 2 * c2 + 1];
                        if ((rn >> t) & 1u) {
                            ww = ps;
                        } else {
                            const double f = ov_rl(fv, kOvMax + t);
                            const double px = f * ps.x;
                            const double py = f * ps.y;
                            ww.x = ww.x - px;
                            ww.y = ww.y - py;
                        }
                    }
                }
                const int j = 2 * c2;
                double2 pq;
                pq.x = (j < C) ? ieee_div(ww.x, p) : 0.0;
                pq.y = (j + 1 < C) ? ieee_div(ww.y, p) : 0.0;
                hst(&prowN[(size_t)(q - 1) * ld + j], pq.x, l2);
                hst(&prowN[(size_t)(q - 1) * ld + j + 1], pq.y, l2);
                const double mxp = f0 * pq.x;
                const double myp2 = f0 * pq.y;
                z.x = z.x - mxp;
                z.y = z.y - myp2;
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
            // ---- RHS column after this pivot ----
            if (have_i) {
                const double prod = cq * prhs;
                myb = (i_first == r) ? prhs : myb - prod;
                hst(&bnew[i_first], myb, l2);
            }
            for (int i = i_first + G * nt; i < R; i += G * nt) {  // further rows of this lane
                const double prod = colq[i] * prhs;
                hst(&bnew[i], (i == r) ? prhs : bprev[i] - prod, l2);
            }
            OV_STAMP(q, 9);
            if (lead && tid == 0) {
                co->r[q - 1] = r;
                B.basis[r - 1] = e;  // :142
                if (pidx < log_cap) {
                    B.log[2 * pidx] = r;
                    B.log[2 * pidx + 1] = e;
                }
            }
            count = q;
            // this workgroup's Z-row partial, published once its row slice / RHS entries have
            // drained; the next head collects the G of them.  The partial of the launch's LAST
            // head is collected by the next launch, whose workgroups may sit on another XCD: that
            // one always goes through the memory side.
            OV_STAMP(q, 10);
            if (kOvWaveHandoff)
                ov_publish_wave_min(n, B.gran + ((size_t)((pidx + 1) & 1) * kOvParts + g * kOvWPG +
                                                 tid / kWave) * 3,
                                    (unsigned)(2 * (pidx + 1) + 1), l2 && q < K);
            else
                ov_publish_min(n, lds_v, lds_i,
                               B.gran + ((size_t)((pidx + 1) & 1) * kOvParts + g) * 3,
                               (unsigned)(2 * (pidx + 1) + 1), l2 && q < K);
            OV_STAMP(q, 11);  // row slice drained, partial published
        }
    }
#undef OV_STAMP

    if (STAMP && stamp) B.dbg[kOvDbgLaunch + 8 * ((staged0 / 16) & 7) + 2] = ov_now();
    if (lead && tid == 0) {  // the next launch's view (fields owned by the heads)
        const bool staged_now = (status == kRunning && pend_in == kRunning);
        if (wait_sweeps > 0 && !staged_now) {
            // a launch that stages nothing has not waited yet: the block written below is the
            // one the sweep of the PREVIOUS step is still reading (kdone, r[], slot, ...)
            unsigned spins = 0;
            while (__hip_atomic_load(B.sflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <
                   (unsigned)wait_sweeps) {
                if (++spins > kOvSpinMax) {
                    err = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        co->status = err ? LPR_DEVICE_ERROR : status_out;
        co->pending = pend_out;
        co->kdone = staged_now ? count : 0;
        co->slot = staged_now ? (sa ^ 1) : sa;
        co->staged = staged0 + (staged_now ? count : 0);
        co->max_iter = mx;
        co->log_cap = log_cap;
        co->error = err | ci->error;
        co->head_xcc = l2 ? (int)my_xcc : (staging ? -1 : ci->head_xcc);
        B.bar[lp ^ 1] = 0u;
        if (wait_sweeps >= 0) {  // the fields of the sweep running beside this launch
            const int ks = (status == kRunning) ? ci->kdone : 0;
            co->applied = ci->applied + ks;
            co->cur = (ks > 0) ? (ci->cur ^ 1) : ci->cur;
            co->sweep = ci->sweep ^ 1;
        }
        if (solo) {
            B.tileq[1] = 0u;  // the in-place sweep that follows always runs on control block 1
            co->applied = ci->applied;
            co->cur = ci->cur;
            co->sweep = ci->sweep;
        }
    }
    if (heads_done >= 0) {
        // The sweep of the next step does not wait for this launch by an event: it polls
        // B.sflag[1] as it starts (ov_tiles).  So this launch says itself when everything it
        // staged (pivot rows, factor columns, the control block) has left this XCD's L2: every
        // wave drains its stores, the workgroup's lane 0 writes the L2 back (agent-scope release)
        // and arrives; the workgroup that arrives last publishes the count.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned old = __hip_atomic_fetch_add(B.sflag + 2, 1u, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
            s_pick[0] = (old + 1u == (unsigned)G) ? 1 : 0;
            if (s_pick[0])
                __hip_atomic_store(B.sflag + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_pick[0]) {  // the workgroup that arrived last: the count, and its copies
            if (tid <= kOvDoneCopies)
                __hip_atomic_store(tid == 0 ? B.sflag + 1 : B.sflag + (size_t)tid * kOvDoneStride,
                                   (unsigned)heads_done, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            if (STAMP && tid == 0)
                B.dbg[kOvDbgLaunch + 8 * ((staged0 / 16) & 7) + 3] = ov_now();
        }
    }
}

// ------------------------------------------------------------------------------------------
// The sweep of the current block (workgroups [G, ...)): every element through kdone pivots
// (:202-210 each) in registers, buffer cur -> buffer cur ^ 1.
//
// A 256-lane workgroup owns a tile of kOvTileRows rows x 256 double2 columns.  A lane keeps its
// slice of the K normalised pivot rows (p[s], 16 B each) in registers for the whole tile; the
// factors f_s[i] are the same for every lane of the workgroup, so they come through the SCALAR
// unit: `fcol` is a __restrict__ const kernel argument that nothing in the launch writes, the
// index is wave-uniform, and the compiler turns the TR consecutive rows of one pivot into one
// s_load_dwordx16 -- no LDS, no workgroup barrier, no lgkmcnt stall inside the multiply-subtract
// chains, and the multiplies read f straight from SGPRs.
// Rows of the tile that are pivot rows of the block (at most kOvMax of the R) would need a test per
// (row, pivot); instead the tile runs straight-line code for every row, does not store those rows,
// and recomputes them afterwards (normalised row p[s0], then the later pivots of the block).
template <typename T>
__device__ __forceinline__ T ov_ld_stream(const T* p) {  // streamed once (out-of-place sweep)
    return __builtin_nontemporal_load(p);
}

typedef double ov_v2d __attribute__((ext_vector_type(2)));

// TR rows of one lane through all kOvMax pivots of the block (the straight-line case).
// The factors of pivot s + 1 are requested as soon as those of pivot s have arrived, so one scalar
// load is in flight behind the TR multiply-subtract pairs of the current pivot.  The empty asm
// pins that order by data dependence (it "reads" a factor of pivot s and "rewrites" the offset
// the load of pivot s + 1 goes through); without it the compiler hoists all kOvMax loads to the
// top of the chunk and spills them (2x the registers, measured in the ISA).
template <int TR>
__device__ __forceinline__ void ov_chunk(ov_v2d (&x)[TR], const ov_v2d (&p)[kOvMax],
                                         const double* __restrict__ fci, int Rp) {
    if (LPR_OV_DIAG & 1) return;
    constexpr int kSteps = (LPR_OV_DIAG & 4) ? kOvMax / 2 : kOvMax;
    double fn[TR];
#pragma unroll
    for (int k = 0; k < TR; ++k) fn[k] = fci[k];  // wave-uniform: one scalar load
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
        double f[TR];
#pragma unroll
        for (int k = 0; k < TR; ++k) f[k] = fn[k];
        if (s + 1 < kSteps) {
            size_t off = (size_t)(s + 1) * Rp;  // (the pointer itself must stay derived from the
            asm volatile("" : "+s"(off) : "s"(f[0]));  // __restrict__ argument: scalar loads)
#pragma unroll
            for (int k = 0; k < TR; ++k) fn[k] = fci[off + k];
        }
#pragma unroll
        for (int k = 0; k < TR; ++k) {
            const double px = f[k] * p[s].x;  // product rounded ...
            const double py = f[k] * p[s].y;
            x[k].x = x[k].x - px;             // ... then the difference (:208)
            x[k].y = x[k].y - py;
        }
        // pivot s is finished before the factors of pivot s + 2 are requested (program order kept)
#pragma unroll
        for (int k = 0; k < TR; ++k) asm volatile("" : "+v"(x[k]));
    }
}

template <int TR, bool INPLACE>
__device__ __forceinline__ void ov_rows_load(ov_v2d (&x)[TR], const ov_v2d* src, int ld2) {
#pragma unroll
    for (int k = 0; k < TR; ++k)
        x[k] = (INPLACE || (LPR_OV_DIAG & 8)) ? src[(size_t)k * ld2]
                                              : ov_ld_stream(&src[(size_t)k * ld2]);
}

template <int TR, bool INPLACE>
__device__ __forceinline__ void ov_rows_store(const ov_v2d (&x)[TR], ov_v2d* dst, int ld2,
                                              unsigned skip) {
#pragma unroll
    for (int k = 0; k < TR; ++k) {
        if ((skip >> k) & 1u) continue;  // a pivot row of the block: recomputed afterwards
        if (INPLACE || (LPR_OV_DIAG & 8)) dst[(size_t)k * ld2] = x[k];
        else if (LPR_OV_DIAG & 64)
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(&dst[(size_t)k * ld2]),
                         "v"(x[k])
                         : "memory");
        else if (LPR_OV_DIAG & 128)
            asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(&dst[(size_t)k * ld2]),
                         "v"(x[k])
                         : "memory");
        else __builtin_nontemporal_store(x[k], &dst[(size_t)k * ld2]);
    }
}

// TR rows per chunk; DB: the next chunk's rows are requested before the current chunk is computed
// (two register sets), so a wave always has loads in flight.
//
// Work distribution: the launch is PERSISTENT (a few workgroups per CU); every workgroup takes the
// next tile from a counter in memory until none is left.  Workgroups are dealt to the 8 XCDs round
// robin, so with a static tile -> workgroup map the slowest XCD sets the time of the sweep -- and
// one XCD is slow by design: the one that hosts the loop heads of the next block (ov_heads_rich),
// which run beside this sweep with a high priority and one wave per SIMD.  With the queue every XCD
// takes what it can, and the heads' XCD takes nothing: its workgroups leave at once when the
// previous launch's heads sat there (the hint `head_xcc` of the control block), or as soon as this
// launch's heads have said where they are (B.hx) -- the heads then run as fast as with nothing
// beside them.  Wrong or missing hints cost time, never correctness: any workgroup may take any
// tile.  `static_tile` >= 0: one given tile (the one-launch form k_ov_step has no queue).
// TROWS rows are processed; the row tiles are counted in units of `unit` rows (>= TROWS) and this
// call takes the TROWS rows at offset `off` inside its unit (the half tiles of the sweep's tail)
template <int TR, bool DB, bool INPLACE, int TROWS = kOvTileRows>
__device__ __forceinline__ void ov_one_tile(const OvBuffers& B, const OvCtl* ci,
                                            const double* __restrict__ fc,
                                            const ov_v2d* __restrict__ prow2, int tb, int ld,
                                            int R, int Rp, int K, int cur, int unit = TROWS,
                                            int off = 0) {
    typedef ov_v2d v2d;
    const int ld2 = ld >> 1;
    const int nct = (ld2 + kOvNT - 1) / kOvNT;
    const int nrt = (R + unit - 1) / unit;
    if (LPR_OV_DIAG & 16) {  // workgroup b sits on XCD b % 8: give each XCD consecutive tiles
        const int per = (nct * nrt + 7) / 8;
        tb = (tb % 8) * per + tb / 8;
    }
    int ct = tb % nct, rt = tb / nct;
    if (LPR_OV_DIAG & 32) {
        rt = tb % nrt;
        ct = tb / nrt;
        if (ct >= nct) return;
    }
    if (rt >= nrt) return;
    if (ci->sweep & 1) {
        ct = nct - 1 - ct;
        rt = nrt - 1 - rt;
    }
    const int c2 = ct * kOvNT + threadIdx.x;
    if (c2 >= ld2) return;
    // in place every element is read and written by the same lane; out of place the buffers differ
    const v2d* Tin2 = reinterpret_cast<const v2d*>(B.Tb[cur]);
    v2d* Tout2 = reinterpret_cast<v2d*>(B.Tb[INPLACE ? cur : (cur ^ 1)]);
    // this lane's slice of the normalised pivot rows, for the whole tile: all kOvMax loads in
    // flight at once (rows >= K of the staging slot are stale but valid memory; they are not used)
    v2d p[kOvMax];
#pragma unroll
    for (int s = 0; s < kOvMax; ++s) {
        if (LPR_OV_DIAG & 2) p[s] = v2d{(double)c2, (double)s};
        else p[s] = prow2[(size_t)s * ld2 + c2];
    }
    const int ibase = rt * unit + off;
    const int iend = min(R, ibase + TROWS);
    if (ibase >= R) return;

    if (K == kOvMax && iend - ibase == TROWS) {
        // rows of this tile that are pivot rows of the block, as a bit mask (wave-uniform)
        unsigned long long prmask = 0ull;  // (64 bits: tiles of up to 64 rows)
#pragma unroll
        for (int s = 0; s < kOvMax; ++s) {
            const int r = ci->r[s];
            if (r >= ibase && r < iend) prmask |= 1ull << (r - ibase);
        }
        const v2d* src = Tin2 + (size_t)ibase * ld2 + c2;
        v2d* dst = Tout2 + (size_t)ibase * ld2 + c2;
        const size_t step = (size_t)TR * ld2;
        if (DB) {
            v2d xa[TR], xb[TR];
            ov_rows_load<TR, INPLACE>(xa, src, ld2);
#pragma unroll 1
            for (int j = 0; j < TROWS; j += 2 * TR) {
                ov_rows_load<TR, INPLACE>(xb, src + step, ld2);
                ov_chunk<TR>(xa, p, fc + ibase + j, Rp);
                ov_rows_store<TR, INPLACE>(xa, dst, ld2, (unsigned)(prmask >> j));
                if (j + 2 * TR < TROWS) ov_rows_load<TR, INPLACE>(xa, src + 2 * step, ld2);
                ov_chunk<TR>(xb, p, fc + ibase + j + TR, Rp);
                ov_rows_store<TR, INPLACE>(xb, dst + step, ld2, (unsigned)(prmask >> (j + TR)));
                src += 2 * step;
                dst += 2 * step;
            }
        } else {
#pragma unroll 1
            for (int j = 0; j < TROWS; j += TR) {
                v2d x[TR];
                ov_rows_load<TR, INPLACE>(x, src, ld2);
                ov_chunk<TR>(x, p, fc + ibase + j, Rp);
                ov_rows_store<TR, INPLACE>(x, dst, ld2, (unsigned)(prmask >> j));
                src += step;
                dst += step;
            }
        }
        if (prmask) {
#pragma unroll
            for (int s0 = 0; s0 < kOvMax; ++s0) {
                const int i = ci->r[s0];
                if (i < ibase || i >= iend) continue;
                bool again = false;  // the row pivots again later in the block: that one stores it
#pragma unroll
                for (int s1 = s0 + 1; s1 < kOvMax; ++s1) again = again || (ci->r[s1] == i);
                if (again) continue;
                v2d t = p[s0];  // the pivot row keeps the normalised values (:199) ...
#pragma unroll
                for (int s1 = s0 + 1; s1 < kOvMax; ++s1) {  // ... and is an ordinary row afterwards
                    const double f = fc[(size_t)s1 * Rp + i];
                    const double px = f * p[s1].x;
                    const double py = f * p[s1].y;
                    t.x = t.x - px;
                    t.y = t.y - py;
                }
                v2d* d = &Tout2[(size_t)i * ld2 + c2];
                if (INPLACE) *d = t;
                else __builtin_nontemporal_store(t, d);
            }
        }
        return;
    }
    // partial blocks (the last block of a solve) and the ragged last row tile
    int rr[kOvMax];
#pragma unroll
    for (int s = 0; s < kOvMax; ++s) rr[s] = (s < K) ? ci->r[s] : -1;
    for (int i0 = ibase; i0 < iend; i0 += TR) {  // TR rows in flight per lane
        v2d x[TR];
#pragma unroll
        for (int k = 0; k < TR; ++k) {
            const int i = i0 + k;
            if (i < iend) {
                const v2d* src = &Tin2[(size_t)i * ld2 + c2];
                x[k] = INPLACE ? *src : ov_ld_stream(src);
            }
        }
#pragma unroll
        for (int k = 0; k < TR; ++k) {
            const int i = i0 + k;
            if (i < iend) {
                v2d t = x[k];
#pragma unroll
                for (int s = 0; s < kOvMax; ++s) {
                    if (s < K) {
                        if (i == rr[s]) {
                            t = p[s];  // the pivot row keeps the normalised values (:199)
                        } else {
                            const double f = fc[(size_t)s * Rp + i];
                            const double px = f * p[s].x;  // product rounded ...
                            const double py = f * p[s].y;
                            t.x = t.x - px;                // ... then the difference (:208)
                            t.y = t.y - py;
                        }
                    }
                }
                v2d* dst = &Tout2[(size_t)i * ld2 + c2];
                if (INPLACE) *dst = t;
                else __builtin_nontemporal_store(t, dst);
            }
        }
    }
}

// avoid: 0 = never leave an XCD to the heads, 1 = live word only (B.hx), 2 = hint + live word
template <int TR, bool DB, bool INPLACE, int TROWS = kOvTileRows>
__device__ __forceinline__ void ov_tiles(const OvBuffers& B, const double* __restrict__ fcol,
                                         const double* __restrict__ prow, int ld, int R, int Rp,
                                         int G, int lp, int static_tile, int avoid,
                                         bool write_ctl = true, int sweeps_done = -1,
                                         int wait_heads = -1, int hint_xcc = -1) {
    static_assert(TROWS % TR == 0 && (!DB || TROWS % (2 * TR) == 0) && TROWS <= 64,
                  "tile rows: a multiple of the chunks in flight, and one bit each in prmask");
    __shared__ int s_tile;
    const bool first_wg = ((int)blockIdx.x == G);
    // this launch has started, so every earlier sweep of the stream is complete and its stores
    // are visible (kernel boundary): tell the heads that wait for exactly that
    if (first_wg && threadIdx.x == 0 && sweeps_done >= 0)
        __hip_atomic_store(B.sflag, (unsigned)sweeps_done, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (first_wg && threadIdx.x == 0 && !INPLACE && B.dbg)
        B.dbg[kOvDbgLaunch + 8 * (wait_heads > 0 ? (wait_heads & 7)
                                                  : (int)((B.ctl[lp].applied / 16 + 1) & 7)) + 4] = ov_now();
    if (wait_heads > 0) {
        // No event ordered this launch behind the loop heads that staged its block: it followed
        // the previous sweep at once (a wait packet costs the command processor ~10 us whether it
        // has to wait or not) and every workgroup asks the heads' own completion word, B.sflag[1],
        // before it reads anything they wrote.  Usually the answer is there.  If it is not, the
        // workgroups that sit on the XCD the heads are expected on (the host's hint, from its last
        // poll) leave instead of waiting -- a loop-head launch that cannot become resident never
        // publishes -- and the others poll.  Any workgroup may leave: the tiles are dealt from a
        // queue.  Only the first one stays whatever happens (it resets the other queue).
        if (threadIdx.x == 0) {
            int go = 0;  // 1: proceed, 0: leave, -1: gave up
            const unsigned* word =
                B.sflag + (size_t)(1 + ((int)blockIdx.x & (kOvDoneCopies - 1))) * kOvDoneStride;
            auto ready = [&]() {
                return __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >=
                       (unsigned)wait_heads;
            };
            if (!first_wg && hint_xcc >= 0 && (int)ov_xcc_id() == hint_xcc) {
                go = 0;  // (whether the word is there or not: these workgroups would leave anyway)
            } else if (ready()) {
                go = 1;
            } else {
                go = -1;
                for (unsigned spins = 0; spins < kOvSpinMax; ++spins) {
                    __builtin_amdgcn_s_sleep(8);
                    if (ready()) {
                        go = 1;
                        break;
                    }
                }
                if (go < 0)
                    __hip_atomic_store(B.sflag + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_tile = go;
        }
        __syncthreads();
        const int go = s_tile;
        __syncthreads();
        if (go <= 0) return;
        if (first_wg && threadIdx.x == 0 && B.dbg) B.dbg[kOvDbgLaunch + 8 * (wait_heads & 7) + 5] = ov_now();
        // what the heads stored is in memory (they wrote their L2 back before publishing); nothing
        // of it can be in this XCD's caches from before the wait (they were invalidated when the
        // launch started and nothing has been read since), the fence makes that explicit.  Not on
        // the heads' own XCD (only the first workgroup can still be there): it reads through the
        // very L2 they wrote, and an invalidate of that L2 under the running heads of the NEXT
        // step cost them 30 us (tools/step_anatomy.py, every other step 210 us instead of 185).
        if (!(hint_xcc >= 0 && (int)ov_xcc_id() == hint_xcc))
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __builtin_amdgcn_s_dcache_inv();
    }
    const OvCtl* ci = B.ctl + lp;
    const int K = (ci->status == kRunning) ? ci->kdone : 0;
    const int cur = ci->cur;
    if (first_wg && threadIdx.x == 0) {  // the next launch's view (fields owned by the sweep)
        OvCtl* co = B.ctl + (lp ^ 1);
        if (write_ctl) {  // (the heads write these when they run ahead of the event, see there)
            co->applied = ci->applied + K;
            co->cur = (K > 0 && !INPLACE) ? (cur ^ 1) : cur;
            co->sweep = ci->sweep ^ 1;
        }
        B.tileq[lp ^ 1] = 0u;  // nobody touches the other queue during this launch
        if (INPLACE) {  // no heads in this launch: their fields are carried over here
            co->status = ci->status;
            co->pending = ci->pending;
            co->kdone = 0;
            co->slot = ci->slot;
            co->staged = ci->staged;
            co->max_iter = ci->max_iter;
            co->log_cap = ci->log_cap;
            co->error = ci->error;
            co->head_xcc = ci->head_xcc;
            B.bar[lp ^ 1] = 0u;
        }
    }
    if (K <= 0) return;
    const int sa = ci->slot;
    const double* __restrict__ fc = fcol + (size_t)sa * kOvMax * Rp;
    const ov_v2d* __restrict__ prow2 =
        reinterpret_cast<const ov_v2d*>(prow + (size_t)sa * kOvMax * ld);
    if (static_tile >= 0) {
        ov_one_tile<TR, DB, INPLACE, TROWS>(B, ci, fc, prow2, static_tile, ld, R, Rp, K, cur);
        return;
    }
    const int ld2 = ld >> 1;
    const int ntiles = ((ld2 + kOvNT - 1) / kOvNT) * ((R + TROWS - 1) / TROWS);
    // The workgroups finish their last tiles at different times and the chip drains: the last
    // quarter of the queue is handed out as HALF tiles (a pivot-row slice load per 16 rows instead
    // of 32 there, a shorter tail: 64-row tiles everywhere measured 181 us, 32-row 174, 16-row 172).
    constexpr bool kHalfTail = (TROWS / 2) % TR == 0 && TROWS == kOvTileRows && (LPR_OV_DIAG & 256) == 0;
#ifndef LPR_OV_TAIL_DIV
#define LPR_OV_TAIL_DIV 4
#endif
    const int tsplit = kHalfTail ? (ntiles - ntiles / LPR_OV_TAIL_DIV) : ntiles;
    const int qtiles = tsplit + 2 * (ntiles - tsplit);
    // do loop heads that stage a block run beside this launch, and where?
    const bool heads_beside = !INPLACE && avoid > 0 && ci->status == kRunning &&
                              ci->pending == kRunning;
    const unsigned my_xcc = ov_xcc_id();
    if (heads_beside && avoid > 1 && ci->head_xcc == (int)my_xcc) return;
    const unsigned xepoch = (unsigned)(ci->staged + 1);
    for (;;) {
        if (threadIdx.x == 0) {
            int tile = -1;
            if (heads_beside) {
                const unsigned long long v =
                    __hip_atomic_load(B.hx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(v >> 32) == xepoch && ((unsigned)v & 0x1ffu) == (0x100u | my_xcc))
                    tile = qtiles;  // this launch's heads share this XCD: leave it to them
            }
            if (tile < 0)
                tile = (int)__hip_atomic_fetch_add(B.tileq + lp, 1u, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
            s_tile = tile;
        }
        __syncthreads();
        const int tile = s_tile;
        __syncthreads();
        if (tile >= qtiles) return;
        if (tile < tsplit) {
            ov_one_tile<TR, DB, INPLACE, TROWS>(B, ci, fc, prow2, tile, ld, R, Rp, K, cur);
        } else {  // the tail of the queue: half tiles
            const int u = tile - tsplit;
            ov_one_tile<TR, false, INPLACE, TROWS / 2>(B, ci, fc, prow2, tsplit + (u >> 1), ld, R, Rp,
                                                       K, cur, TROWS, (u & 1) * (TROWS / 2));
        }
    }
}

template <int TR, bool DB>
__global__ __launch_bounds__(kOvNT) void k_ov_step(const OvBuffers B,
                                                   const double* __restrict__ fcol_ro,
                                                   const double* __restrict__ prow_ro, int ld,
                                                   int R, int C, int Rp, int K, int G, int lp) {
    // fcol_ro / prow_ro alias B.fcol / B.prow, which the heads of this launch write -- but only the
    // OTHER staging slot (ci->slot ^ 1) than the one the tiles read, so the read-only view holds
    if ((int)blockIdx.x < G)
        ov_heads<kOvNT>(B, ld, R, C, Rp, K, G, lp, false);
    else
        ov_tiles<TR, DB, false>(B, fcol_ro, prow_ro, ld, R, Rp, G, lp, (int)blockIdx.x - G, 0);
}

// The two halves as separate kernels on two streams, running concurrently (variant 0x30tr): same
// protocol as k_ov_step, but each kernel has its own register budget (in k_ov_step the heads'
// registers cap the occupancy of the sweep's tiles and vice versa).
template <int NT, bool STAMP>
__global__ __launch_bounds__(NT) void k_ov2_heads(const OvBuffers B, int ld, int R, int C, int Rp,
                                                  int K, int G, int lp, int spread, int no_l2,
                                                  int wait_sweeps, int heads_done) {
    ov_heads_rich<NT, STAMP>(B, ld, R, C, Rp, K, G, lp, false, spread, no_l2, wait_sweeps,
                             heads_done);
}



template <int TR, bool DB>
__global__ __launch_bounds__(kOvNT) void k_ov2_sweep(const OvBuffers B,
                                                     const double* __restrict__ fcol_ro,
                                                     const double* __restrict__ prow_ro, int ld,
                                                     int R, int Rp, int lp, int avoid,
                                                     int write_ctl, int sweeps_done,
                                                     int wait_heads, int hint_xcc) {
    ov_tiles<TR, DB, false>(B, fcol_ro, prow_ro, ld, R, Rp, 0, lp, -1, avoid, write_ctl != 0,
                            sweeps_done, wait_heads, hint_xcc);
}

// The same two halves as separate launches: all K loop heads of a block in ONE persistent launch
// (no sweep running: nothing to chain through but the block's own pivots), then the sweep in
// place.  Heads always run on control block 0, the sweep on control block 1.
template <int NT, bool STAMP>
__global__ __launch_bounds__(NT) void k_ov_heads(const OvBuffers B, int ld, int R, int C, int Rp,
                                                 int K, int G, int spread, int no_l2) {
    ov_heads_rich<NT, STAMP>(B, ld, R, C, Rp, K, G, 0, true, spread, no_l2, -1);
}

template <int TR, bool DB, int TROWS = kOvTileRows>
__global__ __launch_bounds__(kOvNT) void k_ov_sweep(const OvBuffers B,
                                                    const double* __restrict__ fcol_ro,
                                                    const double* __restrict__ prow_ro, int ld,
                                                    int R, int Rp) {
    // nothing runs beside the in-place sweep: one workgroup per tile, dealt by the hardware (the
    // queue's counter costs a burst of ~1000 atomics on one word at the start of every sweep).
    // TROWS = 8: small tableaux, where 32-row tiles would leave most CUs without a tile
    ov_tiles<TR, DB, true, TROWS>(B, fcol_ro, prow_ro, ld, R, Rp, 0, 1, (int)blockIdx.x, 0);
}

}  // namespace lpr

#ifndef LPR_OV_KERNELS_ONLY
// ---------------------------------------------------------------------------------------------
// host side (the driver loop lives in lpr_engine.hip)

struct lpr_overlap_ctx {
    int rows = 0, ld = 0, Rp = 0;
    lpr::OvBuffers b{};
    lpr::OvCtl* h_ctl = nullptr;  // pinned, 2 entries
    double* h_z = nullptr;        // pinned, 2 entries
    double* T2 = nullptr;         // the second tableau buffer (owned here)
    hipStream_t hstream = nullptr;  // the heads' stream of the two-stream variant
    hipEvent_t ev_h[2] = {nullptr, nullptr}, ev_s[2] = {nullptr, nullptr};
    int ev_idx = 0;
    int steps = 0;                  // launch pairs queued by the current solve call
    hipEvent_t last_sweep = nullptr;  // the event that marks the latest sweep of the call as done
    unsigned* h_flags = nullptr;    // pinned copy of b.sflag[0..3]
    int head_xcc_hint = -1;         // the XCD the loop heads shared at the last poll (-1: unknown)
    bool batch_first = true;        // the next step is the first after the streams were joined
};

namespace lpr {

struct OvPoll {
    int32_t status, pending, kdone, cur, error;
    int64_t applied;
    double z[2];
};

int ov_max_pivots() { return kOvMax; }

void ov_release(lpr_tableau* t) {
    lpr_overlap_ctx* c = static_cast<lpr_overlap_ctx*>(t->ov);
    if (!c) return;
    if (c->hstream) {
        hipStreamSynchronize(c->hstream);
        hipStreamDestroy(c->hstream);
        for (int k = 0; k < 2; ++k) {
            hipEventDestroy(c->ev_h[k]);
            hipEventDestroy(c->ev_s[k]);
        }
    }
    hipFree(c->T2);
    hipFree(c->b.prow);
    hipFree(c->b.fcol);
    hipFree(c->b.zrow);
    hipFree(c->b.bvec);
    hipFree(c->b.zparts);
    hipFree(c->b.rparts);
    hipFree(c->b.gran);
    hipFree(c->b.xgran);
    hipFree(c->b.hx);
    hipFree(c->b.tileq);
    hipFree(c->b.sflag);
    hipFree(c->b.dbg);
    hipFree(c->b.ctl);
    hipFree(c->b.bar);
    if (c->h_ctl) hipHostFree(c->h_ctl);
    if (c->h_z) hipHostFree(c->h_z);
    if (c->h_flags) hipHostFree(c->h_flags);
    delete c;
    t->ov = nullptr;
}

int ov_ensure(lpr_tableau* t, bool second_buffer) {
    lpr_overlap_ctx* c = static_cast<lpr_overlap_ctx*>(t->ov);
    if (c && c->rows == t->rows && c->ld == t->ld && (c->T2 || !second_buffer))
        return LPR_OK_OPTIMAL;
    ov_release(t);
    c = new (std::nothrow) lpr_overlap_ctx();
    if (!c) return LPR_OUT_OF_MEMORY;
    c->rows = t->rows;
    c->ld = t->ld;
    c->Rp = align_up(t->rows, 16);
    hipError_t err = hipSuccess;
    auto chk = [&](hipError_t x) { if (err == hipSuccess) err = x; };
    const size_t D = sizeof(double);
    const size_t tbytes = (size_t)t->rows * t->ld * D;
    if (second_buffer) chk(hipMalloc(&c->T2, tbytes));
    chk(hipMalloc(&c->b.prow, (size_t)2 * kOvMax * c->ld * D));
    chk(hipMalloc(&c->b.fcol, (size_t)2 * kOvMax * c->Rp * D));
    chk(hipMalloc(&c->b.zrow, (size_t)c->ld * D));
    chk(hipMalloc(&c->b.bvec, (size_t)2 * c->Rp * D));
    chk(hipMalloc(&c->b.zparts, (size_t)2 * kOvGroups * sizeof(ZPart)));
    chk(hipMalloc(&c->b.rparts, (size_t)(3 * kOvGroups + 1) * sizeof(double)));
    chk(hipMalloc(&c->b.gran, (size_t)9 * kOvParts * sizeof(unsigned long long)));
    chk(hipMalloc(&c->b.xgran, (size_t)kOvGroups * sizeof(unsigned long long)));
    chk(hipMalloc(&c->b.dbg, kOvDbgWords * sizeof(unsigned long long)));
    chk(hipMalloc(&c->b.hx, 2 * sizeof(unsigned long long)));
    chk(hipMalloc(&c->b.tileq, 4 * sizeof(unsigned)));
    chk(hipMalloc(&c->b.sflag, kOvFlagWords * sizeof(unsigned)));
    chk(hipMalloc(&c->b.ctl, 2 * sizeof(OvCtl)));
    chk(hipMalloc(&c->b.bar, 4 * sizeof(unsigned)));
    chk(hipHostMalloc(&c->h_ctl, 2 * sizeof(OvCtl)));
    chk(hipHostMalloc(&c->h_z, 2 * sizeof(double)));
    chk(hipHostMalloc(&c->h_flags, 4 * sizeof(unsigned)));
    t->ov = c;
    if (err != hipSuccess) {
        set_error("overlapped-pivot scratch allocation failed: %s", hipGetErrorString(err));
        ov_release(t);
        return err == hipErrorOutOfMemory ? LPR_OUT_OF_MEMORY : LPR_DEVICE_ERROR;
    }
    hipStream_t s = t->eng->stream;
    if (c->T2) LPR_HIP(hipMemsetAsync(c->T2, 0, tbytes, s));
    LPR_HIP(hipMemsetAsync(c->b.prow, 0, (size_t)2 * kOvMax * c->ld * D, s));
    LPR_HIP(hipMemsetAsync(c->b.fcol, 0, (size_t)2 * kOvMax * c->Rp * D, s));
    LPR_HIP(hipMemsetAsync(c->b.zrow, 0, (size_t)c->ld * D, s));
    LPR_HIP(hipMemsetAsync(c->b.bvec, 0, (size_t)2 * c->Rp * D, s));
    LPR_HIP(hipMemsetAsync(c->b.zparts, 0, (size_t)2 * kOvGroups * sizeof(ZPart), s));
    LPR_HIP(hipMemsetAsync(c->b.dbg, 0, kOvDbgWords * sizeof(unsigned long long), s));
    LPR_HIP(hipMemsetAsync(c->b.sflag, 0, kOvFlagWords * sizeof(unsigned), s));
    std::memset(c->h_ctl, 0, 2 * sizeof(OvCtl));
    std::memset(c->h_flags, 0, 4 * sizeof(unsigned));
    return LPR_OK_OPTIMAL;
}

// sweep tile code (low byte of opts.variant): rows per chunk, + 0x20 = two chunks in flight
static int ov_tile_code(int tr) {
    switch (tr) {
        case 0x04: case 0x10: case 0x24: case 0x28: return tr;
        default: return 0x08;
    }
}

// One XCD has 32 CUs and a head workgroup needs a CU of its own (its lanes keep their slices of the
// block in registers: one wave per SIMD): more than 32 groups cannot be resident on one XCD
// together, and a group that is not resident never answers a hand-off.  Wider tableaux spread.
static int ov_spread(int G, int flags) { return ((flags & 2) || G > 32) ? 1 : 8; }

// persistent sweep: enough workgroups to fill every CU at the kernel's occupancy (4 per CU)
static int ov_sweep_grid(const lpr_tableau* t, int ntiles) {
    const int cap = 4 * (t->eng->num_cus > 0 ? t->eng->num_cus : 256);
    return ntiles < cap ? (ntiles > 0 ? ntiles : 1) : cap;
}

static int ov_groups(const lpr_tableau* t) {
    int g = (t->ld / 2 + kOvNT - 1) / kOvNT;
    if (g < 1) g = 1;
    if (g > kOvGroups) g = kOvGroups;
    return g;
}

// start of a solve call: control block 0, barrier counters, Z row / RHS column / entering column
int ov_begin(lpr_tableau* t, int64_t iter, int64_t max_iter) {
    lpr_overlap_ctx* c = static_cast<lpr_overlap_ctx*>(t->ov);
    hipStream_t s = t->eng->stream;
    c->b.Tb[0] = t->T;
    c->b.Tb[1] = c->T2;
    c->b.basis = t->basis;
    c->b.log = t->log;
    OvCtl* h = c->h_ctl;
    std::memset(h, 0, 2 * sizeof(OvCtl));
    h[0].status = kRunning;
    h[0].pending = kRunning;
    h[0].staged = iter;
    h[0].applied = iter;
    h[0].max_iter = max_iter;
    h[0].log_cap = t->log_cap;
    h[0].head_xcc = -1;
    h[1] = h[0];
    LPR_HIP(hipMemcpyAsync(c->b.ctl, h, 2 * sizeof(OvCtl), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_ov_prologue, dim3(1), dim3(1024), 0, s, t->T, t->ld, t->rows, t->cols,
                       c->b.zrow, c->b.bvec + (size_t)(iter & 1) * c->Rp,
                       c->b.zparts + (iter & 1) * kOvGroups, ov_groups(t),
                       c->b.gran + (size_t)(iter & 1) * 3 * kOvParts, (unsigned)(2 * iter + 1),
                       c->b.gran, c->b.xgran, c->b.bar, c->b.tileq, c->b.hx, c->b.sflag);
    LPR_HIP(hipGetLastError());
    return LPR_OK_OPTIMAL;
}

// the log buffer was re-allocated: new pointer / capacity for the following launches
int ov_set_log(lpr_tableau* t, int parity) {
    lpr_overlap_ctx* c = static_cast<lpr_overlap_ctx*>(t->ov);
    c->b.log = t->log;
    c->h_ctl[0].log_cap = t->log_cap;
    LPR_HIP(hipMemcpyAsync(&c->b.ctl[parity].log_cap, &c->h_ctl[0].log_cap, sizeof(int64_t),
                           hipMemcpyHostToDevice, t->eng->stream));
    // the loop heads read this word from their OWN stream, which only waits for the engine
    // stream's previous sweep: without this wait the first heads of the next batch could still
    // see the old capacity and drop log entries (rare: the log doubles a handful of times a solve)
    LPR_HIP(hipStreamSynchronize(t->eng->stream));
    return LPR_OK_OPTIMAL;
}

void ov_launch_step(lpr_tableau* t, int K, int tr, int lp) {
    lpr_overlap_ctx* c = static_cast<lpr_overlap_ctx*>(t->ov);
    hipStream_t s = t->eng->stream;
    const int G = ov_groups(t);
    const int ld2 = t->ld / 2;
    const int nct = (ld2 + kOvNT - 1) / kOvNT;
    const int nrt = (t->rows + kOvTileRows - 1) / kOvTileRows;
    const dim3 grid(G + nct * nrt), blk(kOvNT);
#define LPR_OV_STEP(TR, DB)                                                                      \
    hipLaunchKernelGGL((k_ov_step<TR, DB>), grid, blk, 0, s, c->b, c->b.fcol, c->b.prow, t->ld,   \
                       t->rows, t->cols, c->Rp, K, G, lp)
    if (tr >= 16) LPR_OV_STEP(16, false);
    else if (tr >= 8) LPR_OV_STEP(8, false);
    else LPR_OV_STEP(4, false);
#undef LPR_OV_STEP
}

// flags (opts.variant >> 16): 1 = diagnostic time stamps, 2 = do not confine the heads to one XCD
// (one workgroup per group, memory-side hand-offs: the round-1 form), 4 = confine them but keep
// the memory-side hand-offs.
void ov_launch_heads(lpr_tableau* t, int K, int flags) {
    lpr_overlap_ctx* c = static_cast<lpr_overlap_ctx*>(t->ov);
    const int G = ov_groups(t);  // 256 lanes per group; 512 measures the same, 1024 slower
    const int spread = ov_spread(G, flags);
    const int no_l2 = (flags & 6) ? 1 : 0;
    hipStream_t s = t->eng->stream;
    if (flags & 1)
        hipLaunchKernelGGL((k_ov_heads<kOvNT, true>), dim3(G * spread), dim3(kOvNT), 0, s, c->b,
                           t->ld, t->rows, t->cols, c->Rp, K, G, spread, no_l2);
    else
        hipLaunchKernelGGL((k_ov_heads<kOvNT, false>), dim3(G * spread), dim3(kOvNT), 0, s, c->b,
                           t->ld, t->rows, t->cols, c->Rp, K, G, spread, no_l2);
}

void ov_launch_sweep(lpr_tableau* t, int tr) {
    lpr_overlap_ctx* c = static_cast<lpr_overlap_ctx*>(t->ov);
    hipStream_t s = t->eng->stream;
    const int nct = (t->ld / 2 + kOvNT - 1) / kOvNT;
    const int nrt = (t->rows + kOvTileRows - 1) / kOvTileRows;
    const int cus = t->eng->num_cus > 0 ? t->eng->num_cus : 256;
    if (nct * nrt < 2 * cus && ov_tile_code(tr) == 0x08) {
        // a small tableau: 32-row tiles would give fewer workgroups than CUs and the sweep is all
        // latency (m = 512: 68 tiles, 30 us for 12.6 MB); 8-row tiles spread it over the chip
        const int nrt8 = (t->rows + 7) / 8;
        hipLaunchKernelGGL((k_ov_sweep<8, false, 8>), dim3(nct * nrt8), dim3(kOvNT), 0, s, c->b,
                           c->b.fcol, c->b.prow, t->ld, t->rows, c->Rp);
        return;
    }
    const dim3 grid(nct * nrt), blk(kOvNT);
#define LPR_OV_SWEEP(TR, DB)                                                                     \
    hipLaunchKernelGGL((k_ov_sweep<TR, DB>), grid, blk, 0, s, c->b, c->b.fcol, c->b.prow, t->ld,  \
                       t->rows, c->Rp)
    switch (ov_tile_code(tr)) {
        case 0x04: LPR_OV_SWEEP(4, false); break;
        case 0x10: LPR_OV_SWEEP(16, false); break;
        case 0x24: LPR_OV_SWEEP(4, true); break;
        case 0x28: LPR_OV_SWEEP(8, true); break;
        default: LPR_OV_SWEEP(8, false); break;
    }
#undef LPR_OV_SWEEP
}

// two-stream variant: heads on their own (high-priority) stream, sweep on the engine stream.
// Step k's kernels both start when both kernels of step k-1 are done.
int ov2_begin(lpr_tableau* t) {
    lpr_overlap_ctx* c = static_cast<lpr_overlap_ctx*>(t->ov);
    if (!c->hstream) {
        int lo = 0, hi = 0;
        LPR_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        LPR_HIP(hipStreamCreateWithPriority(&c->hstream, hipStreamNonBlocking, hi));
        for (int k = 0; k < 2; ++k) {
            LPR_HIP(hipEventCreateWithFlags(&c->ev_h[k], hipEventDisableTiming));
            LPR_HIP(hipEventCreateWithFlags(&c->ev_s[k], hipEventDisableTiming));
        }
    }
    // everything queued on the engine stream so far (prologue, control block) precedes step 0
    c->ev_idx = 0;
    c->steps = 0;
    c->batch_first = true;
    LPR_HIP(hipEventRecord(c->ev_s[1], t->eng->stream));
    LPR_HIP(hipEventRecord(c->ev_h[1], c->hstream));
    c->last_sweep = c->ev_s[1];
    LPR_HIP(hipStreamWaitEvent(c->hstream, c->ev_s[1], 0));
    return LPR_OK_OPTIMAL;
}

int ov2_launch_step(lpr_tableau* t, int K, int tr, int lp, int flags, hipEvent_t ev_start,
                    hipEvent_t ev_stop) {
    lpr_overlap_ctx* c = static_cast<lpr_overlap_ctx*>(t->ov);
    hipStream_t S = t->eng->stream, H = c->hstream;
    const int cur = c->ev_idx, prev = cur ^ 1;
    // default: the heads wait for the previous sweep by a cross-stream event.  flags 32: they
    // follow their predecessor at once and wait on the device (B.sflag) -- 3 % faster, but it
    // NEEDS the two kernels of a step to run concurrently: under a tool that serialises kernels
    // (rocprofv3 --pmc does) the heads would wait for a sweep that cannot start, until the bounded
    // wait gives up with LPR_DEVICE_ERROR.  A solver must not depend on concurrency for progress,
    // so it is opt-in.
    const bool by_event = (flags & 32) == 0;
    const int wait_sweeps = by_event ? -1 : c->steps;
    // flags 64: the SWEEP does not wait for the heads of the step before by an event either: it
    // follows its predecessor on the engine stream at once and asks the heads' completion word as
    // it starts (ov_tiles).  Not for the first step after the streams were joined, nor for the
    // first two steps of a call (the heads it would wait for may not be resident yet: they
    // could find the chip full of polling sweep workgroups), nor while the heads' XCD is unknown.
    const bool sweep_dev = (flags & 64) != 0 && !c->batch_first && c->steps >= 2 &&
                           c->head_xcc_hint >= 0;
    const int heads_done = (flags & 64) ? c->steps + 1 : -1;
    // The events that order the two streams (and the ones that time a sampled step) are the
    // completion signals of the kernels themselves (hipExtLaunchKernelGGL's stop event): a separate
    // hipEventRecord is a packet of its own on the stream, ~3 us each on the critical cycle.
    if (by_event) LPR_HIP(hipStreamWaitEvent(H, c->last_sweep, 0));
    if (!sweep_dev) LPR_HIP(hipStreamWaitEvent(S, c->ev_h[prev], 0));
    const int G = ov_groups(t);
    const int spread = ov_spread(G, flags);
    const int no_l2 = (flags & 6) ? 1 : 0;
    if (flags & 1)
        hipExtLaunchKernelGGL((k_ov2_heads<kOvNT, true>), dim3(G * spread), dim3(kOvNT), 0, H,
                              nullptr, c->ev_h[cur], 0, c->b, t->ld, t->rows, t->cols, c->Rp, K, G,
                              lp, spread, no_l2, wait_sweeps, heads_done);
    else
        hipExtLaunchKernelGGL((k_ov2_heads<kOvNT, false>), dim3(G * spread), dim3(kOvNT), 0, H,
                              nullptr, c->ev_h[cur], 0, c->b, t->ld, t->rows, t->cols, c->Rp, K, G,
                              lp, spread, no_l2, wait_sweeps, heads_done);
    const int nct = (t->ld / 2 + kOvNT - 1) / kOvNT;
    const int nrt = (t->rows + kOvTileRows - 1) / kOvTileRows;
    const dim3 grid(ov_sweep_grid(t, nct * nrt)), blk(kOvNT);
    // leave the heads' XCD to the heads: 2 = by the previous launch's hint and this launch's
    // word, 1 = by this launch's word only, 0 = never (flags 8 / 16; nothing to leave when the
    // heads are not confined to one XCD)
    const int avoid = (flags & (2 | 4 | 8)) ? 0 : ((flags & 16) ? 1 : 2);
    // a sampled step: its stop event is the timing event (start = the kernel's own start)
    hipEvent_t sweep_done = ev_stop ? ev_stop : c->ev_s[cur];
#define LPR_OV2_SWEEP(TR, DB)                                                                     \
    hipExtLaunchKernelGGL((k_ov2_sweep<TR, DB>), grid, blk, 0, S, ev_start, sweep_done, 0, c->b,   \
                          c->b.fcol, c->b.prow, t->ld, t->rows, c->Rp, lp, avoid, by_event ? 1 : 0, \
                          by_event ? -1 : c->steps, sweep_dev ? c->steps : -1, c->head_xcc_hint)
    switch (ov_tile_code(tr)) {
        case 0x04: LPR_OV2_SWEEP(4, false); break;
        case 0x10: LPR_OV2_SWEEP(16, false); break;
        case 0x24: LPR_OV2_SWEEP(4, true); break;
        case 0x28: LPR_OV2_SWEEP(8, true); break;
        default: LPR_OV2_SWEEP(8, false); break;
    }
#undef LPR_OV2_SWEEP
    c->last_sweep = sweep_done;
    c->steps += 1;
    c->ev_idx = prev;
    c->batch_first = false;
    return LPR_OK_OPTIMAL;
}

// the engine stream catches up with the heads' stream (before a poll / the end of the call)
int ov2_join(lpr_tableau* t) {
    lpr_overlap_ctx* c = static_cast<lpr_overlap_ctx*>(t->ov);
    LPR_HIP(hipStreamWaitEvent(t->eng->stream, c->ev_h[c->ev_idx ^ 1], 0));
    c->batch_first = true;
    return LPR_OK_OPTIMAL;
}

// reads control block `parity` back, and entry 0 of both RHS-column buffers (= T[0, cols-1] after
// an even / odd number of pivots): one synchronisation per poll
int ov_poll(lpr_tableau* t, int parity, OvPoll* out) {
    lpr_overlap_ctx* c = static_cast<lpr_overlap_ctx*>(t->ov);
    hipStream_t s = t->eng->stream;
    LPR_HIP(hipMemcpyAsync(c->h_ctl, c->b.ctl, 2 * sizeof(OvCtl), hipMemcpyDeviceToHost, s));
    LPR_HIP(hipMemcpyAsync(&c->h_z[0], c->b.bvec, sizeof(double), hipMemcpyDeviceToHost, s));
    LPR_HIP(hipMemcpyAsync(&c->h_z[1], c->b.bvec + c->Rp, sizeof(double), hipMemcpyDeviceToHost,
                           s));
    LPR_HIP(hipMemcpyAsync(c->h_flags, c->b.sflag, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    LPR_HIP(hipStreamSynchronize(s));
    const OvCtl& h = c->h_ctl[parity];
    c->head_xcc_hint = h.head_xcc;
    out->status = h.status;
    out->pending = h.pending;
    out->kdone = h.kdone;
    out->cur = h.cur;
    out->error = h.error | (c->h_flags[3] ? 1 : 0);  // (a sweep gave up waiting for its heads)
    out->applied = h.applied;
    out->z[0] = c->h_z[0];
    out->z[1] = c->h_z[1];
    return LPR_OK_OPTIMAL;
}

// diagnostic: the stamps of the lead head workgroup (ring of kOvStampPivots pivots x
// kOvStampsPerPivot, 10 ns ticks), then its XCC id and the hand-off mode of its last launch
int ov_read_stamps(lpr_tableau* t, uint64_t* out, int64_t cap, int64_t* count) {
    lpr_overlap_ctx* c = static_cast<lpr_overlap_ctx*>(t->ov);
    if (!c) {
        *count = 0;
        return LPR_OK_OPTIMAL;
    }
    int64_t n = (int64_t)kOvDbgWords;
    if (n > cap) n = cap;
    *count = n;
    if (n > 0 && out) {
        LPR_HIP(hipStreamSynchronize(t->eng->stream));
        LPR_HIP(hipMemcpy(out, c->b.dbg, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return LPR_OK_OPTIMAL;
}

// end of a solve call: the live tableau must be t->T
void ov_adopt_buffer(lpr_tableau* t, int cur) {
    lpr_overlap_ctx* c = static_cast<lpr_overlap_ctx*>(t->ov);
    if (cur == 1) {  // the result is in the second buffer: swap ownership
        double* tmp = t->T;
        t->T = c->T2;
        c->T2 = tmp;
    }
}

}  // namespace lpr
#endif  // LPR_OV_KERNELS_ONLY
