// select_common.hpp -- device helpers shared by the pivot-selection kernels of the primal path
// (primal_kernels.hip, block_kernels.hip).  Not part of the ABI.
#pragma once

#include "engine_common.hpp"

#include <climits>

#pragma clang fp contract(off)

namespace lpr {

// ------------------------------------------------------------------------------------------
// (value, index) lexicographic arg-min.  i < 0 == "no candidate".  Both C# scans keep a candidate
// only when it is STRICTLY below the running best, so the sequential result is the minimum value
// at its lowest index -- which is exactly the lexicographic minimum, and that is associative and
// commutative, so a tree reduction gives the same answer as the C# loop.
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

__device__ __forceinline__ Cand wave_cand_min(Cand c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Cand o;
        o.v = __shfl_xor(c.v, off, kWave);
        o.i = __shfl_xor(c.i, off, kWave);
        c = cand_min(c, o);
    }
    return c;
}

// All threads of the block receive the block-wide minimum.  lds_v/lds_i hold one slot per wave.
__device__ __forceinline__ Cand block_cand_min(Cand c, double* lds_v, int* lds_i) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int nwaves = blockDim.x / kWave;
    c = wave_cand_min(c);
    __syncthreads();  // protect the slots against the previous use
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

// ------------------------------------------------------------------------------------------
// The same lexicographic minimum with DPP moves (row-local permutes, then the two row broadcasts of
// gfx9) instead of six rounds of ds_bpermute, as two plain reductions: the minimum VALUE over the
// candidates, then the minimum INDEX over the lanes that hold it.  Candidates must not be NaN (a
// ratio that passed `>= 0`, a Z-row entry that passed `<`); ties go to the lower index as in
// cand_min; the value returned when there is no candidate is unused by the callers.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fmin(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return fmin(v, __hiloint2double(ohi, olo));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_imin(int i) {
    return min(i, __builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ Cand dpp_wave_cand_min(Cand c) {
    double v = (c.i >= 0) ? c.v : INFINITY;
    v = dpp_fmin<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = dpp_fmin<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = dpp_fmin<0x141, 0xf>(v);  // row_half_mirror
    v = dpp_fmin<0x140, 0xf>(v);  // row_mirror: every row of 16 holds its minimum
    v = dpp_fmin<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
    v = dpp_fmin<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3: lane 63 holds the minimum
    const int vlo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int vhi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    const double vmin = __hiloint2double(vhi, vlo);
    int i = (c.i >= 0 && c.v == vmin) ? c.i : INT_MAX;
    i = dpp_imin<0xB1, 0xf>(i);
    i = dpp_imin<0x4E, 0xf>(i);
    i = dpp_imin<0x141, 0xf>(i);
    i = dpp_imin<0x140, 0xf>(i);
    i = dpp_imin<0x142, 0xa>(i);
    i = dpp_imin<0x143, 0xc>(i);
    const int imin = __builtin_amdgcn_readlane(i, 63);
    Cand r;
    r.v = vmin;
    r.i = (imin == INT_MAX) ? -1 : imin;
    return r;
}

// Block-wide form: a DPP reduction per wave, one slot per wave in LDS, then every wave reduces the
// slots itself (one per lane) -- two barriers, no serial walk over the slots.
__device__ __forceinline__ Cand dpp_block_cand_min(Cand c, double* lds_v, int* lds_i) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int nwaves = blockDim.x / kWave;
    c = dpp_wave_cand_min(c);
    __syncthreads();  // protect the slots against the previous use
    if (lane == 0) {
        lds_v[wave] = c.v;
        lds_i[wave] = c.i;
    }
    __syncthreads();
    Cand o;
    o.v = (lane < nwaves) ? lds_v[lane] : 0.0;
    o.i = (lane < nwaves) ? lds_i[lane] : -1;
    return dpp_wave_cand_min(o);
}

// The same for workgroups of at most 16 waves, with ONE barrier: the slots alternate between two
// banks (`bank` = parity of the call count, the caller's), so a wave that is already publishing
// for the next reduction cannot overwrite a slot another wave still reads; and the second stage
// reduces 16 lanes, not 64 (four row-local DPP stages instead of six per pass).
// lds_v / lds_i: 32 entries each.
__device__ __forceinline__ Cand dpp_block_cand_min16(Cand c, double* lds_v, int* lds_i, int bank) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int nwaves = blockDim.x / kWave;
    c = dpp_wave_cand_min(c);
    if (lane == 0) {
        lds_v[bank * 16 + wave] = c.v;
        lds_i[bank * 16 + wave] = c.i;
    }
    __syncthreads();
    const int l = lane & 15;
    const bool ok = l < nwaves;
    const double ov = lds_v[bank * 16 + (ok ? l : 0)];
    const int oi = ok ? lds_i[bank * 16 + l] : -1;
    double v = (oi >= 0) ? ov : INFINITY;
    v = dpp_fmin<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v = dpp_fmin<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v = dpp_fmin<0x141, 0xf>(v);  // row_half_mirror
    v = dpp_fmin<0x140, 0xf>(v);  // row_mirror: every lane of the row holds the minimum
    int i = (oi >= 0 && ov == v) ? oi : INT_MAX;
    i = dpp_imin<0xB1, 0xf>(i);
    i = dpp_imin<0x4E, 0xf>(i);
    i = dpp_imin<0x141, 0xf>(i);
    i = dpp_imin<0x140, 0xf>(i);
    Cand r;
    r.v = v;
    r.i = (i == INT_MAX) ? -1 : i;
    // (every row of 16 lanes did the same reduction: the result is wave-uniform; say so)
    r.i = __builtin_amdgcn_readfirstlane(r.i);
    return r;
}

// ------------------------------------------------------------------------------------------
// Partial results of the next-entering-column arg-min, one per k_pivot_head workgroup.  Two banks,
// selected by the parity of the pivot counter: head t reads bank (iter & 1) -- written by head t-1
// or by k_bootstrap -- and writes bank ((iter + 1) & 1), so a fast workgroup can never overwrite a
// partial that a slow one has not read yet.
struct ZPart {
    double v;
    int32_t i;
    int32_t pad;
};

__device__ __forceinline__ Cand reduce_zparts(const ZPart* __restrict__ bank, int G) {
    Cand c;
    c.v = 0.0;
    c.i = -1;
    for (int p = 0; p < G; ++p) {  // G <= kMaxHeadGroups uniform loads, a line or two of L2
        Cand o;
        o.v = bank[p].v;
        o.i = bank[p].i;
        c = cand_min(c, o);
    }
    return c;
}

}  // namespace lpr
