// select_common.hpp -- device helpers shared by the pivot-selection kernels of the primal path
// (primal_kernels.hip, block_kernels.hip).  Not part of the ABI.
#pragma once

#include "engine_common.hpp"

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
