// engine_common.hpp -- internal definitions shared by the gfx950 engine sources.
// Not part of the ABI (include/lpr_engine.h is).
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lpr_engine.h"

// The C# rounds every product before it is added/subtracted (e.g. PrimalSimplexSolver.cs:208).
// hipcc contracts a*b+c into v_fma_f64 by default; that is one rounding fewer and changes bits,
// so contraction is off for every translation unit of the engine (also -ffp-contract=off in the
// Makefile; tests/test_build.py greps the ISA of the update kernels for stray FMAs).
#pragma clang fp contract(off)

namespace lpr {

constexpr int kWave = 64;
constexpr int kLdAlign = 16;          // tableau rows padded to 16 doubles = 128 B
constexpr int32_t kRunning = -100;    // device status word while the pivot loop is live
constexpr int kMaxHeadGroups = 32;    // k_pivot_head workgroups (partials per bank)

void set_error(const char* fmt, ...);
const char* get_error();

#define LPR_HIP(expr)                                                                     \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::lpr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                             __FILE__, __LINE__);                                         \
            return LPR_DEVICE_ERROR;                                                      \
        }                                                                                 \
    } while (0)

inline int align_up(int x, int a) { return (x + a - 1) / a * a; }

// a / b, correctly rounded.  The device's own fp64 division (v_div_scale / v_rcp / Newton steps /
// v_div_fmas / v_div_fixup) ends in ONE fused multiply-add of the residual with an APPROXIMATE
// reciprocal; when a / b lies within ~1e-16 ulp of the midpoint between two doubles that step can
// round the other way than IEEE 754 division (what the C#'s `/` is on x64).  Found by
// tools/fuzz_side_gpu.py: -0x1.6666666666663p+0 / -0x1.ffffffffffffbp+1 is 0.35000000000000003,
// the device said 0.35 -- a 1-ulp pivot-row entry that then spread through a column
// (tools/div_probe.hip counts such pairs).  The repair is exact: for a quotient q within one ulp
// of a / b the residual a - q b is representable, so fma(-q, b, a) IS that residual; the same for
// the neighbour of q on the side the residual points to; the smaller residual wins (a division
// cannot produce an exact tie).  Outside the comfortable exponent range (zero, subnormal, huge,
// inf, NaN) the hardware result is kept: the residuals would not be exact there.
#if defined(__HIPCC__)
// does the hardware quotient q of a / b need a second look?  (no: the residual is clearly -- by
// 2^-20 of it -- below half an ulp of q times |b|.  Five instructions; one lane in a million says
// yes; so do infinite and NaN quotients and those near the ends of the exponent range, where the
// bound is meaningless: ieee_div_repair hands those back as they are.)
__device__ __forceinline__ bool ieee_div_suspect(double a, double b, double q) {
    const double r = __builtin_fma(-q, b, a);
    // |b| * 2^(exponent of q - 54) * (1 - 2^-20): half an ulp of q times |b|, less a margin
    const double h = __builtin_ldexp(b * 0x1.ffffep-1, __builtin_amdgcn_frexp_exp(q) - 54);
    return !(__builtin_fabs(r) < __builtin_fabs(h));
}
__device__ __forceinline__ double ieee_div_repair(double a, double b, double q) {
    const double aq = __builtin_fabs(q), aa = __builtin_fabs(a), ab = __builtin_fabs(b);
    if (!(aq > 0x1p-900 && aq < 0x1p900 && aa > 0x1p-900 && aa < 0x1p900 && ab > 0x1p-900 &&
          ab < 0x1p900))
        return q;
    const double r = __builtin_fma(-q, b, a);
    if (r == 0.0) return q;
    const long long qb = __builtin_bit_cast(long long, q);
    const bool up = (r > 0.0) == (b > 0.0);  // a / b - q has the sign of r / b
    const bool away = (q > 0.0) == up;       // the neighbour on that side is the larger magnitude
    const double q2 = __builtin_bit_cast(double, qb + (away ? 1ll : -1ll));
    const double r2 = __builtin_fma(-q2, b, a);
    const double ar = __builtin_fabs(r), ar2 = __builtin_fabs(r2);
    if (ar2 < ar || (ar2 == ar && (qb & 1ll))) return q2;
    return q;
}
__device__ __forceinline__ double ieee_div(double a, double b) {
    const double q = a / b;
    if (__builtin_expect(ieee_div_suspect(a, b, q), 0)) return ieee_div_repair(a, b, q);
    return q;
}
// N quotients behind ONE branch (the divisions and their checks interleave; a latency-bound loop
// head pays for the branch, not for the seven instructions)
template <int N>
__device__ __forceinline__ void ieee_div_n(const double (&a)[N], const double (&b)[N],
                                           double (&q)[N]) {
    bool any = false;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        q[k] = a[k] / b[k];
        any = any || ieee_div_suspect(a[k], b[k], q[k]);
    }
    if (__builtin_expect(any, 0)) {
#pragma unroll
        for (int k = 0; k < N; ++k) q[k] = ieee_div_repair(a[k], b[k], q[k]);
    }
}
#endif

// Per-tableau control block in device memory.  Written by the select kernel, read by the update
// kernel and polled by the host once per batch of pivots (never once per pivot).
struct PivotState {
    int32_t status;     // kRunning or an lpr_status
    int32_t cur_r;      // pivot row of the pending update (tableau row, 1-based constraint index)
    int32_t cur_e;      // pivot column of the pending update
    int32_t sweep;      // parity of the update sweep direction (serpentine traversal)
    int64_t iter;       // pivots performed so far
    int64_t max_iter;   // stop when iter reaches this (<= 0: no limit)
    int64_t log_cap;    // capacity of the pivot log in pairs
    int64_t iter_pending;  // iter + 1 once k_pivot_head has chosen a pivot; k_update commits it
};

}  // namespace lpr

struct lpr_tableau;
struct lpr_revised;
struct lpr_bb;
struct lpr_sens;
struct lpr_comm;

struct lpr_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 0;
    char arch[64] = {0};
    // tableaux created on this engine and not yet destroyed; lpr_engine_close releases their
    // device memory and orphans them so that a late lpr_tableau_destroy stays safe
    std::vector<lpr_tableau*> live;
    std::vector<lpr_revised*> live_rev;
    std::vector<lpr_bb*> live_bb;
    std::vector<lpr_sens*> live_sens;
    std::vector<lpr_comm*> live_comm;  // RCCL communicators whose collectives run on this stream
};

struct lpr_tableau {
    lpr_engine* eng = nullptr;
    int rows = 0, cols = 0, ld = 0;  // ld = cols rounded up to kLdAlign
    double* T = nullptr;             // rows x ld, row-major, padding columns kept at 0
    double* T2 = nullptr;            // second buffer of the fused small-tableau path (lazy)
    double* rowbuf = nullptr;        // ld doubles: normalised pivot row
    double* colbuf = nullptr;        // rows doubles: pivot column before the update
    double* next_col = nullptr;      // rows doubles: column next_e of the tableau AFTER the update
    double* next_rhs = nullptr;      // rows doubles: RHS column of the tableau AFTER the update
    void* zparts = nullptr;          // 2 banks x kMaxHeadGroups partial arg-mins of the next Z row
    int32_t* basis = nullptr;        // rows-1
    int32_t* log = nullptr;          // 2*log_cap: (row, col) pairs
    int64_t log_cap = 0;
    lpr::PivotState* state = nullptr;     // device
    lpr::PivotState* h_state = nullptr;   // pinned host mirror
    int32_t* scratch_i = nullptr;    // small device scratch for single-step results
    int32_t* h_scratch_i = nullptr;  // pinned
    double* xbuf = nullptr;          // extract-solution output (lazy)
    int xbuf_n = 0;
    int64_t total_pivots = 0;
    // kernel timing (opts.time_kernels)
    std::vector<hipEvent_t> ev;      // pairs (start, stop)
    int64_t timed_launches = 0;
    double timed_total_ms = 0.0;
    int64_t timed_steps = 0;         // overlapped paths: steps (sweep || next heads) timed
    double timed_step_ms = 0.0;
    bool poisoned = false;           // a device-side hand-off timed out: staging state unusable
    // captured batch of (select, update) pairs
    hipGraphExec_t graph = nullptr;
    int graph_batch = 0;
    int graph_variant = -1;
    // every kernel argument a capture bakes in: a graph is replayed only while all of them still
    // hold (the fused / overlapped paths swap T and T2, the cut path grows rows, the log and the
    // block scratch are re-allocated on demand)
    struct GraphKey {
        const void *T = nullptr, *T2 = nullptr, *log = nullptr, *basis = nullptr, *blk = nullptr;
        const void *next_col = nullptr, *colbuf = nullptr;
        int rows = 0, cols = 0, ld = 0;
        bool operator==(const GraphKey& o) const {
            return T == o.T && T2 == o.T2 && log == o.log && basis == o.basis && blk == o.blk &&
                   next_col == o.next_col && colbuf == o.colbuf && rows == o.rows &&
                   cols == o.cols && ld == o.ld;
        }
    } graph_key;
    void* ov = nullptr;               // lpr_overlap_ctx of the overlapped K-pivot path (overlap_kernels.hip)
    void* blk = nullptr;              // lpr_block_ctx of the K-pivots-per-sweep path (block_kernels.hip)
    void* cut = nullptr;              // lpr_cut_ctx of the cutting-plane side path (cut_kernels.hip)
    void* small = nullptr;            // lpr_small_ctx of the cache-resident path (small_kernels.hip)
};
