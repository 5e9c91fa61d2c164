// revised_common.hpp -- device/host state of the revised primal simplex solver (internal).
#pragma once

#include "engine_common.hpp"

namespace lpr {

// Control block in device memory, polled by the host once per batch of iterations.
struct RevState {
    int32_t status;       // kRunning or an lpr_status
    int32_t entering;     // entering variable chosen by k_rev_enter (-1: optimal)
    int32_t leaving_row;  // basis row chosen by k_rev_ratio
    int32_t pad;
    int64_t iter;         // completed pivots
    int64_t max_iter;     // stop when iter reaches this (<= 0: none)
    int64_t log_cap;      // capacity of the (row, enter, leave) log in triples
    // arrival counters of the fused kernels (revised_fused.hip): the workgroup whose add comes
    // last runs the selection on what the others have stored, then puts the counter back to 0
    int32_t arrive_rc;    // k_rev_rc_enter
    int32_t arrive_xu;    // k_rev_xu_ratio
};

}  // namespace lpr

// RevisedPrimalSimplexSolver's fields (RevisedPrimalSimplexSolver.cs:15-33) in HBM.  `B` (:30) is
// maintained by the C# but never read; it is not kept.
struct lpr_revised {
    lpr_engine* eng = nullptr;
    int n = 0, m = 0;
    int lda = 0, ldb = 0;       // leading dimensions of A (m x n) and B^-1 (m x m), 16-double padded
    int is_min = 0;
    double* A = nullptr;        // m x lda
    double* At = nullptr;       // n x ldb: A transposed (column j of A contiguous: GetColumn(A, j) :390-396
                                // is ONE 8 m-byte read instead of m DRAM pages; 8 m n more bytes of HBM)
    double* Binv = nullptr;     // m x ldb
    double* b = nullptr;        // m
    double* c = nullptr;        // n   (= -cOrig for min, :51)
    double* cOrig = nullptr;    // n
    double* cB = nullptr;       // m
    double* xB = nullptr;       // m
    double* y = nullptr;        // m
    double* rcx = nullptr;      // n
    double* wmin = nullptr;     // ceil(n/32) + ceil(m/16): per producing workgroup, the minimum of
                                // -rc over its entering candidates (+inf: none), structural groups
                                // first (k_rev_rc_enter), then slack groups (k_rev_update_y)
    double* acol = nullptr;     // m   GetColumn(A, e)
    double* u = nullptr;        // m   direction
    double* fac = nullptr;      // m   column r of E (:272)
    double* browbuf = nullptr;  // ldb old pivot row of B^-1
    double* x = nullptr;        // n   SolutionVector
    double* z = nullptr;        // 1   finalZ
    int32_t* basic = nullptr;   // m   basicVariables (by row)
    uint8_t* is_basic = nullptr;  // n + m: complement of nonBasicVariables
    int32_t* log = nullptr;     // 3 * log_cap
    int64_t log_cap = 0;
    lpr::RevState* state = nullptr;
    lpr::RevState* h_state = nullptr;  // pinned
    double* gemm_out = nullptr;  // m x ldc scratch of lpr_revised_binv_a (lazy)
    int ldc = 0;
    // what CaptureSnapshot (:294-387) prints besides y / rc / xB / B^-1 (lazy, lpr_revised_step)
    double* snap_ratios = nullptr;   // m   ratios_pre (:161,:174), +inf where u_i <= EPS
    int32_t* snap_basis = nullptr;   // m   basisForRatios_Pre (:186)
    double* snap_scal = nullptr;     // [0] enteringRC_pre (:189-191), [1] zWorking = Dot(cB, xB)
    double* h_snap_scal = nullptr;   // pinned, 4 doubles: + [2] zOriginal
    unsigned long long* dbg_stamps = nullptr;  // LPR_REV_STAMPS=1 (diagnostic), 8 words
    int64_t total_iter = 0;
    int last_status = 0;
};
