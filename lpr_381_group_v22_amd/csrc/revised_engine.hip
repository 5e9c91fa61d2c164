// revised_engine.hip -- C ABI of the revised primal simplex (include/lpr_engine.h, lpr_revised_*).
// Host side of RevisedPrimalSimplexSolver.Solve() (Simplex/RevisedPrimalSimplexSolver.cs:82-251):
// queues batches of iterations, polls one device status word per batch.  No CPU fallback.
#include "engine_common.hpp"
#include "revised_common.hpp"

#include <cstdlib>
#include <new>

#pragma clang fp contract(off)

namespace lpr {

// revised_kernels.hip
void rev_launch_iteration(lpr_revised* s, bool snapshot);
void rev_launch_iteration_batched(lpr_revised* s);
void rev_launch_y(lpr_revised* s);
void rev_launch_prices(lpr_revised* s);
void rev_launch_zworking(lpr_revised* s);
void rev_launch_matmul_exact(lpr_revised* s, double* Cout, int ldc);
void rev_launch_extract(lpr_revised* s);
void rev_launch_init(lpr_revised* s);
void rev_launch_synthetic(lpr_revised* s, uint64_t seed);
void rev_launch_transpose_a(lpr_revised* s);
void rev_launch_gemm(lpr_revised* s, double* Cout, int ldc);

static void rev_release_device(lpr_revised* s) {
    hipSetDevice(s->eng->device);
    if (s->eng->stream) hipStreamSynchronize(s->eng->stream);
    hipFree(s->A); hipFree(s->At); hipFree(s->Binv); hipFree(s->b); hipFree(s->c); hipFree(s->cOrig);
    hipFree(s->cB); hipFree(s->xB); hipFree(s->y); hipFree(s->rcx); hipFree(s->acol);
    hipFree(s->wmin);
    s->wmin = nullptr;
    hipFree(s->u); hipFree(s->fac); hipFree(s->browbuf); hipFree(s->x); hipFree(s->z);
    hipFree(s->basic); hipFree(s->is_basic); hipFree(s->log); hipFree(s->state);
    hipFree(s->gemm_out);
    hipFree(s->dbg_stamps);
    s->dbg_stamps = nullptr;
    hipFree(s->snap_ratios); hipFree(s->snap_basis); hipFree(s->snap_scal);
    if (s->h_snap_scal) hipHostFree(s->h_snap_scal);
    s->snap_ratios = s->snap_scal = s->h_snap_scal = nullptr;
    s->snap_basis = nullptr;
    if (s->h_state) hipHostFree(s->h_state);
    s->At = nullptr;
    s->A = s->Binv = s->b = s->c = s->cOrig = s->cB = s->xB = s->y = s->rcx = s->acol = nullptr;
    s->u = s->fac = s->browbuf = s->x = s->z = s->gemm_out = nullptr;
    s->basic = s->log = nullptr;
    s->is_basic = nullptr;
    s->state = s->h_state = nullptr;
}

// called by lpr_engine_close for handles the caller has not destroyed
void rev_orphan(lpr_revised* s) {
    rev_release_device(s);
    s->eng = nullptr;
}

static int rev_alloc(lpr_engine* e, int n, int m, int is_min, lpr_revised** out) {
    if (!e || !out || n <= 0 || m <= 0) {
        // :43-44 ArgumentException("Objective/Constraints cannot be null or empty.")
        set_error("lpr_revised: objective and constraints cannot be empty (n=%d m=%d)", n, m);
        return LPR_BAD_ARGUMENT;
    }
    LPR_HIP(hipSetDevice(e->device));
    lpr_revised* s = new (std::nothrow) lpr_revised();
    if (!s) return LPR_OUT_OF_MEMORY;
    s->eng = e;
    s->n = n;
    s->m = m;
    s->is_min = is_min;
    s->lda = align_up(n, kLdAlign);
    s->ldb = align_up(m, kLdAlign);
    s->log_cap = 1 << 16;
    hipError_t err = hipSuccess;
    auto chk = [&](hipError_t x) { if (err == hipSuccess) err = x; };
    const size_t D = sizeof(double);
    chk(hipMalloc(&s->A, (size_t)m * s->lda * D));
    chk(hipMalloc(&s->At, (size_t)n * s->ldb * D));
    chk(hipMalloc(&s->Binv, (size_t)m * s->ldb * D));
    chk(hipMalloc(&s->b, (size_t)s->ldb * D));
    chk(hipMalloc(&s->c, (size_t)s->lda * D));
    chk(hipMalloc(&s->cOrig, (size_t)s->lda * D));
    chk(hipMalloc(&s->cB, (size_t)s->ldb * D));
    chk(hipMalloc(&s->xB, (size_t)s->ldb * D));
    chk(hipMalloc(&s->y, (size_t)s->ldb * D));
    chk(hipMalloc(&s->rcx, (size_t)s->lda * D));
    chk(hipMalloc(&s->wmin, (size_t)((n + 31) / 32 + (m + 15) / 16 + 1) * D));
    chk(hipMalloc(&s->acol, (size_t)s->ldb * D));
    chk(hipMalloc(&s->u, (size_t)s->ldb * D));
    chk(hipMalloc(&s->fac, (size_t)s->ldb * D));
    chk(hipMalloc(&s->browbuf, (size_t)s->ldb * D));
    chk(hipMalloc(&s->x, (size_t)s->lda * D));
    chk(hipMalloc(&s->z, D));
    chk(hipMalloc(&s->basic, (size_t)m * sizeof(int32_t)));
    chk(hipMalloc(&s->is_basic, (size_t)(n + m)));
    chk(hipMalloc(&s->log, (size_t)s->log_cap * 3 * sizeof(int32_t)));
    chk(hipMalloc(&s->state, sizeof(RevState)));
    chk(hipHostMalloc(&s->h_state, sizeof(RevState)));
    if (err != hipSuccess) {
        set_error("device allocation for a revised solver (m=%d n=%d) failed: %s", m, n,
                  hipGetErrorString(err));
        rev_release_device(s);
        delete s;
        return err == hipErrorOutOfMemory ? LPR_OUT_OF_MEMORY : LPR_DEVICE_ERROR;
    }
    hipStream_t st = e->stream;
    LPR_HIP(hipMemsetAsync(s->A, 0, (size_t)m * s->lda * D, st));
    LPR_HIP(hipMemsetAsync(s->Binv, 0, (size_t)m * s->ldb * D, st));
    LPR_HIP(hipMemsetAsync(s->cB, 0, (size_t)s->ldb * D, st));  // cB[i] = 0.0 :77
    LPR_HIP(hipMemsetAsync(s->xB, 0, (size_t)s->ldb * D, st));
    LPR_HIP(hipMemsetAsync(s->x, 0, (size_t)s->lda * D, st));
    LPR_HIP(hipMemsetAsync(s->z, 0, D, st));
    std::memset(s->h_state, 0, sizeof(RevState));
    s->h_state->status = LPR_OK_OPTIMAL;
    s->h_state->entering = -1;
    s->h_state->log_cap = s->log_cap;
    LPR_HIP(hipMemcpyAsync(s->state, s->h_state, sizeof(RevState), hipMemcpyHostToDevice, st));
    rev_launch_init(s);
    LPR_HIP(hipGetLastError());
    e->live_rev.push_back(s);
    *out = s;
    return LPR_OK_OPTIMAL;
}

static int rev_ensure_log(lpr_revised* s, int64_t need) {
    if (need <= s->log_cap) return LPR_OK_OPTIMAL;
    int64_t cap = s->log_cap;
    while (cap < need) cap *= 2;
    int32_t* nl = nullptr;
    LPR_HIP(hipMalloc(&nl, (size_t)cap * 3 * sizeof(int32_t)));
    LPR_HIP(hipMemcpyAsync(nl, s->log, (size_t)s->log_cap * 3 * sizeof(int32_t),
                           hipMemcpyDeviceToDevice, s->eng->stream));
    LPR_HIP(hipStreamSynchronize(s->eng->stream));
    LPR_HIP(hipFree(s->log));
    s->log = nl;
    s->log_cap = cap;
    return LPR_OK_OPTIMAL;
}

}  // namespace lpr

using namespace lpr;

#define LPR_LIVE_REV(s)                                                              \
    do {                                                                             \
        if (!(s) || !(s)->eng) {                                                     \
            set_error("revised-solver handle is null or its engine has been closed"); \
            return LPR_BAD_ARGUMENT;                                                 \
        }                                                                            \
    } while (0)

extern "C" {

int lpr_revised_create(lpr_engine* e, int n, int m, const double* objective, const double* A,
                       int lda, const double* b, int is_min, lpr_revised** out) {
    if (!objective || !A || !b || lda < n) {
        set_error("lpr_revised_create: bad arguments (n=%d m=%d lda=%d)", n, m, lda);
        return LPR_BAD_ARGUMENT;
    }
    lpr_revised* s = nullptr;
    int rc = rev_alloc(e, n, m, is_min, &s);
    if (rc != LPR_OK_OPTIMAL) return rc;
    hipStream_t st = e->stream;
    std::vector<double> cc(n);
    for (int j = 0; j < n; ++j) cc[j] = is_min ? -objective[j] : objective[j];  // :51
    hipError_t err = hipMemcpy2DAsync(s->A, (size_t)s->lda * sizeof(double), A,
                                      (size_t)lda * sizeof(double), (size_t)n * sizeof(double), m,
                                      hipMemcpyHostToDevice, st);
    auto chk = [&](hipError_t x) { if (err == hipSuccess) err = x; };
    chk(hipMemcpyAsync(s->b, b, (size_t)m * sizeof(double), hipMemcpyHostToDevice, st));
    chk(hipMemcpyAsync(s->cOrig, objective, (size_t)n * sizeof(double), hipMemcpyHostToDevice,
                       st));
    chk(hipMemcpyAsync(s->c, cc.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    if (err == hipSuccess) {
        rev_launch_transpose_a(s);
        chk(hipGetLastError());
    }
    chk(hipStreamSynchronize(st));  // inputs (and cc) are borrowed for this call only
    if (err != hipSuccess) {
        set_error("lpr_revised_create: %s", hipGetErrorString(err));
        lpr_revised_destroy(s);
        return LPR_DEVICE_ERROR;
    }
    *out = s;
    return LPR_OK_OPTIMAL;
}

int lpr_revised_synthetic(lpr_engine* e, int m, int n, uint64_t seed, lpr_revised** out) {
    lpr_revised* s = nullptr;
    int rc = rev_alloc(e, n, m, 0, &s);
    if (rc != LPR_OK_OPTIMAL) return rc;
    rev_launch_synthetic(s, seed);
    rev_launch_transpose_a(s);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err != hipSuccess) {
        set_error("lpr_revised_synthetic: %s", hipGetErrorString(err));
        lpr_revised_destroy(s);
        return LPR_DEVICE_ERROR;
    }
    *out = s;
    return LPR_OK_OPTIMAL;
}

int lpr_revised_destroy(lpr_revised* s) {
    if (!s) return LPR_BAD_ARGUMENT;
    if (s->eng) {
        rev_release_device(s);
        auto& lv = s->eng->live_rev;
        for (size_t k = 0; k < lv.size(); ++k)
            if (lv[k] == s) {
                lv.erase(lv.begin() + k);
                break;
            }
    }
    delete s;
    return LPR_OK_OPTIMAL;
}

int lpr_revised_solve(lpr_revised* s, const lpr_solve_opts* opts, lpr_revised_result* res) {
    LPR_LIVE_REV(s);
    if (!res) return LPR_BAD_ARGUMENT;
    lpr_solve_opts o;
    std::memset(&o, 0, sizeof o);
    if (opts) o = *opts;
    hipStream_t st = s->eng->stream;
    LPR_HIP(hipSetDevice(s->eng->device));
    const int64_t start = s->total_iter;
    const int64_t max_iter = o.max_pivots > 0 ? start + o.max_pivots : 0;
    // iterations queued between two polls of the status word: a poll idles the device for a
    // round trip (~40 us, a fifth of an iteration at m = 4096), iterations queued behind the end
    // of a solve return at once (~15 us each): start at 8, double up to 64
    int batch = o.batch > 0 ? o.batch : 8;
    const int batch_max = o.batch > 0 ? o.batch : 64;

    RevState* hs = s->h_state;
    hs->status = kRunning;
    hs->entering = -1;
    hs->leaving_row = -1;
    hs->iter = start;
    hs->max_iter = max_iter;
    hs->log_cap = s->log_cap;
    LPR_HIP(hipMemcpyAsync(s->state, hs, sizeof(RevState), hipMemcpyHostToDevice, st));

    const char* sv = std::getenv("LPR_REV_STAMPS");
    const bool stamps = sv && sv[0] == '1';
    if (stamps && !s->dbg_stamps) LPR_HIP(hipMalloc(&s->dbg_stamps, 16 * sizeof(unsigned long long)));
    // y = c_B B^-1 of the state the call starts from; every pivot's update pass then leaves the
    // next iteration's y behind (same sums, same order: RevisedPrimalSimplexSolver.cs:93 = :219)
    rev_launch_y(s);
    int status = kRunning;
    int64_t iter = start;
    while (status == kRunning) {
        int rc = rev_ensure_log(s, iter + batch + 1);
        if (rc != LPR_OK_OPTIMAL) return rc;
        if (s->log_cap != hs->log_cap) {
            hs->log_cap = s->log_cap;
            LPR_HIP(hipMemcpyAsync(&s->state->log_cap, &hs->log_cap, sizeof(int64_t),
                                   hipMemcpyHostToDevice, st));
        }
        for (int k = 0; k < batch; ++k) {
            if (stamps && k == batch - 1) {  // the last iteration of a batch is the one stamped
                unsigned long long init[16] = {~0ull, 0, 0, 0, ~0ull, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                LPR_HIP(hipMemcpyAsync(s->dbg_stamps, init, sizeof init, hipMemcpyHostToDevice, st));
                LPR_HIP(hipStreamSynchronize(st));
            }
            rev_launch_iteration_batched(s);
        }
        if (stamps) {
            unsigned long long t[16];
            LPR_HIP(hipStreamSynchronize(st));
            LPR_HIP(hipMemcpy(t, s->dbg_stamps, sizeof t, hipMemcpyDeviceToHost));
            std::fprintf(stderr, "xu walker of workgroup 100: chunks 8..24 (2048 adds per output) %.2f us = "
                         "%.2f ns per add; entered %.2f us after the kernel's first workgroup\n",
                         (double)((long long)(t[13] - t[12])) * 0.01,
                         (double)((long long)(t[13] - t[12])) * 10.0 / 2048.0,
                         (double)((long long)(t[12] - t[4])) * 0.01);
            std::fprintf(stderr, "rev stamps (us): rc walk %.2f, to tail %.2f, enter tail %.2f | xu walk "
                         "%.2f, to tail %.2f, ratio tail %.2f | rc start -> xu start %.2f | enter: fold "
                         "%.2f gather %.2f | ratio: loads+exits %.2f replay %.2f rest %.2f\n",
                         (t[1] - t[0]) * 0.01, (double)((long long)(t[2] - t[1])) * 0.01,
                         (t[3] - t[2]) * 0.01, (t[5] - t[4]) * 0.01,
                         (double)((long long)(t[6] - t[5])) * 0.01, (t[7] - t[6]) * 0.01,
                         (t[4] - t[0]) * 0.01, (double)((long long)(t[8] - t[2])) * 0.01,
                         (double)((long long)(t[3] - t[8])) * 0.01,
                         (double)((long long)(t[9] - t[6])) * 0.01,
                         (double)((long long)(t[10] - t[9])) * 0.01,
                         (double)((long long)(t[7] - t[10])) * 0.01);
        }
        LPR_HIP(hipGetLastError());
        LPR_HIP(hipMemcpyAsync(hs, s->state, sizeof(RevState), hipMemcpyDeviceToHost, st));
        LPR_HIP(hipStreamSynchronize(st));
        const int64_t done = hs->iter - iter;
        iter = hs->iter;
        status = hs->status;
        if (batch < batch_max) batch = batch * 2 < batch_max ? batch * 2 : batch_max;
        if (max_iter > 0 && iter + batch > max_iter) {
            batch = (int)(max_iter - iter);  // no point in queueing past the caller's limit
            if (batch < 1) batch = 1;
        }
        if (status == kRunning && done == 0) {
            set_error("revised simplex loop made no progress (device status still running)");
            return LPR_DEVICE_ERROR;
        }
    }
    s->total_iter = iter;
    s->last_status = status;
    res->status = status;
    res->reserved = 0;
    res->iterations = iter - start;
    res->total_iterations = iter;
    res->z = 0.0;
    if (status == LPR_OK_OPTIMAL) {
        rev_launch_extract(s);
        LPR_HIP(hipGetLastError());
        LPR_HIP(hipMemcpyAsync(&res->z, s->z, sizeof(double), hipMemcpyDeviceToHost, st));
        LPR_HIP(hipStreamSynchronize(st));
    }
    return status;
}

static int rev_ensure_snap(lpr_revised* s) {
    if (s->snap_ratios) return LPR_OK_OPTIMAL;
    LPR_HIP(hipMalloc(&s->snap_ratios, (size_t)s->ldb * sizeof(double)));
    LPR_HIP(hipMalloc(&s->snap_basis, (size_t)s->m * sizeof(int32_t)));
    LPR_HIP(hipMalloc(&s->snap_scal, 4 * sizeof(double)));
    LPR_HIP(hipHostMalloc(&s->h_snap_scal, 4 * sizeof(double)));
    LPR_HIP(hipMemsetAsync(s->snap_scal, 0, 4 * sizeof(double), s->eng->stream));
    return LPR_OK_OPTIMAL;
}

int lpr_revised_step(lpr_revised* s, lpr_revised_snapshot_info* info) {
    LPR_LIVE_REV(s);
    if (!info) return LPR_BAD_ARGUMENT;
    hipStream_t st = s->eng->stream;
    LPR_HIP(hipSetDevice(s->eng->device));
    int rc = rev_ensure_snap(s);
    if (rc != LPR_OK_OPTIMAL) return rc;
    rc = rev_ensure_log(s, s->total_iter + 2);
    if (rc != LPR_OK_OPTIMAL) return rc;
    RevState* hs = s->h_state;
    hs->status = kRunning;
    hs->entering = -1;
    hs->leaving_row = -1;
    hs->iter = s->total_iter;
    hs->max_iter = 0;
    hs->log_cap = s->log_cap;
    LPR_HIP(hipMemcpyAsync(s->state, hs, sizeof(RevState), hipMemcpyHostToDevice, st));
    rev_launch_iteration(s, true);
    LPR_HIP(hipGetLastError());
    LPR_HIP(hipMemcpyAsync(hs, s->state, sizeof(RevState), hipMemcpyDeviceToHost, st));
    LPR_HIP(hipStreamSynchronize(st));
    std::memset(info, 0, sizeof *info);
    info->entering = hs->entering;
    info->leaving_row = -1;
    info->leaving_var = -1;
    int status = hs->status;
    if (status == kRunning) {  // a pivot was made: the post-pivot quantities of :217-247
        if (hs->iter != s->total_iter + 1) {
            set_error("revised simplex step made no progress (device status still running)");
            return LPR_DEVICE_ERROR;
        }
        int32_t trip[3] = {-1, -1, -1};
        LPR_HIP(hipMemcpyAsync(trip, s->log + 3 * s->total_iter, sizeof trip,
                               hipMemcpyDeviceToHost, st));
        s->total_iter += 1;
        rev_launch_prices(s);     // xB, y_post, rc_post (status is still "running")
        rev_launch_extract(s);    // ComputeOriginalZFromCurrentBasis(xB) (:246, :253-262)
        rev_launch_zworking(s);   // Dot(cB, xB) (:245)
        LPR_HIP(hipGetLastError());
        LPR_HIP(hipMemcpyAsync(s->h_snap_scal, s->snap_scal, 2 * sizeof(double),
                               hipMemcpyDeviceToHost, st));
        LPR_HIP(hipMemcpyAsync(s->h_snap_scal + 2, s->z, sizeof(double), hipMemcpyDeviceToHost,
                               st));
        LPR_HIP(hipStreamSynchronize(st));
        info->leaving_row = trip[0];
        info->leaving_var = trip[2];
        info->entering_rc_pre = s->h_snap_scal[0];
        info->z_working = s->h_snap_scal[1];
        info->z_original = s->h_snap_scal[2];
        status = LPR_PIVOT_LIMIT;  // "one pivot done, not finished"
    } else if (status == LPR_OK_OPTIMAL) {  // the "Optimal" snapshot of :124-146
        rev_launch_extract(s);              // ExtractSolution (:126): SolutionVector, finalZ
        rev_launch_zworking(s);
        LPR_HIP(hipGetLastError());
        LPR_HIP(hipMemcpyAsync(s->h_snap_scal + 1, s->snap_scal + 1, sizeof(double),
                               hipMemcpyDeviceToHost, st));
        LPR_HIP(hipMemcpyAsync(s->h_snap_scal + 2, s->z, sizeof(double), hipMemcpyDeviceToHost,
                               st));
        LPR_HIP(hipStreamSynchronize(st));
        info->z_working = s->h_snap_scal[1];
        info->z_original = s->h_snap_scal[2];
    } else if (status == LPR_PIVOT_TOO_SMALL) {
        // the C# has done the bookkeeping of :194-212 before UpdateBInverse throws (:267)
        s->total_iter = hs->iter;
    }
    s->last_status = status == LPR_PIVOT_LIMIT ? (int)kRunning : status;
    info->status = status;
    return status;
}

int lpr_revised_snapshot_read(lpr_revised* s, double* y, double* rc, double* u, double* ratios,
                              int32_t* basis_pre, double* xB) {
    LPR_LIVE_REV(s);
    hipStream_t st = s->eng->stream;
    LPR_HIP(hipSetDevice(s->eng->device));
    const size_t D = sizeof(double);
    if ((ratios || basis_pre) && !s->snap_ratios) {
        set_error("lpr_revised_snapshot_read: no lpr_revised_step has run on this handle");
        return LPR_BAD_ARGUMENT;
    }
    std::vector<double> yy;
    if (rc) yy.resize((size_t)s->m);
    if (y) LPR_HIP(hipMemcpyAsync(y, s->y, (size_t)s->m * D, hipMemcpyDeviceToHost, st));
    if (rc) {
        LPR_HIP(hipMemcpyAsync(rc, s->rcx, (size_t)s->n * D, hipMemcpyDeviceToHost, st));
        LPR_HIP(hipMemcpyAsync(yy.data(), s->y, (size_t)s->m * D, hipMemcpyDeviceToHost, st));
    }
    if (u) LPR_HIP(hipMemcpyAsync(u, s->u, (size_t)s->m * D, hipMemcpyDeviceToHost, st));
    if (ratios)
        LPR_HIP(hipMemcpyAsync(ratios, s->snap_ratios, (size_t)s->m * D, hipMemcpyDeviceToHost,
                               st));
    if (basis_pre)
        LPR_HIP(hipMemcpyAsync(basis_pre, s->snap_basis, (size_t)s->m * sizeof(int32_t),
                               hipMemcpyDeviceToHost, st));
    if (xB) LPR_HIP(hipMemcpyAsync(xB, s->xB, (size_t)s->m * D, hipMemcpyDeviceToHost, st));
    LPR_HIP(hipStreamSynchronize(st));
    if (rc)  // rcS_k = -y_k (:100-102, :225-227): a sign flip, exact
        for (int k = 0; k < s->m; ++k) rc[s->n + k] = -yy[(size_t)k];
    return LPR_OK_OPTIMAL;
}

int lpr_revised_binv_a_exact(lpr_revised* s, double* out) {
    LPR_LIVE_REV(s);
    if (!out) return LPR_BAD_ARGUMENT;
    hipStream_t st = s->eng->stream;
    LPR_HIP(hipSetDevice(s->eng->device));
    if (!s->gemm_out) {
        s->ldc = align_up(s->n, kLdAlign);
        LPR_HIP(hipMalloc(&s->gemm_out, (size_t)s->m * s->ldc * sizeof(double)));
    }
    rev_launch_matmul_exact(s, s->gemm_out, s->ldc);
    LPR_HIP(hipGetLastError());
    LPR_HIP(hipMemcpy2DAsync(out, (size_t)s->n * sizeof(double), s->gemm_out,
                             (size_t)s->ldc * sizeof(double), (size_t)s->n * sizeof(double), s->m,
                             hipMemcpyDeviceToHost, st));
    LPR_HIP(hipStreamSynchronize(st));
    return LPR_OK_OPTIMAL;
}

int lpr_revised_solution(lpr_revised* s, double* x, double* z) {
    LPR_LIVE_REV(s);
    hipStream_t st = s->eng->stream;
    LPR_HIP(hipSetDevice(s->eng->device));
    if (x) LPR_HIP(hipMemcpyAsync(x, s->x, (size_t)s->n * sizeof(double), hipMemcpyDeviceToHost,
                                  st));
    if (z) LPR_HIP(hipMemcpyAsync(z, s->z, sizeof(double), hipMemcpyDeviceToHost, st));
    LPR_HIP(hipStreamSynchronize(st));
    return LPR_OK_OPTIMAL;
}

int lpr_revised_basis_read(lpr_revised* s, int32_t* basis_out) {
    LPR_LIVE_REV(s);
    if (!basis_out) return LPR_BAD_ARGUMENT;
    hipStream_t st = s->eng->stream;
    LPR_HIP(hipSetDevice(s->eng->device));
    LPR_HIP(hipMemcpyAsync(basis_out, s->basic, (size_t)s->m * sizeof(int32_t),
                           hipMemcpyDeviceToHost, st));
    LPR_HIP(hipStreamSynchronize(st));
    return LPR_OK_OPTIMAL;
}

int lpr_revised_log_read(lpr_revised* s, int32_t* row_out, int32_t* enter_out,
                         int32_t* leave_out, int64_t cap, int64_t* count) {
    LPR_LIVE_REV(s);
    if (!count || cap < 0) return LPR_BAD_ARGUMENT;
    int64_t k = s->total_iter < s->log_cap ? s->total_iter : s->log_cap;
    if (k > cap) k = cap;
    *count = k;
    if (k == 0) return LPR_OK_OPTIMAL;
    LPR_HIP(hipSetDevice(s->eng->device));
    std::vector<int32_t> tmp((size_t)k * 3);
    LPR_HIP(hipMemcpy(tmp.data(), s->log, (size_t)k * 3 * sizeof(int32_t),
                      hipMemcpyDeviceToHost));
    for (int64_t q = 0; q < k; ++q) {
        if (row_out) row_out[q] = tmp[3 * q];
        if (enter_out) enter_out[q] = tmp[3 * q + 1];
        if (leave_out) leave_out[q] = tmp[3 * q + 2];
    }
    return LPR_OK_OPTIMAL;
}

int lpr_revised_binv_read(lpr_revised* s, double* out) {
    LPR_LIVE_REV(s);
    if (!out) return LPR_BAD_ARGUMENT;
    hipStream_t st = s->eng->stream;
    LPR_HIP(hipSetDevice(s->eng->device));
    LPR_HIP(hipMemcpy2DAsync(out, (size_t)s->m * sizeof(double), s->Binv,
                             (size_t)s->ldb * sizeof(double), (size_t)s->m * sizeof(double), s->m,
                             hipMemcpyDeviceToHost, st));
    LPR_HIP(hipStreamSynchronize(st));
    return LPR_OK_OPTIMAL;
}

int lpr_revised_xb_read(lpr_revised* s, double* out) {
    LPR_LIVE_REV(s);
    if (!out) return LPR_BAD_ARGUMENT;
    hipStream_t st = s->eng->stream;
    LPR_HIP(hipSetDevice(s->eng->device));
    LPR_HIP(hipMemcpyAsync(out, s->xB, (size_t)s->m * sizeof(double), hipMemcpyDeviceToHost, st));
    LPR_HIP(hipStreamSynchronize(st));
    return LPR_OK_OPTIMAL;
}

int lpr_revised_binv_a(lpr_revised* s, double* out, double* ms) {
    LPR_LIVE_REV(s);
    hipStream_t st = s->eng->stream;
    LPR_HIP(hipSetDevice(s->eng->device));
    if (!s->gemm_out) {
        s->ldc = align_up(s->n, kLdAlign);
        LPR_HIP(hipMalloc(&s->gemm_out, (size_t)s->m * s->ldc * sizeof(double)));
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ms) {
        LPR_HIP(hipEventCreate(&e0));
        LPR_HIP(hipEventCreate(&e1));
        LPR_HIP(hipEventRecord(e0, st));
    }
    rev_launch_gemm(s, s->gemm_out, s->ldc);
    LPR_HIP(hipGetLastError());
    if (ms) LPR_HIP(hipEventRecord(e1, st));
    if (out)
        LPR_HIP(hipMemcpy2DAsync(out, (size_t)s->n * sizeof(double), s->gemm_out,
                                 (size_t)s->ldc * sizeof(double), (size_t)s->n * sizeof(double),
                                 s->m, hipMemcpyDeviceToHost, st));
    LPR_HIP(hipStreamSynchronize(st));
    if (ms) {
        float f = 0.f;
        LPR_HIP(hipEventElapsedTime(&f, e0, e1));
        *ms = f;
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
    return LPR_OK_OPTIMAL;
}

}  // extern "C"
