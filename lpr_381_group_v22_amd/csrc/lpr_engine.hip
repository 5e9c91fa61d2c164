// lpr_engine.hip -- C ABI (include/lpr_engine.h): engine + full-tableau primal simplex driver.
// The pivot loop of PrimalSimplexSolver.Solve() (Simplex/PrimalSimplexSolver.cs:102-150) runs on
// the device: the host only queues batches of (k_select, k_update) pairs and polls one status
// word per batch.  There is no CPU fallback: without a gfx950 device lpr_engine_open fails.
#include "engine_common.hpp"

#include <cstdlib>

#include <cstdarg>
#include <new>

#pragma clang fp contract(off)

namespace lpr {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}
const char* get_error() { return g_err.c_str(); }

// primal_kernels.hip
void launch_select(lpr_tableau* t, int mode, int e_in, int r_in, int32_t* out_i);
void launch_bootstrap(lpr_tableau* t);
void launch_pivot_head(lpr_tableau* t);
void launch_update(lpr_tableau* t, int variant, int check_status, int dump_next);
int num_update_variants();
void launch_extract(lpr_tableau* t, int n, double* x);
void launch_pivot_fused(lpr_tableau* t, const double* in, double* out);
void launch_build(lpr_tableau* t, int n, int m, const double* d_obj, const double* d_A, int lda,
                  const int32_t* d_ncoef, const int8_t* d_rel, const double* d_rhs, int is_max);
void launch_synthetic(lpr_tableau* t, int m, int n, uint64_t seed);

// block_kernels.hip
int blk_max_pivots();
void blk_release(lpr_tableau* t);
int blk_ensure(lpr_tableau* t);
int blk_upload_state(lpr_tableau* t, int64_t iter, int64_t max_iter);
int blk_set_log_cap(lpr_tableau* t);
int blk_poll(lpr_tableau* t, int32_t* status, int32_t* pending, int64_t* iter);
void blk_launch_bootstrap(lpr_tableau* t);
void blk_launch_heads(lpr_tableau* t, int K);
void blk_launch_update(lpr_tableau* t, int tr);
// overlap_kernels.hip
int ov_max_pivots();
void ov_release(lpr_tableau* t);
// small_kernels.hip
bool small_fits(const lpr_tableau* t);
int small_pivots_per_block();
void small_release(lpr_tableau* t);
int small_ensure(lpr_tableau* t);
int small_upload_state(lpr_tableau* t, int64_t iter, int64_t max_iter);
int small_set_log_cap(lpr_tableau* t);
void small_launch_block(lpr_tableau* t);
int small_poll(lpr_tableau* t, int32_t* status, int64_t* iter);
int ov_ensure(lpr_tableau* t, bool second_buffer);
int ov_begin(lpr_tableau* t, int64_t iter, int64_t max_iter);
int ov_set_log(lpr_tableau* t, int parity);
void ov_launch_step(lpr_tableau* t, int K, int tr, int lp);
void ov_launch_heads(lpr_tableau* t, int K, int flags);
int ov2_begin(lpr_tableau* t);
int ov2_launch_step(lpr_tableau* t, int K, int tr, int lp, int flags, hipEvent_t ev_start,
                    hipEvent_t ev_stop);
int ov_read_stamps(lpr_tableau* t, uint64_t* out, int64_t cap, int64_t* count);
int ov2_join(lpr_tableau* t);
void ov_launch_sweep(lpr_tableau* t, int tr);
struct OvPoll {
    int32_t status, pending, kdone, cur, error;
    int64_t applied;
    double z[2];  // RHS column entry 0 (= T[0, cols-1]) after an even / odd number of pivots
};
int ov_poll(lpr_tableau* t, int parity, OvPoll* out);
void ov_adopt_buffer(lpr_tableau* t, int cur);
// revised_engine.hip
void rev_orphan(lpr_revised* s);
// bb_engine.hip
void bb_orphan(lpr_bb* b);
// sens_engine.hip
void sens_orphan(lpr_sens* s);
void comm_orphan(lpr_comm* c);
}  // namespace lpr
// cut_kernels.hip
void lpr_cut_release(lpr_tableau* t);
namespace lpr {

enum : int { kSelEnter = 1, kSelLeave = 2, kSelCommit = 4, kSelFull = 7 };
constexpr int kTimeStride = 4;  // opts.time_kernels samples one update launch in four (one-pivot
                                // and 0x60tr paths; the overlapped paths time every sweep)

static int alloc_tableau(lpr_engine* e, int rows, int cols, lpr_tableau** out) {
    if (!e || !out || rows < 1 || cols < 2 || rows > 65535) {
        set_error("bad tableau shape %d x %d", rows, cols);
        return LPR_BAD_ARGUMENT;
    }
    LPR_HIP(hipSetDevice(e->device));
    lpr_tableau* t = new (std::nothrow) lpr_tableau();
    if (!t) return LPR_OUT_OF_MEMORY;
    t->eng = e;
    t->rows = rows;
    t->cols = cols;
    t->ld = align_up(cols, kLdAlign);
    t->log_cap = 1 << 16;
    // Test hook: a small initial pivot-log capacity (LPR_TEST_LOG_CAP pairs, 16..65536) so that the
    // growth path (ensure_log + the K-pivot paths' ov_set_log / blk_set_log_cap) is crossed after a
    // few blocks, inside a solve short enough for the CPU oracle to check (tests/test_block_gpu.py).
    if (const char* lc = std::getenv("LPR_TEST_LOG_CAP")) {
        const long v = std::strtol(lc, nullptr, 10);
        if (v >= 16 && v <= (1 << 16)) t->log_cap = v;
    }
    const size_t tbytes = (size_t)rows * t->ld * sizeof(double);
    hipError_t err = hipSuccess;
    auto chk = [&](hipError_t x) { if (err == hipSuccess) err = x; };
    chk(hipMalloc(&t->T, tbytes));
    chk(hipMalloc(&t->rowbuf, (size_t)t->ld * sizeof(double)));
    chk(hipMalloc(&t->colbuf, (size_t)align_up(rows, 16) * sizeof(double)));
    chk(hipMalloc(&t->next_col, (size_t)align_up(rows, 16) * sizeof(double)));
    chk(hipMalloc(&t->next_rhs, (size_t)align_up(rows, 16) * sizeof(double)));
    chk(hipMalloc(&t->zparts, (size_t)2 * kMaxHeadGroups * 16));
    chk(hipMalloc(&t->basis, (size_t)(rows > 1 ? rows - 1 : 1) * sizeof(int32_t)));
    chk(hipMalloc(&t->log, (size_t)t->log_cap * 2 * sizeof(int32_t)));
    chk(hipMalloc(&t->state, sizeof(PivotState)));
    chk(hipMalloc(&t->scratch_i, 16 * sizeof(int32_t)));
    chk(hipHostMalloc(&t->h_state, sizeof(PivotState)));
    chk(hipHostMalloc(&t->h_scratch_i, 16 * sizeof(int32_t)));
    if (err != hipSuccess) {
        set_error("device allocation of a %d x %d tableau failed: %s", rows, cols,
                  hipGetErrorString(err));
        lpr_tableau_destroy(t);
        return err == hipErrorOutOfMemory ? LPR_OUT_OF_MEMORY : LPR_DEVICE_ERROR;
    }
    std::memset(t->h_state, 0, sizeof(PivotState));
    t->h_state->status = LPR_OK_OPTIMAL;
    t->h_state->log_cap = t->log_cap;
    LPR_HIP(hipMemcpyAsync(t->state, t->h_state, sizeof(PivotState), hipMemcpyHostToDevice,
                           e->stream));
    LPR_HIP(hipMemsetAsync(t->T, 0, tbytes, e->stream));
    LPR_HIP(hipMemsetAsync(t->rowbuf, 0, (size_t)t->ld * sizeof(double), e->stream));
    e->live.push_back(t);
    *out = t;
    return LPR_OK_OPTIMAL;
}

static void drop_graph(lpr_tableau* t) {
    if (t->graph) {
        hipGraphExecDestroy(t->graph);
        t->graph = nullptr;
    }
    t->graph_batch = 0;
    t->graph_variant = -1;
    t->graph_key = lpr_tableau::GraphKey();
}

// what a capture made now would bake in
static lpr_tableau::GraphKey graph_key_now(const lpr_tableau* t) {
    lpr_tableau::GraphKey k;
    k.T = t->T;
    k.T2 = t->T2;
    k.log = t->log;
    k.basis = t->basis;
    k.blk = t->blk;
    k.next_col = t->next_col;
    k.colbuf = t->colbuf;
    k.rows = t->rows;
    k.cols = t->cols;
    k.ld = t->ld;
    return k;
}

static bool graph_valid(const lpr_tableau* t, int batch, int variant) {
    return t->graph && t->graph_batch == batch && t->graph_variant == variant &&
           t->graph_key == graph_key_now(t);
}

static void graph_stamp(lpr_tableau* t, int batch, int variant) {
    t->graph_batch = batch;
    t->graph_variant = variant;
    t->graph_key = graph_key_now(t);
}

// Frees everything the tableau holds on the device (the engine must still be alive).
static void release_device(lpr_tableau* t) {
    hipSetDevice(t->eng->device);
    if (t->eng->stream) hipStreamSynchronize(t->eng->stream);
    drop_graph(t);
    lpr_cut_release(t);
    blk_release(t);
    ov_release(t);
    small_release(t);
    for (hipEvent_t ev : t->ev) hipEventDestroy(ev);
    t->ev.clear();
    hipFree(t->T);
    hipFree(t->T2);
    hipFree(t->rowbuf);
    hipFree(t->colbuf);
    hipFree(t->next_col);
    hipFree(t->next_rhs);
    hipFree(t->zparts);
    hipFree(t->basis);
    hipFree(t->log);
    hipFree(t->state);
    hipFree(t->scratch_i);
    hipFree(t->xbuf);
    if (t->h_state) hipHostFree(t->h_state);
    if (t->h_scratch_i) hipHostFree(t->h_scratch_i);
    t->T = t->rowbuf = t->colbuf = t->xbuf = t->next_col = t->next_rhs = nullptr;
    t->zparts = nullptr;
    t->T2 = nullptr;
    t->basis = t->log = t->scratch_i = t->h_scratch_i = nullptr;
    t->state = t->h_state = nullptr;
}

// Grow the pivot log so that `need` pairs fit (keeps the old entries).
static int ensure_log(lpr_tableau* t, int64_t need) {
    if (need <= t->log_cap) return LPR_OK_OPTIMAL;
    int64_t cap = t->log_cap;
    while (cap < need) cap *= 2;
    int32_t* nl = nullptr;
    LPR_HIP(hipMalloc(&nl, (size_t)cap * 2 * sizeof(int32_t)));
    LPR_HIP(hipMemcpyAsync(nl, t->log, (size_t)t->log_cap * 2 * sizeof(int32_t),
                           hipMemcpyDeviceToDevice, t->eng->stream));
    LPR_HIP(hipStreamSynchronize(t->eng->stream));
    LPR_HIP(hipFree(t->log));
    t->log = nl;
    t->log_cap = cap;
    drop_graph(t);  // the log pointer is a captured kernel argument
    return LPR_OK_OPTIMAL;
}

// Tile shape: the tallest row tile (more loads in flight per lane, pivot-row slice reused over
// more rows) that still gives the 256 CUs >= 8 workgroups each; measured on the 4097 x 12289
// tableau TR=32 is the fastest (profiles/).  The serpentine sweep (bit 8) reverses the tile order
// on alternate pivots so the tail of one sweep is re-read from the 256 MiB Infinity Cache.
static int default_variant(const lpr_tableau* t) {
    const long ctiles = (t->ld / 2 + 255) / 256;
    struct { int tr; int idx; } opts[] = {{32, 5}, {16, 0}, {8, 1}, {4, 8}, {2, 9}, {1, 10}};
    for (auto& o : opts) {
        const long blocks = ctiles * ((t->rows + o.tr - 1) / o.tr);
        if (blocks >= 2048 || o.tr == 1) return o.idx | 0x100;
    }
    return 0x100;
}

static int default_batch(const lpr_tableau* t) {
    // aim at a few milliseconds of device work between host polls
    const double bytes = 16.0 * t->rows * (double)t->ld;
    const double us = bytes / 4.0e6 + 6.0;  // ~4 TB/s + select/launch overhead
    int b = (int)(4000.0 / us);
    if (b < 8) b = 8;
    if (b > 512) b = 512;
    return b;
}

// Small tableaux (<= kFusedBytes): one k_pivot_fused launch per pivot, ping-pong between T and
// T2.  opts.variant == 0x7fff forces the two-kernel path (tests), 0x7ffe forces the fused one.
// (round 2, after the wave-granular hand-offs of the loop heads and the 8-row tiles of the small
// in-place sweep, pivots/s K-pivot path vs this one: m = 256 (1.6 MB) 143.6 k vs 137.9 k, m = 320
// 141.9 k vs 121.0 k, m = 512 137.7 k vs 99.7 k.  Below ~1 MB a solve is a few dozen pivots and the
// start-up of the K-pivot path -- prologue, a fill and a drain step -- is what counts.)
static constexpr size_t kFusedBytes = (size_t)1 << 20;

static bool use_fused(const lpr_tableau* t, const lpr_solve_opts& o) {
    if (o.time_kernels) return false;
    if (o.block >= 2) return false;  // the K-pivots-per-sweep path was asked for explicitly
    if (o.variant == 0x7fff) return false;
    if (o.variant == 0x7ffe) return true;
    if (o.variant != 0) return false;
    return (size_t)t->rows * t->ld * sizeof(double) <= kFusedBytes;
}

static int solve_fused(lpr_tableau* t, const lpr_solve_opts& o, lpr_solve_result* res) {
    lpr_engine* e = t->eng;
    hipStream_t s = e->stream;
    const size_t tbytes = (size_t)t->rows * t->ld * sizeof(double);
    if (!t->T2) {
        LPR_HIP(hipMalloc(&t->T2, tbytes));
        LPR_HIP(hipMemsetAsync(t->T2, 0, tbytes, s));
    }
    int batch = o.batch > 0 ? o.batch : default_batch(t);
    batch = (batch + 1) & ~1;  // even: a full batch leaves the tableau in the buffer it started in
    const int64_t start_iter = t->total_pivots;
    const int64_t max_iter = o.max_pivots > 0 ? start_iter + o.max_pivots : 0;
    PivotState* hs = t->h_state;
    hs->status = kRunning;
    hs->iter = start_iter;
    hs->max_iter = 0;  // the host enforces the limit on this path
    hs->log_cap = t->log_cap;
    LPR_HIP(hipMemcpyAsync(&t->state->status, &hs->status, sizeof(int32_t), hipMemcpyHostToDevice,
                           s));
    LPR_HIP(hipMemcpyAsync(&t->state->iter, &hs->iter, 3 * sizeof(int64_t),
                           hipMemcpyHostToDevice, s));
    int status = kRunning;
    int64_t iter = start_iter;
    while (status == kRunning) {
        int nb = batch;
        if (max_iter > 0 && max_iter - iter < nb) nb = (int)(max_iter - iter);
        if (nb <= 0) break;
        int rc = ensure_log(t, iter + nb + 1);
        if (rc != LPR_OK_OPTIMAL) return rc;
        if (t->log_cap != hs->log_cap) {
            hs->log_cap = t->log_cap;
            LPR_HIP(hipMemcpyAsync(&t->state->log_cap, &hs->log_cap, sizeof(int64_t),
                                   hipMemcpyHostToDevice, s));
        }
        // replay a captured batch (valid for the current T / T2 orientation) -- when one exists
        // already, or when enough work is ahead to pay for capturing it (a few ms)
        const bool have = graph_valid(t, nb, -2);
        const bool full = (nb == batch) && (have || max_iter == 0 || max_iter - iter >= 4 * batch);
        if (full) {
            if (!have) {
                drop_graph(t);
                hipGraph_t g = nullptr;
                LPR_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int k = 0; k < nb; ++k)
                    launch_pivot_fused(t, (k & 1) ? t->T2 : t->T, (k & 1) ? t->T : t->T2);
                LPR_HIP(hipStreamEndCapture(s, &g));
                hipError_t ierr = hipGraphInstantiate(&t->graph, g, nullptr, nullptr, 0);
                hipGraphDestroy(g);
                if (ierr != hipSuccess) {
                    t->graph = nullptr;
                    set_error("hipGraphInstantiate failed: %s", hipGetErrorString(ierr));
                    return LPR_DEVICE_ERROR;
                }
                graph_stamp(t, nb, -2);
            }
            LPR_HIP(hipGraphLaunch(t->graph, s));
        } else {
            for (int k = 0; k < nb; ++k)
                launch_pivot_fused(t, (k & 1) ? t->T2 : t->T, (k & 1) ? t->T : t->T2);
        }
        LPR_HIP(hipGetLastError());
        LPR_HIP(hipMemcpyAsync(hs, t->state, sizeof(PivotState), hipMemcpyDeviceToHost, s));
        LPR_HIP(hipStreamSynchronize(s));
        const int64_t done = hs->iter - iter;
        if (done & 1) {  // an odd number of pivots happened: the live tableau is in T2
            double* tmp = t->T;
            t->T = t->T2;
            t->T2 = tmp;
        }
        iter = hs->iter;
        status = hs->status;
        if (status == kRunning && done != nb) {
            set_error("fused pivot loop lost a launch (done %lld of %d)", (long long)done, nb);
            return LPR_DEVICE_ERROR;
        }
    }
    if (status == kRunning) {  // stopped by the pivot limit: classify the current tableau
        launch_select(t, kSelEnter, -1, -1, t->scratch_i);
        LPR_HIP(hipMemcpyAsync(t->h_scratch_i, t->scratch_i, 4 * sizeof(int32_t),
                               hipMemcpyDeviceToHost, s));
        LPR_HIP(hipStreamSynchronize(s));
        const int ec = t->h_scratch_i[0];
        if (ec < 0) status = LPR_OK_OPTIMAL;
        else {
            launch_select(t, kSelLeave, ec, -1, t->scratch_i);
            LPR_HIP(hipMemcpyAsync(t->h_scratch_i, t->scratch_i, 4 * sizeof(int32_t),
                                   hipMemcpyDeviceToHost, s));
            LPR_HIP(hipStreamSynchronize(s));
            status = t->h_scratch_i[1] < 0 ? LPR_UNBOUNDED : LPR_PIVOT_LIMIT;
        }
    }
    t->total_pivots = iter;
    res->status = status;
    res->block = 1;
    res->pivots = iter - start_iter;
    res->total_pivots = iter;
    double z = 0.0;
    LPR_HIP(hipMemcpyAsync(&z, t->T + (t->cols - 1), sizeof(double), hipMemcpyDeviceToHost, s));
    LPR_HIP(hipStreamSynchronize(s));
    res->z = z;
    return status;
}

// Cache-resident tableaux (R <= 1024, ld <= 2048: small_kernels.hip): the 16 loop heads of a block
// in ONE workgroup, hand-offs through LDS, then one in-place sweep.  The default for every tableau
// that fits (opts.variant == 0, opts.block == 0); opts.variant 0x20xx forces it.
// Measured on the round's final code (tools/size_sweep.sh, profiles/r03_size_sweep.jsonl; pivots/s,
// this path vs heads-then-sweep 0x4008): m = 512 (6.3 MB) 214 k vs 154 k; 640 x 1280 (9.9 MB) 206 k
// vs 150 k; 1020 x 400 (11.6 MB) 194 k vs 110 k; 800 x 1200 (12.8 MB) 193 k vs 145 k; m = n = 1000
// (16 MB) 184 k vs 146 k; 1023 x 1000 (16.6 MB, the largest that fits) 183 k vs 146 k: it wins
// wherever it fits.  (Mid round 3 the 16 MB case read 132 k vs 140 k and the default stopped at
// 10 MB; the later work on the heads moved it.)
static bool use_small(const lpr_tableau* t, const lpr_solve_opts& o) {
    if (!small_fits(t)) return false;
    if ((o.variant & 0xff00) == 0x2000) return true;
    return o.variant == 0 && o.block == 0;
}

static int solve_small(lpr_tableau* t, const lpr_solve_opts& o, lpr_solve_result* res) {
    hipStream_t s = t->eng->stream;
    int rc = small_ensure(t);
    if (rc != LPR_OK_OPTIMAL) return rc;
    const int K = small_pivots_per_block();
    const int64_t start_iter = t->total_pivots;
    const int64_t max_iter = o.max_pivots > 0 ? start_iter + o.max_pivots : 0;
    rc = small_upload_state(t, start_iter, max_iter);
    if (rc != LPR_OK_OPTIMAL) return rc;
    // blocks queued between two polls of the control block: 8, doubling up to 64 (a poll idles the
    // device for a round trip; a block queued behind the end of the solve returns at once)
    int nb = o.batch > 0 ? (o.batch + K - 1) / K : 8;
    const int nb_max = o.batch > 0 ? nb : 64;
    int32_t status = kRunning;
    int64_t iter = start_iter;
    while (status == kRunning) {
        int q = nb;
        if (max_iter > 0) {  // the blocks the limit can use, + the head that reports the end
            const int64_t need = (max_iter - iter + K - 1) / K + 1;
            if (need < q) q = (int)(need < 1 ? 1 : need);
        }
        const int64_t log_before = t->log_cap;
        rc = ensure_log(t, iter + (int64_t)q * K + 1);
        if (rc != LPR_OK_OPTIMAL) return rc;
        if (t->log_cap != log_before) {
            rc = small_set_log_cap(t);
            if (rc != LPR_OK_OPTIMAL) return rc;
        }
        for (int k = 0; k < q; ++k) small_launch_block(t);
        LPR_HIP(hipGetLastError());
        rc = small_poll(t, &status, &iter);
        if (rc != LPR_OK_OPTIMAL) return rc;
        if (nb < nb_max) nb = nb * 2 < nb_max ? nb * 2 : nb_max;
    }
    t->total_pivots = iter;
    res->status = status;
    res->block = K;
    res->pivots = iter - start_iter;
    res->total_pivots = iter;
    double z = 0.0;
    LPR_HIP(hipMemcpyAsync(&z, t->T + (t->cols - 1), sizeof(double), hipMemcpyDeviceToHost, s));
    LPR_HIP(hipStreamSynchronize(s));
    res->z = z;
    return status;
}

// Tableaux above kFusedBytes: K pivots per sweep.  opts.block: 0 = auto (16), 1 = the
// one-pivot-per-sweep path, 2..16 = that many (2..8 in block_kernels.hip's form, variant 0x60tr).
// opts.variant: 0 = by size (heads-then-sweep up to kOverlapBytes, two-stream overlap above),
// 0x30tr / 0x40tr / 0x50tr / 0x60tr force a form with tr-row sweep tiles, any other non-zero value
// (a k_update tile variant, 0x7fff) the one-pivot path.
static constexpr int kDefaultBlock = 16;
// (round 2, tools/size_sweep.sh: with 8-10 us loop heads the two-stream overlap wins from ~80 MB:
// 67 MB 99.6 k vs 104.2 k pivots/s for heads-then-sweep, 101 MB 98.7 k vs 92.2 k, 227 MB 93.6 k vs
// 78.6 k; round 1's crossover was ~300 MB)
static constexpr size_t kOverlapBytes = (size_t)80 << 20;

static int block_size(const lpr_tableau* t, const lpr_solve_opts& o) {
    // a specific one-pivot update-kernel variant was asked for (0x60tr = this path, tile rows tr)
    if (o.variant != 0 && (o.variant & 0xff00) != 0x6000 && (o.variant & 0xff00) != 0x5000 &&
        (o.variant & 0xff00) != 0x4000 && (o.variant & 0xff00) != 0x3000)
        return 1;
    int k = o.block;
    if (k == 0) k = kDefaultBlock;
    if (k < 1) k = 1;
    const int kmax = ((o.variant & 0xff00) == 0x6000) ? blk_max_pivots() : ov_max_pivots();
    if (k > kmax) k = kmax;
    if (t->rows < 2) k = 1;
    return k;
}

static int solve_blocked(lpr_tableau* t, const lpr_solve_opts& o, int K, lpr_solve_result* res) {
    lpr_engine* e = t->eng;
    hipStream_t s = e->stream;
    int rc = blk_ensure(t);
    if (rc != LPR_OK_OPTIMAL) return rc;
    const bool timed = o.time_kernels != 0;
    const int tr = ((o.variant & 0xff00) == 0x6000) ? (o.variant & 0xff) : 8;  // sweep tile rows
    int nblocks = o.batch > 0 ? (o.batch + K - 1) / K : (default_batch(t) + K - 1) / K;
    if (nblocks < 1) nblocks = 1;
    const int64_t start_iter = t->total_pivots;
    const int64_t max_iter = o.max_pivots > 0 ? start_iter + o.max_pivots : 0;
    rc = blk_upload_state(t, start_iter, max_iter);
    if (rc != LPR_OK_OPTIMAL) return rc;
    blk_launch_bootstrap(t);

    int32_t status = kRunning, pending = kRunning;
    int64_t iter = start_iter;
    while (status == kRunning && pending == kRunning) {
        int nb = nblocks;
        if (max_iter > 0) {  // no more blocks than the limit can use (+1: the deciding head)
            const int64_t left = max_iter - iter;
            const int64_t need = left / K + 1;
            if (need < nb) nb = (int)need;
        }
        const int64_t log_before = t->log_cap;
        rc = ensure_log(t, iter + (int64_t)nb * K + 1);
        if (rc != LPR_OK_OPTIMAL) return rc;
        if (t->log_cap != log_before) {
            rc = blk_set_log_cap(t);
            if (rc != LPR_OK_OPTIMAL) return rc;
        }
        if (timed) {
            while ((int)t->ev.size() < 2 * nb) {
                hipEvent_t ev;
                LPR_HIP(hipEventCreate(&ev));
                t->ev.push_back(ev);
            }
            for (int k = 0; k < nb; ++k) {
                blk_launch_heads(t, K);
                const bool sample = (k % kTimeStride) == 0;
                if (sample) LPR_HIP(hipEventRecord(t->ev[2 * k], s));
                blk_launch_update(t, tr);
                if (sample) LPR_HIP(hipEventRecord(t->ev[2 * k + 1], s));
            }
        } else {
            const int gv = -100 - K * 64 - tr;  // graph key of this path
            if (!graph_valid(t, nb, gv)) {
                drop_graph(t);
                hipGraph_t g = nullptr;
                LPR_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int k = 0; k < nb; ++k) {
                    blk_launch_heads(t, K);
                    blk_launch_update(t, tr);
                }
                LPR_HIP(hipStreamEndCapture(s, &g));
                hipError_t ierr = hipGraphInstantiate(&t->graph, g, nullptr, nullptr, 0);
                hipGraphDestroy(g);
                if (ierr != hipSuccess) {
                    t->graph = nullptr;
                    set_error("hipGraphInstantiate failed: %s", hipGetErrorString(ierr));
                    return LPR_DEVICE_ERROR;
                }
                graph_stamp(t, nb, gv);
            }
            LPR_HIP(hipGraphLaunch(t->graph, s));
        }
        LPR_HIP(hipGetLastError());
        int64_t now = iter;
        rc = blk_poll(t, &status, &pending, &now);
        if (rc != LPR_OK_OPTIMAL) return rc;
        const int64_t done = now - iter;
        if (timed) {  // only sweeps that applied a full block count (K pivots each)
            const int64_t full = done / K;
            for (int64_t k = 0; k < full && k < nb; k += kTimeStride) {
                float ms = 0.f;
                LPR_HIP(hipEventElapsedTime(&ms, t->ev[2 * k], t->ev[2 * k + 1]));
                t->timed_total_ms += ms;
                t->timed_launches += 1;
            }
        }
        iter = now;
        if (status == kRunning && pending == kRunning && done == 0) {
            set_error("blocked pivot loop made no progress (device status still running)");
            return LPR_DEVICE_ERROR;
        }
    }
    if (status == kRunning) status = pending;  // decided by the last block, not yet published
    t->total_pivots = iter;
    res->status = status;
    res->block = K;
    res->pivots = iter - start_iter;
    res->total_pivots = iter;
    double z = 0.0;
    LPR_HIP(hipMemcpyAsync(&z, t->T + (t->cols - 1), sizeof(double), hipMemcpyDeviceToHost, s));
    LPR_HIP(hipStreamSynchronize(s));
    res->z = z;
    return status;
}

// Large tableaux, default: K pivots per sweep with the next block's loop heads running beside the
// current block's (out-of-place) sweep -- overlap_kernels.hip.
//
// A "step" is one launch pair: step j sweeps block j (K pivots) while its heads decide block j + 1.
// A call that applies N blocks takes N + 1 steps (the first only decides, the last only sweeps);
// the host learns the outcome from the control block the last step wrote: `status`, or -- when the
// heads found the end (`pending`) and left nothing staged (`kdone == 0`) -- `pending` itself, so no
// further launch is needed to publish it.
// opts.time_kernels: HIP events on the sweep's own stream bracket EVERY sweep launch (start =
// both kernels of the previous step done); sweep time = stop - start, step time = start of the
// next step - start of this one.  Only steps that swept a full block are counted.
static int solve_overlapped(lpr_tableau* t, const lpr_solve_opts& o, int K, int tr, bool overlap,
                            bool two_streams, int hflags, lpr_solve_result* res) {
    lpr_engine* e = t->eng;
    hipStream_t s = e->stream;
    int rc = ov_ensure(t, overlap);
    if (rc != LPR_OK_OPTIMAL) return rc;
    const bool timed = o.time_kernels != 0;
    // opts.time_kernels = n > 1: events around every n-th step only (two event records per step
    // cost the two-stream pipeline ~3 % -- they sit on the sweep's stream, on the critical cycle)
    const int tstride = o.time_kernels > 1 ? (o.time_kernels < 16 ? o.time_kernels : 16) : 1;
    // steps between host polls: about 5 ms of device work (a step = one sweep at ~5 TB/s, or K loop
    // heads at ~8 us, whichever is longer); a poll costs the device ~0.1 ms of idling
    int nlaunch;
    if (o.batch > 0) {
        nlaunch = (o.batch + K - 1) / K;
    } else {
        const double sweep_us = 16.0 * t->rows * (double)t->ld / 5.0e6;
        const double step_us = sweep_us > 8.0 * K ? sweep_us : 8.0 * K;
        nlaunch = (int)(5000.0 / step_us);
        if (nlaunch > 64) nlaunch = 64;
    }
    if (nlaunch < 2) nlaunch = 2;
    const int64_t start_iter = t->total_pivots;
    const int64_t max_iter = o.max_pivots > 0 ? start_iter + o.max_pivots : 0;
    rc = ov_begin(t, start_iter, max_iter);
    if (rc != LPR_OK_OPTIMAL) return rc;
    if (two_streams && ov2_begin(t) != LPR_OK_OPTIMAL)
        two_streams = false;  // no second stream: the one-launch form of the overlap does the same

    OvPoll pl;
    pl.status = kRunning;
    pl.pending = kRunning;
    pl.applied = start_iter;
    int32_t status = kRunning;
    int64_t applied = start_iter;
    int64_t step = 0;            // launches (pairs) queued by this call; parity = control block
    int64_t swept_full = 0;      // full blocks swept so far by this call (for the event windows)
    int idle_batches = 0;
    while (status == kRunning) {
        int nb = nlaunch;
        if (pl.pending != kRunning) {
            nb = 1;  // the end has been found: the staged tail is swept, the outcome published
        } else if (max_iter > 0) {  // the blocks the limit allows (+ the step that only decides)
            const int64_t left = max_iter - applied;
            const int64_t need = (left + K - 1) / K + (step == 0 ? 1 : 0);
            if (need < nb) nb = (int)need;
        }
        if (nb < 1) nb = 1;
        const int64_t log_before = t->log_cap;
        const int32_t* log_ptr = t->log;
        rc = ensure_log(t, applied + (int64_t)(nb + 1) * K + 1);
        if (rc != LPR_OK_OPTIMAL) return rc;
        if (t->log_cap != log_before || t->log != log_ptr) {
            rc = ov_set_log(t, overlap ? (int)(step & 1) : 0);
            if (rc != LPR_OK_OPTIMAL) return rc;
        }
        // One or two launches per K pivots: plain launches keep the device busy (measured: a
        // captured graph is no faster here, and capturing one costs milliseconds per solve call).
        if (timed) {  // (sized for a full batch at once: creating an event costs ~10 us)
            while ((int)t->ev.size() < 2 * (nb > nlaunch ? nb : nlaunch) + 2) {
                hipEvent_t ev;
                LPR_HIP(hipEventCreate(&ev));
                t->ev.push_back(ev);
            }
        }
        for (int k = 0; k < nb; ++k, ++step) {
            const int lp = (int)(step & 1);
            const bool tk = timed && (k % tstride == 0);
            if (two_streams) {
                rc = ov2_launch_step(t, K, tr, lp, hflags, tk ? t->ev[2 * k] : nullptr,
                                     tk ? t->ev[2 * k + 1] : nullptr);
                if (rc != LPR_OK_OPTIMAL) return rc;
                continue;
            }
            if (!overlap) ov_launch_heads(t, K, hflags);
            if (tk) LPR_HIP(hipEventRecord(t->ev[2 * k], s));
            if (overlap) ov_launch_step(t, K, tr, lp);
            else ov_launch_sweep(t, tr);
            if (tk) LPR_HIP(hipEventRecord(t->ev[2 * k + 1], s));
        }
        if (two_streams) {
            rc = ov2_join(t);
            if (rc != LPR_OK_OPTIMAL) return rc;
        }
        if (timed) LPR_HIP(hipEventRecord(t->ev[2 * nb], s));  // closes the last step's window
        LPR_HIP(hipGetLastError());
        rc = ov_poll(t, overlap ? (int)(step & 1) : 0, &pl);
        if (rc != LPR_OK_OPTIMAL) return rc;
        if (pl.error || pl.status == LPR_DEVICE_ERROR) {
            // the staging state is unusable (a head group never arrived): the handle is poisoned
            t->poisoned = true;
            set_error("overlapped pivot loop: a hand-off between the loop heads timed out; the "
                      "tableau handle is no longer usable");
            return LPR_DEVICE_ERROR;
        }
        status = pl.status;
        if (status == kRunning && pl.pending != kRunning && pl.kdone == 0)
            status = pl.pending;  // the end was found and nothing is left to sweep
        if (timed) {
            // steps of this batch sweep full blocks in order: first (only in the very first batch
            // of an overlapped call) the step that sweeps nothing, then full blocks, then at most
            // one partial block and idle steps
            const int64_t full_now = (pl.applied - start_iter) / K - swept_full;
            const int first = (step == nb && overlap) ? 1 : 0;
            for (int k = 0; k < nb && k < first + full_now; k += tstride) {
                if (k < first) continue;  // (the sampled steps are the multiples of tstride)
                float ms = 0.f;
                LPR_HIP(hipEventElapsedTime(&ms, t->ev[2 * k], t->ev[2 * k + 1]));
                t->timed_total_ms += ms;
                t->timed_launches += 1;
                // start of this step to the start of the next SAMPLED one (or the closing event):
                // that many steps, all of them full as long as they lie before first + full_now
                const int kn = (k + tstride < nb) ? k + tstride : nb;
                if (kn <= first + full_now) {
                    float st = 0.f;
                    LPR_HIP(hipEventElapsedTime(&st, t->ev[2 * k], t->ev[2 * kn]));
                    t->timed_step_ms += st;
                    t->timed_steps += kn - k;
                }
            }
            swept_full += full_now;
        }
        idle_batches = (pl.applied == applied) ? idle_batches + 1 : 0;
        applied = pl.applied;
        if (status == kRunning && idle_batches >= 3) {
            set_error("overlapped pivot loop made no progress (device status still running)");
            return LPR_DEVICE_ERROR;
        }
    }
    ov_adopt_buffer(t, pl.cur);
    t->total_pivots = applied;
    res->status = status;
    res->block = K;
    res->pivots = applied - start_iter;
    res->total_pivots = applied;
    res->z = pl.z[applied & 1];  // T[0, cols-1]: the RHS column the heads carry, entry 0
    return status;
}

}  // namespace lpr

using namespace lpr;

#define LPR_LIVE(t)                                                              \
    do {                                                                         \
        if (!(t) || !(t)->eng) {                                                 \
            set_error("tableau handle is null or its engine has been closed");   \
            return LPR_BAD_ARGUMENT;                                             \
        }                                                                        \
        if ((t)->poisoned) {                                                     \
            set_error("tableau handle is unusable: an earlier solve ended in a " \
                      "device-side hand-off timeout");                           \
            return LPR_DEVICE_ERROR;                                             \
        }                                                                        \
    } while (0)

extern "C" {

int lpr_abi_version(void) { return LPR_ABI_VERSION; }

const char* lpr_last_error(void) { return get_error(); }

int lpr_engine_open(int device, lpr_engine** out) {
    if (!out) return LPR_BAD_ARGUMENT;
    *out = nullptr;
    int count = 0;
    hipError_t err = hipGetDeviceCount(&count);
    if (err != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s); this engine has no CPU fallback",
                  err == hipSuccess ? "device count is 0" : hipGetErrorString(err));
        return LPR_DEVICE_ERROR;
    }
    if (device < 0 || device >= count) {
        set_error("device %d out of range (0..%d)", device, count - 1);
        return LPR_BAD_ARGUMENT;
    }
    LPR_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    LPR_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this engine is built for gfx950 only", device,
                  prop.gcnArchName);
        return LPR_DEVICE_ERROR;
    }
    lpr_engine* e = new (std::nothrow) lpr_engine();
    if (!e) return LPR_OUT_OF_MEMORY;
    e->device = device;
    e->num_cus = prop.multiProcessorCount;
    std::snprintf(e->arch, sizeof e->arch, "%s", prop.gcnArchName);
    hipError_t serr = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (serr != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(serr));
        delete e;
        return LPR_DEVICE_ERROR;
    }
    *out = e;
    return LPR_OK_OPTIMAL;
}

int lpr_engine_close(lpr_engine* e) {
    if (!e) return LPR_BAD_ARGUMENT;
    hipSetDevice(e->device);
    for (lpr_tableau* t : e->live) {  // orphan what the caller forgot to destroy
        release_device(t);
        t->eng = nullptr;
    }
    e->live.clear();
    for (lpr_revised* r : e->live_rev) rev_orphan(r);
    e->live_rev.clear();
    for (lpr_bb* b : e->live_bb) bb_orphan(b);
    e->live_bb.clear();
    for (lpr_sens* q : e->live_sens) sens_orphan(q);
    e->live_sens.clear();
    for (lpr_comm* c : e->live_comm) comm_orphan(c);
    e->live_comm.clear();
    if (e->stream) {
        hipStreamSynchronize(e->stream);
        hipStreamDestroy(e->stream);
    }
    delete e;
    return LPR_OK_OPTIMAL;
}

int lpr_engine_sync(lpr_engine* e) {
    if (!e) return LPR_BAD_ARGUMENT;
    LPR_HIP(hipStreamSynchronize(e->stream));
    return LPR_OK_OPTIMAL;
}

uint64_t lpr_engine_stream(lpr_engine* e) { return e ? (uint64_t)(uintptr_t)e->stream : 0; }

// ------------------------------------------------------------------------------ tableau

int lpr_tableau_from_lp(lpr_engine* e, int n, int m, const double* objective, const double* A,
                        int lda, const int32_t* ncoef, const int8_t* relation, const double* rhs,
                        int is_max, lpr_tableau** out) {
    if (!e || !out || n < 0 || m < 0 || (n > 0 && !objective) || (m > 0 && (!rhs || lda < n)) ||
        (m > 0 && n > 0 && !A)) {
        set_error("lpr_tableau_from_lp: bad arguments (n=%d m=%d lda=%d)", n, m, lda);
        return LPR_BAD_ARGUMENT;
    }
    lpr_tableau* t = nullptr;
    int rc = alloc_tableau(e, m + 1, n + m + 1, &t);
    if (rc != LPR_OK_OPTIMAL) return rc;
    hipStream_t s = e->stream;
    double *d_obj = nullptr, *d_A = nullptr, *d_rhs = nullptr;
    int32_t* d_ncoef = nullptr;
    int8_t* d_rel = nullptr;
    hipError_t err = hipSuccess;
    auto chk = [&](hipError_t x) { if (err == hipSuccess) err = x; };
    if (n > 0) {
        chk(hipMalloc(&d_obj, (size_t)n * sizeof(double)));
        if (err == hipSuccess)
            chk(hipMemcpyAsync(d_obj, objective, (size_t)n * sizeof(double),
                               hipMemcpyHostToDevice, s));
    }
    if (m > 0) {
        if (n > 0) {
            chk(hipMalloc(&d_A, (size_t)m * n * sizeof(double)));
            if (err == hipSuccess)
                chk(hipMemcpy2DAsync(d_A, (size_t)n * sizeof(double), A,
                                     (size_t)lda * sizeof(double), (size_t)n * sizeof(double), m,
                                     hipMemcpyHostToDevice, s));
        }
        chk(hipMalloc(&d_rhs, (size_t)m * sizeof(double)));
        if (err == hipSuccess)
            chk(hipMemcpyAsync(d_rhs, rhs, (size_t)m * sizeof(double), hipMemcpyHostToDevice, s));
        if (ncoef) {
            chk(hipMalloc(&d_ncoef, (size_t)m * sizeof(int32_t)));
            if (err == hipSuccess)
                chk(hipMemcpyAsync(d_ncoef, ncoef, (size_t)m * sizeof(int32_t),
                                   hipMemcpyHostToDevice, s));
        }
        if (relation) {
            chk(hipMalloc(&d_rel, (size_t)m));
            if (err == hipSuccess)
                chk(hipMemcpyAsync(d_rel, relation, (size_t)m, hipMemcpyHostToDevice, s));
        }
    }
    if (err == hipSuccess) {
        launch_build(t, n, m, d_obj, d_A, n, d_ncoef, d_rel, d_rhs, is_max);
        chk(hipGetLastError());
        chk(hipStreamSynchronize(s));  // inputs are borrowed for this call only
    }
    hipFree(d_obj);
    hipFree(d_A);
    hipFree(d_rhs);
    hipFree(d_ncoef);
    hipFree(d_rel);
    if (err != hipSuccess) {
        set_error("lpr_tableau_from_lp: %s", hipGetErrorString(err));
        lpr_tableau_destroy(t);
        return LPR_DEVICE_ERROR;
    }
    *out = t;
    return LPR_OK_OPTIMAL;
}

int lpr_tableau_create(lpr_engine* e, int rows, int cols, const double* rowmajor,
                       const int32_t* basis, lpr_tableau** out) {
    if (!e || !out || !rowmajor) {
        set_error("lpr_tableau_create: null argument");
        return LPR_BAD_ARGUMENT;
    }
    lpr_tableau* t = nullptr;
    int rc = alloc_tableau(e, rows, cols, &t);
    if (rc != LPR_OK_OPTIMAL) return rc;
    hipStream_t s = e->stream;
    hipError_t err = hipMemcpy2DAsync(t->T, (size_t)t->ld * sizeof(double), rowmajor,
                                      (size_t)cols * sizeof(double), (size_t)cols * sizeof(double),
                                      rows, hipMemcpyHostToDevice, s);
    if (err == hipSuccess && rows > 1) {
        if (basis)
            err = hipMemcpyAsync(t->basis, basis, (size_t)(rows - 1) * sizeof(int32_t),
                                 hipMemcpyHostToDevice, s);
        else
            err = hipMemsetAsync(t->basis, 0xff, (size_t)(rows - 1) * sizeof(int32_t), s);
    }
    if (err == hipSuccess) err = hipStreamSynchronize(s);
    if (err != hipSuccess) {
        set_error("lpr_tableau_create: %s", hipGetErrorString(err));
        lpr_tableau_destroy(t);
        return LPR_DEVICE_ERROR;
    }
    *out = t;
    return LPR_OK_OPTIMAL;
}

int lpr_tableau_synthetic(lpr_engine* e, int m, int n, uint64_t seed, lpr_tableau** out) {
    if (!e || !out || m < 1 || n < 1) {
        set_error("lpr_tableau_synthetic: bad arguments (m=%d n=%d)", m, n);
        return LPR_BAD_ARGUMENT;
    }
    lpr_tableau* t = nullptr;
    int rc = alloc_tableau(e, m + 1, n + m + 1, &t);
    if (rc != LPR_OK_OPTIMAL) return rc;
    launch_synthetic(t, m, n, seed);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err != hipSuccess) {
        set_error("lpr_tableau_synthetic: %s", hipGetErrorString(err));
        lpr_tableau_destroy(t);
        return LPR_DEVICE_ERROR;
    }
    *out = t;
    return LPR_OK_OPTIMAL;
}

int lpr_tableau_destroy(lpr_tableau* t) {
    if (!t) return LPR_BAD_ARGUMENT;
    if (t->eng) {  // still attached: release device memory, detach from the engine
        release_device(t);
        auto& lv = t->eng->live;
        for (size_t k = 0; k < lv.size(); ++k)
            if (lv[k] == t) {
                lv.erase(lv.begin() + k);
                break;
            }
    }
    delete t;
    return LPR_OK_OPTIMAL;
}

int lpr_tableau_shape(const lpr_tableau* t, int* rows, int* cols, int* ld) {
    if (!t) return LPR_BAD_ARGUMENT;
    if (rows) *rows = t->rows;
    if (cols) *cols = t->cols;
    if (ld) *ld = t->ld;
    return LPR_OK_OPTIMAL;
}

// ------------------------------------------------------------------------------ solve

int lpr_primal_solve(lpr_tableau* t, const lpr_solve_opts* opts, lpr_solve_result* res) {
    LPR_LIVE(t);
    if (!t || !res) {
        set_error("lpr_primal_solve: null argument");
        return LPR_BAD_ARGUMENT;
    }
    lpr_solve_opts o;
    std::memset(&o, 0, sizeof o);
    if (opts) o = *opts;
    const int hflags = (o.variant >> 16) & 255;  // loop-head placement / diagnostics (K-pivot paths)
    o.variant &= 0xffff;                       // path + tile
    lpr_engine* e = t->eng;
    hipStream_t s = e->stream;
    LPR_HIP(hipSetDevice(e->device));

    if (use_small(t, o)) return solve_small(t, o, res);
    if (use_fused(t, o)) return solve_fused(t, o, res);
    {
        // variant 0x60tr: in-place blocked path; 0x50tr or none: the overlapped path
        const int K = block_size(t, o);
        if (K > 1 && (o.variant & 0xff00) == 0x6000) return solve_blocked(t, o, K, res);
        // default / 0x50tr: sweep out of place with the next block's heads inside the same launch;
        // 0x40tr: all heads of a block in one persistent launch, then the sweep in place (also the
        // fallback when the second tableau buffer cannot be allocated)
        if (K > 1) {
            // measured (tools/size_sweep.sh): up to ~80 MB the heads-then-sweep form is faster
            // (9.1-9.6 us per pivot), above it hiding the sweep behind the next heads wins
            const size_t tbytes = (size_t)t->rows * t->ld * sizeof(double);
            // above kOverlapBytes the default is the two-stream form of the overlap (0x30tr)
            const bool big = (o.variant & 0xff00) == 0 && tbytes > kOverlapBytes;
            const bool two_streams = (o.variant & 0xff00) == 0x3000 || big;
            bool overlap = two_streams || (o.variant & 0xff00) == 0x5000;
            const int tr = (o.variant & 0xff00) ? (o.variant & 0xff) : 8;
            if (overlap && ov_ensure(t, true) == LPR_OUT_OF_MEMORY) overlap = false;
            return solve_overlapped(t, o, K, tr, overlap, two_streams && overlap, hflags, res);
        }
    }
    const int variant = (o.variant > 0 && o.variant < 0x7000) ? o.variant - 1 : default_variant(t);
    int batch = o.batch > 0 ? o.batch : default_batch(t);
    const bool timed = o.time_kernels != 0;
    const int64_t start_iter = t->total_pivots;
    const int64_t max_iter = o.max_pivots > 0 ? start_iter + o.max_pivots : 0;

    PivotState* hs = t->h_state;
    hs->status = kRunning;
    hs->iter = start_iter;
    hs->max_iter = max_iter;
    hs->log_cap = t->log_cap;
    // cur_r/cur_e/sweep keep their device values: upload everything except those via two copies
    LPR_HIP(hipMemcpyAsync(&t->state->status, &hs->status, sizeof(int32_t), hipMemcpyHostToDevice,
                           s));
    LPR_HIP(hipMemcpyAsync(&t->state->iter, &hs->iter, 3 * sizeof(int64_t),
                           hipMemcpyHostToDevice, s));

    // One slow pass (strided column gather) primes next_e / next_col / next_rhs; from then on every
    // k_update leaves them ready for the following k_pivot_head.
    launch_bootstrap(t);

    int status = kRunning;
    int64_t iter = start_iter;
    while (status == kRunning) {
        int nb = batch;
        if (timed && max_iter > 0) {
            const int64_t left = max_iter - iter;
            if (left < nb) nb = (int)left;  // may be 0: then only the closing select runs
        }
        int rc = ensure_log(t, iter + nb + 1);
        if (rc != LPR_OK_OPTIMAL) return rc;
        if (t->log_cap != hs->log_cap) {
            hs->log_cap = t->log_cap;
            LPR_HIP(hipMemcpyAsync(&t->state->log_cap, &hs->log_cap, sizeof(int64_t),
                                   hipMemcpyHostToDevice, s));
        }

        if (timed) {
            while ((int)t->ev.size() < 2 * nb) {
                hipEvent_t ev;
                LPR_HIP(hipEventCreate(&ev));
                t->ev.push_back(ev);
            }
            // Bracket every kTimeStride-th rank-1 update with events: a record costs a few
            // microseconds of stream time, which would otherwise be charged to every pivot.
            for (int k = 0; k < nb; ++k) {
                launch_pivot_head(t);
                const bool sample = (k % kTimeStride) == 0;
                if (sample) LPR_HIP(hipEventRecord(t->ev[2 * k], s));
                launch_update(t, variant, 1, 1);
                if (sample) LPR_HIP(hipEventRecord(t->ev[2 * k + 1], s));
            }
            if (nb == 0 || (max_iter > 0 && iter + nb >= max_iter))
                launch_pivot_head(t);  // closing loop head -> final status
        } else {
            if (!graph_valid(t, nb, variant)) {
                drop_graph(t);
                hipGraph_t g = nullptr;
                LPR_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int k = 0; k < nb; ++k) {
                    launch_pivot_head(t);
                    launch_update(t, variant, 1, 1);
                }
                LPR_HIP(hipStreamEndCapture(s, &g));
                hipError_t ierr = hipGraphInstantiate(&t->graph, g, nullptr, nullptr, 0);
                hipGraphDestroy(g);
                if (ierr != hipSuccess) {
                    t->graph = nullptr;
                    set_error("hipGraphInstantiate failed: %s", hipGetErrorString(ierr));
                    return LPR_DEVICE_ERROR;
                }
                graph_stamp(t, nb, variant);
            }
            LPR_HIP(hipGraphLaunch(t->graph, s));
        }
        LPR_HIP(hipGetLastError());
        LPR_HIP(hipMemcpyAsync(hs, t->state, sizeof(PivotState), hipMemcpyDeviceToHost, s));
        LPR_HIP(hipStreamSynchronize(s));
        const int64_t done = hs->iter - iter;
        if (timed) {
            for (int64_t k = 0; k < done && k < nb; k += kTimeStride) {  // launches that pivoted
                float ms = 0.f;
                LPR_HIP(hipEventElapsedTime(&ms, t->ev[2 * k], t->ev[2 * k + 1]));
                t->timed_total_ms += ms;
                t->timed_launches += 1;
            }
        }
        iter = hs->iter;
        status = hs->status;
        if (status == kRunning && done == 0 && nb > 0) {
            set_error("pivot loop made no progress (device status still running)");
            return LPR_DEVICE_ERROR;
        }
    }
    t->total_pivots = iter;
    res->status = status;
    res->block = 1;
    res->pivots = iter - start_iter;
    res->total_pivots = iter;
    double z = 0.0;
    LPR_HIP(hipMemcpyAsync(&z, t->T + (t->cols - 1), sizeof(double), hipMemcpyDeviceToHost, s));
    LPR_HIP(hipStreamSynchronize(s));
    res->z = z;
    return status;
}

static int read_scratch(lpr_tableau* t, int idx, int32_t* out) {
    hipStream_t s = t->eng->stream;
    LPR_HIP(hipGetLastError());
    LPR_HIP(hipMemcpyAsync(t->h_scratch_i, t->scratch_i, 4 * sizeof(int32_t),
                           hipMemcpyDeviceToHost, s));
    LPR_HIP(hipStreamSynchronize(s));
    *out = t->h_scratch_i[idx];
    return LPR_OK_OPTIMAL;
}

int lpr_select_entering(lpr_tableau* t, int32_t* col) {
    LPR_LIVE(t);
    if (!t || !col) return LPR_BAD_ARGUMENT;
    LPR_HIP(hipSetDevice(t->eng->device));
    launch_select(t, kSelEnter, -1, -1, t->scratch_i);
    return read_scratch(t, 0, col);
}

int lpr_select_leaving(lpr_tableau* t, int32_t col, int32_t* row) {
    LPR_LIVE(t);
    if (!t || !row || col < 0 || col >= t->cols - 1) {
        set_error("lpr_select_leaving: column %d out of range", col);
        return LPR_BAD_ARGUMENT;
    }
    LPR_HIP(hipSetDevice(t->eng->device));
    launch_select(t, kSelLeave, col, -1, t->scratch_i);
    return read_scratch(t, 1, row);
}

int lpr_pivot(lpr_tableau* t, int32_t row, int32_t col) {
    LPR_LIVE(t);
    if (!t || row < 1 || row >= t->rows || col < 0 || col >= t->cols) {
        set_error("lpr_pivot: (%d, %d) out of range", row, col);
        return LPR_BAD_ARGUMENT;
    }
    LPR_HIP(hipSetDevice(t->eng->device));
    int rc = ensure_log(t, t->total_pivots + 1);
    if (rc != LPR_OK_OPTIMAL) return rc;
    hipStream_t s = t->eng->stream;
    PivotState* hs = t->h_state;
    hs->iter = t->total_pivots;
    hs->max_iter = 0;
    hs->log_cap = t->log_cap;
    LPR_HIP(hipMemcpyAsync(&t->state->iter, &hs->iter, 3 * sizeof(int64_t),
                           hipMemcpyHostToDevice, s));
    launch_select(t, kSelCommit, col, row, nullptr);
    launch_update(t, default_variant(t), 0, 0);
    LPR_HIP(hipGetLastError());
    LPR_HIP(hipStreamSynchronize(s));
    t->total_pivots += 1;
    return LPR_OK_OPTIMAL;
}

int lpr_extract_solution(lpr_tableau* t, int n, double* x, double* z) {
    LPR_LIVE(t);
    if (!t || n < 0 || n > t->cols - 1 || (n > 0 && !x)) {
        set_error("lpr_extract_solution: bad arguments (n=%d)", n);
        return LPR_BAD_ARGUMENT;
    }
    LPR_HIP(hipSetDevice(t->eng->device));
    hipStream_t s = t->eng->stream;
    if (n > 0) {
        if (t->xbuf_n < n) {
            hipFree(t->xbuf);
            t->xbuf = nullptr;
            t->xbuf_n = 0;
            LPR_HIP(hipMalloc(&t->xbuf, (size_t)n * sizeof(double)));
            t->xbuf_n = n;
        }
        launch_extract(t, n, t->xbuf);
        LPR_HIP(hipGetLastError());
        LPR_HIP(hipMemcpyAsync(x, t->xbuf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    if (z)
        LPR_HIP(hipMemcpyAsync(z, t->T + (t->cols - 1), sizeof(double), hipMemcpyDeviceToHost,
                               s));
    LPR_HIP(hipStreamSynchronize(s));
    return LPR_OK_OPTIMAL;
}

int lpr_tableau_read_block(lpr_tableau* t, int row0, int nrows, int col0, int ncols,
                           double* out) {
    LPR_LIVE(t);
    if (!t || !out || row0 < 0 || col0 < 0 || nrows < 0 || ncols < 0 ||
        row0 + nrows > t->rows || col0 + ncols > t->cols) {
        set_error("lpr_tableau_read_block: block out of range");
        return LPR_BAD_ARGUMENT;
    }
    if (nrows == 0 || ncols == 0) return LPR_OK_OPTIMAL;
    LPR_HIP(hipSetDevice(t->eng->device));
    hipStream_t s = t->eng->stream;
    LPR_HIP(hipMemcpy2DAsync(out, (size_t)ncols * sizeof(double),
                             t->T + (size_t)row0 * t->ld + col0, (size_t)t->ld * sizeof(double),
                             (size_t)ncols * sizeof(double), nrows, hipMemcpyDeviceToHost, s));
    LPR_HIP(hipStreamSynchronize(s));
    return LPR_OK_OPTIMAL;
}

int lpr_tableau_read(lpr_tableau* t, double* rowmajor_out) {
    LPR_LIVE(t);
    if (!t) return LPR_BAD_ARGUMENT;
    return lpr_tableau_read_block(t, 0, t->rows, 0, t->cols, rowmajor_out);
}

int lpr_basis_read(lpr_tableau* t, int32_t* basis_out) {
    LPR_LIVE(t);
    if (!t || !basis_out) return LPR_BAD_ARGUMENT;
    if (t->rows <= 1) return LPR_OK_OPTIMAL;
    LPR_HIP(hipSetDevice(t->eng->device));
    hipStream_t s = t->eng->stream;
    LPR_HIP(hipMemcpyAsync(basis_out, t->basis, (size_t)(t->rows - 1) * sizeof(int32_t),
                           hipMemcpyDeviceToHost, s));
    LPR_HIP(hipStreamSynchronize(s));
    return LPR_OK_OPTIMAL;
}

int lpr_pivot_log_read(lpr_tableau* t, int32_t* rows_out, int32_t* cols_out, int64_t cap,
                       int64_t* count) {
    LPR_LIVE(t);
    if (!t || !count || cap < 0) return LPR_BAD_ARGUMENT;
    int64_t n = t->total_pivots < t->log_cap ? t->total_pivots : t->log_cap;
    if (n > cap) n = cap;
    *count = n;
    if (n == 0) return LPR_OK_OPTIMAL;
    LPR_HIP(hipSetDevice(t->eng->device));
    std::vector<int32_t> tmp((size_t)n * 2);
    LPR_HIP(hipMemcpy(tmp.data(), t->log, (size_t)n * 2 * sizeof(int32_t),
                      hipMemcpyDeviceToHost));
    for (int64_t k = 0; k < n; ++k) {
        if (rows_out) rows_out[k] = tmp[2 * k];
        if (cols_out) cols_out[k] = tmp[2 * k + 1];
    }
    return LPR_OK_OPTIMAL;
}

int lpr_tableau_kernel_stats(lpr_tableau* t, int64_t* launches, double* total_ms,
                             double* avg_ms) {
    LPR_LIVE(t);
    if (!t) return LPR_BAD_ARGUMENT;
    if (launches) *launches = t->timed_launches;
    if (total_ms) *total_ms = t->timed_total_ms;
    if (avg_ms) *avg_ms = t->timed_launches ? t->timed_total_ms / t->timed_launches : 0.0;
    return LPR_OK_OPTIMAL;
}

int lpr_tableau_step_stats(lpr_tableau* t, int64_t* steps, double* total_ms) {
    LPR_LIVE(t);
    if (steps) *steps = t->timed_steps;
    if (total_ms) *total_ms = t->timed_step_ms;
    return LPR_OK_OPTIMAL;
}

int lpr_debug_head_stamps(lpr_tableau* t, uint64_t* out, int64_t cap, int64_t* count) {
    LPR_LIVE(t);
    if (!count || cap < 0) return LPR_BAD_ARGUMENT;
    LPR_HIP(hipSetDevice(t->eng->device));
    return ov_read_stamps(t, out, cap, count);
}

}  // extern "C"
