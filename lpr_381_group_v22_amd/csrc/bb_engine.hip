// bb_engine.hip -- C ABI of the Branch & Bound path (include/lpr_engine.h, lpr_bb_*).
//
// The tree logic of BranchAndBound.ExecuteBranchAndBound (IntegerProgramming/
// BranchBoundSimplexSolver.cs:1006-1233) -- a stack, a few comparisons per node -- runs on the
// host; every tableau operation (RoundTableau, IdentifyBasicVariables, AddConstraint, the dual /
// primal pivots of DoDualSimplex, the decision-value scans) runs on the device, batched over the
// children that are evaluated together.  Node tableaux never leave HBM.  No CPU fallback.
#include "bb_common.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#include <cmath>
#include <new>

#pragma clang fp contract(off)

struct lpr_comm;
namespace lpr {
// comm_engine.hip
int comm_rank(const lpr_comm* c);
int comm_world(const lpr_comm* c);
int comm_all_reduce_max(lpr_comm* c, double* v, int n);
int comm_all_gather(lpr_comm* c, const void* send, void* recv, int bytes);

// bb_kernels.hip
void bb_launch_copy_in(lpr_bb* b, const double* src, int src_ld, int rows, int cols, double* dst);
void bb_launch_round(lpr_bb* b, int nslots, int rows_max, int clean);
void bb_launch_node_info(lpr_bb* b, int count);
void bb_launch_add_constraint(lpr_bb* b, int nslots, int nparents, int rows_max, int cols_max,
                              bool side, bool inplace);
void bb_launch_pivot_step(lpr_bb* b, int nslots, int rows_max, int cols_max, int step_no);
void bb_launch_finish(lpr_bb* b, int nslots, int cols_max);
void bb_launch_gather_info(lpr_bb* b, int count);

// ---- .NET Framework rounding on the host (IsInteger :595-599 works on n values per node) ----
static double dn_round_int(double x) {  // Math.Round(double), COMDouble::Round
    if (std::isnan(x) || std::isinf(x)) return x;
    if (std::fabs(x) < 9.2e18 && x == (double)((long long)x)) return x;
    const double t = x + 0.5;
    double f = std::floor(t);
    if (f == t && std::fmod(t, 2.0) != 0) f -= 1.0;
    return std::copysign(f, x);
}
// `(int)d` of the C# (:870-871) as the x64 JIT of .NET Framework 4.7.2 compiles it (cvttsd2si): a
// value outside int's range, or NaN, gives 0x80000000.  Spelt out: in C++ that cast is undefined.
static int dn_to_int32(double x) {
    if (!(x > -2147483649.0 && x < 2147483648.0)) return INT32_MIN;
    return (int)x;
}
static double dn_round4(double x) {  // Math.Round(double, 4)
    if (std::fabs(x) < 1e16) {
        x = x * 10000.0;
        x = dn_round_int(x);
        x = x / 10000.0;
    }
    return x;
}
static bool is_integer(double v) {  // :595-599
    const double r = dn_round4(v);
    return std::fabs(r - dn_round_int(r)) <= 1e-6;
}

static void bb_release_device(lpr_bb* b) {
    hipSetDevice(b->eng->device);
    if (b->eng->stream) hipStreamSynchronize(b->eng->stream);
    for (double* p : b->all_bufs) hipFree(p);
    b->all_bufs.clear();
    b->free_bufs.clear();
    b->nodes.clear();
    hipFree(b->d_slots); hipFree(b->rowbuf); hipFree(b->colbuf); hipFree(b->bflag);
    hipFree(b->bkey); hipFree(b->blist); hipFree(b->bcount); hipFree(b->trace); hipFree(b->info);
    hipFree(b->rowlist);
    hipFree(b->touched);
    b->rowlist = nullptr;
    b->touched = nullptr;
    hipFree(b->d_running);
    if (b->h_slots) hipHostFree(b->h_slots);
    if (b->h_info) hipHostFree(b->h_info);
    if (b->h_running) hipHostFree(b->h_running);
    b->d_slots = b->h_slots = nullptr;
    b->rowbuf = b->colbuf = b->info = b->h_info = nullptr;
    b->bflag = b->bkey = b->blist = b->bcount = b->trace = b->d_running = b->h_running = nullptr;
    b->slot_cap = 0;
}

void bb_orphan(lpr_bb* b) {
    bb_release_device(b);
    b->eng = nullptr;
}

static double bb_now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int bb_alloc_buf(lpr_bb* b, double** out) {
    if (!b->free_bufs.empty()) {
        *out = b->free_bufs.back();
        b->free_bufs.pop_back();
        return LPR_OK_OPTIMAL;
    }
    const double t0 = bb_now();
    struct Acc {
        lpr_bb* b; double t0;
        ~Acc() { b->prof.alloc += bb_now() - t0; b->prof.mallocs += 1; }
    } acc{b, t0};
    // grow the pool by a slab of buffers at a time (hundreds of children per level: one
    // hipMalloc each would cost more than solving them)
    // ... and every slab as large as everything allocated before it (128 MB, 128, 256, 512 MB, ...):
    // a tree that doubles per level then needs one hipMalloc per level, not one per 128 MB
    // (51 of them cost 0.84 of the 15.7 ms of the bench tree)
    const size_t bytes = b->buf_elems * sizeof(double);
    size_t slab = (size_t)128 << 20;
    if (b->pool_bytes > slab) slab = b->pool_bytes;
    if (slab > ((size_t)8 << 30)) slab = (size_t)8 << 30;
    size_t per = slab / bytes;
    if (per < 4) per = 4;
    double* p = nullptr;
    hipError_t err = hipMalloc(&p, per * bytes);
    if (err != hipSuccess) {
        per = 1;
        err = hipMalloc(&p, bytes);
    }
    if (err != hipSuccess) {
        set_error("B&B node buffer allocation (%zu bytes) failed: %s", bytes,
                  hipGetErrorString(err));
        return err == hipErrorOutOfMemory ? LPR_OUT_OF_MEMORY : LPR_DEVICE_ERROR;
    }
    b->all_bufs.push_back(p);  // slab base: what hipFree gets at destroy
    b->pool_bytes += per * bytes;
    for (size_t k = 1; k < per; ++k) b->free_bufs.push_back(p + k * b->buf_elems);
    *out = p;
    return LPR_OK_OPTIMAL;
}

static int bb_ensure_slots(lpr_bb* b, int need) {
    if (need <= b->slot_cap) return LPR_OK_OPTIMAL;
    struct Acc {
        lpr_bb* b; double t0;
        ~Acc() { b->prof.slots += bb_now() - t0; }
    } acc{b, bb_now()};
    int cap = b->slot_cap ? b->slot_cap : 2;
    while (cap < need) cap *= 2;
    LPR_HIP(hipStreamSynchronize(b->eng->stream));
    hipFree(b->d_slots); hipFree(b->rowbuf); hipFree(b->colbuf); hipFree(b->bflag);
    hipFree(b->bkey); hipFree(b->blist); hipFree(b->bcount); hipFree(b->trace); hipFree(b->info);
    hipFree(b->rowlist);
    hipFree(b->touched);
    b->rowlist = nullptr;
    b->touched = nullptr;
    if (b->h_slots) hipHostFree(b->h_slots);
    if (b->h_info) hipHostFree(b->h_info);
    b->d_slots = b->h_slots = nullptr;
    b->rowbuf = b->colbuf = b->info = b->h_info = nullptr;
    b->bflag = b->bkey = b->blist = b->bcount = b->trace = nullptr;
    b->slot_cap = 0;
    b->trace_cap = 4 * (b->rows_cap + b->ld) + 64;  // generous: pivots per child LP
    const size_t S = (size_t)cap;
    LPR_HIP(hipMalloc(&b->d_slots, S * sizeof(BBSlot)));
    LPR_HIP(hipHostMalloc(&b->h_slots, S * sizeof(BBSlot)));
    LPR_HIP(hipMalloc(&b->rowbuf, S * b->ld * sizeof(double)));
    LPR_HIP(hipMalloc(&b->colbuf, S * b->rows_cap * sizeof(double)));
    LPR_HIP(hipMalloc(&b->bflag, S * b->ld * sizeof(int32_t)));
    LPR_HIP(hipMalloc(&b->bkey, S * b->ld * sizeof(int32_t)));
    LPR_HIP(hipMalloc(&b->blist, S * b->ld * sizeof(int32_t)));
    LPR_HIP(hipMalloc(&b->bcount, S * sizeof(int32_t)));
    LPR_HIP(hipMalloc(&b->rowlist, S * b->rows_cap * sizeof(int32_t)));
    LPR_HIP(hipMalloc(&b->touched, S * align_up(b->rows_cap, 16)));
    LPR_HIP(hipMalloc(&b->trace, S * b->trace_cap * 3 * sizeof(int32_t)));
    LPR_HIP(hipMalloc(&b->info, S * (b->nvars + 1) * sizeof(double)));
    LPR_HIP(hipHostMalloc(&b->h_info, S * (b->nvars + 1) * sizeof(double)));
    if (!b->d_running) {
        LPR_HIP(hipMalloc(&b->d_running, 2 * sizeof(int32_t)));
        LPR_HIP(hipHostMalloc(&b->h_running, 2 * sizeof(int32_t)));
    }
    b->slot_cap = cap;
    return LPR_OK_OPTIMAL;
}

static int bb_new_node(lpr_bb* b, double* T, int rows, int cols, int depth) {
    lpr_bb::Node nd;
    nd.T = T;
    nd.rows = rows;
    nd.cols = cols;
    nd.depth = depth;
    nd.live = true;
    b->nodes.push_back(nd);
    return (int)b->nodes.size() - 1;
}

// RoundAllTableaux on pop (:1047) + GetObjective / decision values for `count` nodes.
// z_out[count], vals_out[count * nvars].
static int bb_node_info(lpr_bb* b, const int32_t* ids, int count, double* z_out,
                        double* vals_out) {
    if (count <= 0) return LPR_OK_OPTIMAL;
    int rc = bb_ensure_slots(b, count);
    if (rc != LPR_OK_OPTIMAL) return rc;
    hipStream_t st = b->eng->stream;
    int rows_max = 0;
    for (int k = 0; k < count; ++k) {
        const int id = ids[k];
        if (id < 0 || id >= (int)b->nodes.size() || !b->nodes[id].live) {
            set_error("lpr_bb: node %d is not live", id);
            return LPR_BAD_ARGUMENT;
        }
        BBSlot& s = b->h_slots[k];
        std::memset(&s, 0, sizeof s);
        s.cur = b->nodes[id].T;
        s.rows = b->nodes[id].rows;
        s.cols = b->nodes[id].cols;
        rows_max = s.rows > rows_max ? s.rows : rows_max;
    }
    LPR_HIP(hipMemcpyAsync(b->d_slots, b->h_slots, (size_t)count * sizeof(BBSlot),
                           hipMemcpyHostToDevice, st));
    // currentTableaux = RoundAllTableaux(...) :1047.  A node stored by bb_expand has been rounded
    // once already (:1124 / :1187); below 1e11 rounding is idempotent, so the pass is skipped unless
    // a node of the batch holds a larger (or non-finite) entry, or its history is unknown (the root)
    bool any_big = false, all_side = true;
    for (int k = 0; k < count; ++k) {
        any_big = any_big || b->nodes[ids[k]].big;
        all_side = all_side && b->nodes[ids[k]].side;
    }
    if (!any_big && all_side) {
        bb_launch_gather_info(b, count);  // scored by k_bb_finish when they were solved
    } else {
        if (any_big) {
            bb_launch_round(b, count, rows_max, 0);
            for (int k = 0; k < count; ++k) b->nodes[ids[k]].side = false;  // (re-rounded: stale)
        }
        bb_launch_node_info(b, count);
    }
    LPR_HIP(hipGetLastError());
    LPR_HIP(hipMemcpyAsync(b->h_info, b->info, (size_t)count * (b->nvars + 1) * sizeof(double),
                           hipMemcpyDeviceToHost, st));
    LPR_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < count; ++k) {
        const double* p = b->h_info + (size_t)k * (b->nvars + 1);
        z_out[k] = p[0];
        for (int i = 0; i < b->nvars; ++i) vals_out[(size_t)k * b->nvars + i] = p[1 + i];
    }
    return LPR_OK_OPTIMAL;
}

// AddConstraint (:694-803) + DoDualSimplex (:289-468) + RoundAllTableaux (:1124/:1187) for `count`
// children, all in one batch.  kind: 0 = lower ("<=", type 0), 1 = upper (">=", type 1).
// status_out: kBBSolved / kBBInfeasible / kBBFailed; child_ids_out: node id or -1.
// What lpr_bb_expand_traced keeps of every child besides its outcome: every tableau of
// DoDualSimplex's list (`tableaux`, :292,:341,:388), compact rows x cols, in order.
struct BBKeep {
    std::vector<std::vector<double>> tabs;  // [child] -> concatenated tableaux
    std::vector<int> ntab;                  // [child] -> how many
};

static int bb_keep_copy(lpr_bb* b, const BBSlot& s, std::vector<double>* dst) {
    const size_t n = (size_t)s.rows * s.cols;
    const size_t at = dst->size();
    dst->resize(at + n);
    LPR_HIP(hipMemcpy2DAsync(dst->data() + at, (size_t)s.cols * sizeof(double), s.cur,
                             (size_t)b->ld * sizeof(double), (size_t)s.cols * sizeof(double),
                             s.rows, hipMemcpyDeviceToHost, b->eng->stream));
    LPR_HIP(hipStreamSynchronize(b->eng->stream));
    return LPR_OK_OPTIMAL;
}

static int bb_expand(lpr_bb* b, int count, const int32_t* parent_ids, const int32_t* var,
                     const double* bound, const int32_t* kind, int32_t* child_ids_out,
                     int32_t* status_out, int32_t* pivots_out,
                     std::vector<int32_t>* trace_out /* triples per child, flattened */,
                     std::vector<int32_t>* trace_off, BBKeep* keep = nullptr,
                     bool consume = false) {
    // consume: the caller releases every parent right after this call (the tree drivers do).  The
    // SECOND of two consecutive children of one parent then takes the parent's buffer over instead
    // of copying it (k_bb_child_inplace): half of k_bb_child_init's traffic, the largest single
    // item of a level.  Only for parents that carry their own scan and were rounded below 1e11
    // (`side`, not `big`): their stored rows are what a child starts from, up to the sign of zeros.
    if (count <= 0) return LPR_OK_OPTIMAL;
    int rc = bb_ensure_slots(b, count);
    if (rc != LPR_OK_OPTIMAL) return rc;
    hipStream_t st = b->eng->stream;
    int rows_max = 0, cols_max = 0;
    // IdentifyBasicVariables reads the PARENT only: one scan per distinct parent of the batch
    std::vector<std::pair<int, int>> seen;  // (parent id, scan row); parents come in runs
    int nparents = 0;
    for (int k = 0; k < count; ++k) {
        const int pid = parent_ids[k];
        if (pid < 0 || pid >= (int)b->nodes.size() || !b->nodes[pid].live) {
            set_error("lpr_bb: parent node %d is not live", pid);
            return LPR_BAD_ARGUMENT;
        }
        const lpr_bb::Node& pn = b->nodes[pid];
        if (pn.rows + 1 > b->rows_cap || align_up(pn.cols + 1, kLdAlign) > b->ld) {
            set_error("lpr_bb: depth limit reached (max_depth=%d)", b->max_depth);
            return LPR_BB_NODE_CAP;
        }
        BBSlot& s = b->h_slots[k];
        std::memset(&s, 0, sizeof s);
        rc = bb_alloc_buf(b, &s.cur);
        if (rc != LPR_OK_OPTIMAL) return rc;
        rc = bb_alloc_buf(b, &s.nxt);
        if (rc != LPR_OK_OPTIMAL) return rc;
        s.rows = pn.rows + 1;
        s.cols = pn.cols + 1;
        s.state = kBBDual;
        s.reverse = kind[k] ? 1 : 0;
        s.var = var[k];
        s.crow = pn.rows;
        s.bound = bound[k];
        s.parent = pn.T;
        int u = -1;
        for (size_t q = seen.size(); q-- > 0 && seen.size() - q <= 4;)  // the last few parents
            if (seen[q].first == pid) { u = seen[q].second; break; }
        if (u < 0) {
            u = nparents++;
            seen.emplace_back(pid, u);
            b->h_slots[u].rep = k;  // (slot u <= k has been filled already, or is this one)
        }
        s.pscan = u;
        rows_max = s.rows > rows_max ? s.rows : rows_max;
        cols_max = s.cols > cols_max ? s.cols : cols_max;
    }
    bool side = true;  // every parent carries its own IdentifyBasicVariables scan
    for (int k = 0; k < count; ++k)
        side = side && b->nodes[parent_ids[k]].side && !b->nodes[parent_ids[k]].big;
    bool any_inplace = false;
    if (consume && side && !keep) {
        for (int k = 1; k < count; ++k) {
            const int pid = parent_ids[k];
            if (pid != parent_ids[k - 1] || b->h_slots[k - 1].inplace) continue;
            if (k + 1 < count && parent_ids[k + 1] == pid) continue;  // (not a pair: leave it)
            BBSlot& s = b->h_slots[k];
            b->free_bufs.push_back(s.cur);   // the buffer set aside for it above is not needed
            s.cur = b->nodes[pid].T;
            s.inplace = 1;
            b->nodes[pid].live = false;      // the child owns the buffer from here on
            b->nodes[pid].T = nullptr;
            any_inplace = true;
        }
    }
    LPR_HIP(hipMemcpyAsync(b->d_slots, b->h_slots, (size_t)count * sizeof(BBSlot),
                           hipMemcpyHostToDevice, st));
    b->h_running[0] = count;
    b->h_running[1] = 0;
    LPR_HIP(hipMemcpyAsync(b->d_running, b->h_running, 2 * sizeof(int32_t), hipMemcpyHostToDevice,
                           st));
    bb_launch_add_constraint(b, count, nparents, rows_max, cols_max, side, any_inplace);
    std::vector<int> kept_pivots;
    if (keep) {  // tableaux[0]: what AddConstraint hands to DoDualSimplex (:1105,:1172)
        keep->tabs.assign(count, {});
        keep->ntab.assign(count, 0);
        kept_pivots.assign(count, 0);
        LPR_HIP(hipGetLastError());
        for (int k = 0; k < count; ++k) {
            rc = bb_keep_copy(b, b->h_slots[k], &keep->tabs[k]);
            if (rc != LPR_OK_OPTIMAL) return rc;
            keep->ntab[k] = 1;
        }
    }

    // DoDualSimplex: pivot steps until every child has left the running states
    // Pivot steps queued between polls of the running counter.  A poll idles the device ~50 us; a
    // step queued after the last child has finished still dispatches every workgroup of the batch
    // (10-40 us at a wide level).  The children of one level need about as many steps as those of
    // the level before (running[1] of that batch), so the first batch is sized by it; without a
    // history 4, then doubling up to 32.
    // (round 3: a step queued behind the end of a batch is two near-empty launches, ~10 us, since
    // k_bb_update takes 8 listed rows per workgroup instead of one workgroup per 4 rows of every
    // child; a poll costs the device ~70 us of idling: the first batch is sized generously)
    int poll = b->last_steps > 4 ? b->last_steps + 2 : 6;
    if (keep) poll = 1;  // every tableau is copied off the device: one step at a time
    int queued = 0;
    int64_t guard = 0;
    for (;;) {
        // k_bb_select and k_bb_update always run as a pair: a select that starts a pivot sets
        // do_update (and backup / restore) for the update that follows it
        for (int k = 0; k < poll; ++k) bb_launch_pivot_step(b, count, rows_max, cols_max, ++queued);
        LPR_HIP(hipGetLastError());
        LPR_HIP(hipMemcpyAsync(b->h_running, b->d_running, 2 * sizeof(int32_t),
                               hipMemcpyDeviceToHost, st));
        LPR_HIP(hipStreamSynchronize(st));
        b->prof.polls += 1;
        b->prof.steps += poll;
        if (keep) {  // the list grows by the tableau of a pivot, shrinks by a dropped one (:395-400)
            LPR_HIP(hipMemcpyAsync(b->h_slots, b->d_slots, (size_t)count * sizeof(BBSlot),
                                   hipMemcpyDeviceToHost, st));
            LPR_HIP(hipStreamSynchronize(st));
            for (int k = 0; k < count; ++k) {
                const BBSlot& s = b->h_slots[k];
                const size_t n = (size_t)s.rows * s.cols;
                if (s.pivots > kept_pivots[k]) {
                    rc = bb_keep_copy(b, s, &keep->tabs[k]);
                    if (rc != LPR_OK_OPTIMAL) return rc;
                    keep->ntab[k] += 1;
                } else if (s.pivots < kept_pivots[k] && keep->ntab[k] > 0) {
                    keep->tabs[k].resize(keep->tabs[k].size() - n);
                    keep->ntab[k] -= 1;
                }
                kept_pivots[k] = s.pivots;
            }
            poll = 1;
            if (b->h_running[0] <= 0) break;
            continue;
        }
        if (b->h_running[0] <= 0) break;
        poll = (queued <= 4 && b->last_steps <= 4) ? 8 : 4;  // (then in fours: the tail is short)
        if (b->last_steps <= 4 && queued >= 12) poll = queued < 32 ? queued : 32;
        if (++guard > (1 << 16)) {
            // the reference has no pivot cap either (a cycling LP spins for ever in the C#);
            // the engine gives up instead of hanging the stream
            set_error("lpr_bb: DoDualSimplex did not terminate (cycling LP?)");
            return LPR_PIVOT_LIMIT;
        }
    }
    b->last_steps = b->h_running[1];
    // the loop only ends when no slot is running: the select that finishes a slot leaves
    // do_update = 0, and a dropped last pivot (restore) has been undone by the update kernel of the
    // same step
    // RoundAllTableaux(newTableaux) :1124 / :1187 + what the children will be scored / expanded by
    bb_launch_finish(b, count, cols_max);
    LPR_HIP(hipGetLastError());
    LPR_HIP(hipMemcpyAsync(b->h_slots, b->d_slots, (size_t)count * sizeof(BBSlot),
                           hipMemcpyDeviceToHost, st));
    std::vector<int32_t> tr;
    if (trace_out) {
        tr.resize((size_t)count * b->trace_cap * 3);
        LPR_HIP(hipMemcpyAsync(tr.data(), b->trace, tr.size() * sizeof(int32_t),
                               hipMemcpyDeviceToHost, st));
    }
    LPR_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < count; ++k) {
        const BBSlot& s = b->h_slots[k];
        status_out[k] = s.state;
        if (pivots_out) pivots_out[k] = s.pivots;
        b->total_pivots += s.trace_n;
        if (trace_out) {
            trace_off->push_back((int32_t)trace_out->size());
            const int nt = s.trace_n < b->trace_cap ? s.trace_n : b->trace_cap;
            const int32_t* src = tr.data() + (size_t)k * b->trace_cap * 3;
            trace_out->insert(trace_out->end(), src, src + (size_t)nt * 3);
        }
        if (s.state == kBBSolved) {
            child_ids_out[k] = bb_new_node(b, s.cur, s.rows, s.cols,
                                           b->nodes[parent_ids[k]].depth + 1);
            b->nodes[child_ids_out[k]].big = s.big != 0;
            b->nodes[child_ids_out[k]].side = s.big == 0;
            b->free_bufs.push_back(s.nxt);
        } else {
            child_ids_out[k] = -1;
            b->free_bufs.push_back(s.cur);
            b->free_bufs.push_back(s.nxt);
        }
    }
    if (trace_out) trace_off->push_back((int32_t)trace_out->size());
    return LPR_OK_OPTIMAL;
}

static void bb_release_node(lpr_bb* b, int id) {
    if (id < 0 || id >= (int)b->nodes.size() || !b->nodes[id].live) return;
    b->nodes[id].live = false;
    b->free_bufs.push_back(b->nodes[id].T);
    b->nodes[id].T = nullptr;
}

static int bb_create_common(lpr_engine* e, int rows, int cols, int nvars, int max_depth,
                            lpr_bb** out) {
    if (!e || !out || rows < 1 || cols < 2 || nvars < 0 || nvars > cols - 1) {
        set_error("lpr_bb_create: bad arguments (rows=%d cols=%d nvars=%d)", rows, cols, nvars);
        return LPR_BAD_ARGUMENT;
    }
    if (max_depth <= 0) max_depth = 64;
    if (rows + max_depth > 65535) {
        set_error("lpr_bb_create: rows + max_depth exceeds 65535");
        return LPR_BAD_ARGUMENT;
    }
    if ((size_t)align_up(cols + max_depth, kLdAlign) * sizeof(int) > kBBEliminateLdsMax) {
        // k_bb_eliminate (AddConstraint :756-796) ranks the basic columns of a child in LDS
        set_error("lpr_bb_create: cols + max_depth = %d exceeds the %zu columns the Branch & Bound "
                  "kernels keep in LDS", cols + max_depth, kBBEliminateLdsMax / sizeof(int));
        return LPR_BAD_ARGUMENT;
    }
    LPR_HIP(hipSetDevice(e->device));
    lpr_bb* b = new (std::nothrow) lpr_bb();
    if (!b) return LPR_OUT_OF_MEMORY;
    b->eng = e;
    b->rows0 = rows;
    b->cols0 = cols;
    b->nvars = nvars;
    b->max_depth = max_depth;
    b->rows_cap = rows + max_depth;
    b->ld = align_up(cols + max_depth, kLdAlign);
    b->buf_elems = (size_t)(b->rows_cap + 2) * b->ld;
    e->live_bb.push_back(b);
    *out = b;
    return LPR_OK_OPTIMAL;
}

}  // namespace lpr

using namespace lpr;

#define LPR_LIVE_BB(b)                                                              \
    do {                                                                            \
        if (!(b) || !(b)->eng) {                                                    \
            set_error("B&B handle is null or its engine has been closed");         \
            return LPR_BAD_ARGUMENT;                                                \
        }                                                                           \
    } while (0)

extern "C" {

int lpr_bb_create(lpr_engine* e, const double* final_tableau, int rows, int cols, int nvars,
                  int max_depth, lpr_bb** out) {
    if (!final_tableau) {
        set_error("lpr_bb_create: null tableau");
        return LPR_BAD_ARGUMENT;
    }
    lpr_bb* b = nullptr;
    int rc = bb_create_common(e, rows, cols, nvars, max_depth, &b);
    if (rc != LPR_OK_OPTIMAL) return rc;
    double* T = nullptr;
    rc = bb_alloc_buf(b, &T);
    if (rc != LPR_OK_OPTIMAL) {
        lpr_bb_destroy(b);
        return rc;
    }
    hipStream_t st = e->stream;
    hipError_t err = hipMemsetAsync(T, 0, b->buf_elems * sizeof(double), st);
    if (err == hipSuccess)
        err = hipMemcpy2DAsync(T, (size_t)b->ld * sizeof(double), final_tableau,
                               (size_t)cols * sizeof(double), (size_t)cols * sizeof(double), rows,
                               hipMemcpyHostToDevice, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    if (err != hipSuccess) {
        set_error("lpr_bb_create: %s", hipGetErrorString(err));
        lpr_bb_destroy(b);
        return LPR_DEVICE_ERROR;
    }
    bb_new_node(b, T, rows, cols, 0);  // node 0 = the root (Convert(primal.FinalTableau))
    *out = b;
    return LPR_OK_OPTIMAL;
}

int lpr_bb_create_from_tableau(lpr_tableau* t, int nvars, int max_depth, lpr_bb** out) {
    if (!t || !t->eng) {
        set_error("lpr_bb_create_from_tableau: tableau handle is null or orphaned");
        return LPR_BAD_ARGUMENT;
    }
    lpr_bb* b = nullptr;
    int rc = bb_create_common(t->eng, t->rows, t->cols, nvars, max_depth, &b);
    if (rc != LPR_OK_OPTIMAL) return rc;
    double* T = nullptr;
    rc = bb_alloc_buf(b, &T);
    if (rc != LPR_OK_OPTIMAL) {
        lpr_bb_destroy(b);
        return rc;
    }
    bb_launch_copy_in(b, t->T, t->ld, t->rows, t->cols, T);  // device -> device, no host trip
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipStreamSynchronize(t->eng->stream);
    if (err != hipSuccess) {
        set_error("lpr_bb_create_from_tableau: %s", hipGetErrorString(err));
        lpr_bb_destroy(b);
        return LPR_DEVICE_ERROR;
    }
    bb_new_node(b, T, t->rows, t->cols, 0);
    *out = b;
    return LPR_OK_OPTIMAL;
}

int lpr_bb_destroy(lpr_bb* b) {
    if (!b) return LPR_BAD_ARGUMENT;
    if (b->eng) {
        bb_release_device(b);
        auto& lv = b->eng->live_bb;
        for (size_t k = 0; k < lv.size(); ++k)
            if (lv[k] == b) {
                lv.erase(lv.begin() + k);
                break;
            }
    }
    delete b;
    return LPR_OK_OPTIMAL;
}

int lpr_bb_node_info(lpr_bb* b, const int32_t* ids, int count, double* z_out, double* vals_out) {
    LPR_LIVE_BB(b);
    if (count < 0 || (count > 0 && (!ids || !z_out || (b->nvars > 0 && !vals_out))))
        return LPR_BAD_ARGUMENT;
    LPR_HIP(hipSetDevice(b->eng->device));
    return bb_node_info(b, ids, count, z_out, vals_out);
}

int lpr_bb_expand(lpr_bb* b, int count, const int32_t* parent_ids, const int32_t* var,
                  const double* bound, const int32_t* kind, int32_t* child_ids_out,
                  int32_t* status_out, int32_t* pivots_out) {
    LPR_LIVE_BB(b);
    if (count < 0 || (count > 0 && (!parent_ids || !var || !bound || !kind || !child_ids_out ||
                                    !status_out)))
        return LPR_BAD_ARGUMENT;
    for (int k = 0; k < count; ++k)
        if (var[k] < 0 || var[k] >= b->nvars) {
            set_error("lpr_bb_expand: branching variable %d out of range", var[k]);
            return LPR_BAD_ARGUMENT;
        }
    LPR_HIP(hipSetDevice(b->eng->device));
    return bb_expand(b, count, parent_ids, var, bound, kind, child_ids_out, status_out,
                     pivots_out, nullptr, nullptr);
}

int lpr_bb_expand_traced(lpr_bb* b, int count, const int32_t* parent_ids, const int32_t* var,
                         const double* bound, const int32_t* kind, int32_t* child_ids_out,
                         int32_t* status_out, int32_t* pivots_out, int32_t* trace_out,
                         int64_t trace_cap, int64_t* trace_off_out, double* tab_out, int64_t tab_cap,
                         int64_t* tab_off_out, int32_t* ntab_out) {
    LPR_LIVE_BB(b);
    if (count < 0 || (count > 0 && (!parent_ids || !var || !bound || !kind || !child_ids_out ||
                                    !status_out || !trace_off_out || !tab_off_out || !ntab_out)) ||
        trace_cap < 0 || tab_cap < 0)
        return LPR_BAD_ARGUMENT;
    for (int k = 0; k < count; ++k)
        if (var[k] < 0 || var[k] >= b->nvars) {
            set_error("lpr_bb_expand_traced: branching variable %d out of range", var[k]);
            return LPR_BAD_ARGUMENT;
        }
    LPR_HIP(hipSetDevice(b->eng->device));
    std::vector<int32_t> tr, off;
    BBKeep keep;
    int rc = bb_expand(b, count, parent_ids, var, bound, kind, child_ids_out, status_out,
                       pivots_out, &tr, &off, &keep);
    if (rc != LPR_OK_OPTIMAL) return rc;
    int64_t tpos = 0;
    for (int k = 0; k < count; ++k) {
        trace_off_out[k] = off[k] / 3;
        const size_t n = keep.tabs[k].size();
        tab_off_out[k] = tpos;
        ntab_out[k] = keep.ntab[k];
        if (tab_out && tpos + (int64_t)n <= tab_cap)
            std::memcpy(tab_out + tpos, keep.tabs[k].data(), n * sizeof(double));
        tpos += (int64_t)n;
    }
    trace_off_out[count] = off[count] / 3;
    tab_off_out[count] = tpos;
    const int64_t nt = (int64_t)tr.size() / 3;
    if (trace_out) std::memcpy(trace_out, tr.data(), (size_t)(nt < trace_cap ? nt : trace_cap) * 3 * sizeof(int32_t));
    if ((tab_out && tpos > tab_cap) || (trace_out && nt > trace_cap)) {
        set_error("lpr_bb_expand_traced: output buffers too small (%lld tableau doubles, %lld pivot "
                  "triples needed)", (long long)tpos, (long long)nt);
        return LPR_BAD_ARGUMENT;
    }
    return LPR_OK_OPTIMAL;
}

int lpr_bb_release(lpr_bb* b, const int32_t* ids, int count) {
    LPR_LIVE_BB(b);
    for (int k = 0; k < count; ++k) bb_release_node(b, ids[k]);
    return LPR_OK_OPTIMAL;
}

int lpr_bb_node_read(lpr_bb* b, int32_t id, double* out, int32_t* rows, int32_t* cols) {
    LPR_LIVE_BB(b);
    if (id < 0 || id >= (int)b->nodes.size() || !b->nodes[id].live) {
        set_error("lpr_bb_node_read: node %d is not live", id);
        return LPR_BAD_ARGUMENT;
    }
    const lpr_bb::Node& nd = b->nodes[id];
    if (rows) *rows = nd.rows;
    if (cols) *cols = nd.cols;
    if (!out) return LPR_OK_OPTIMAL;
    LPR_HIP(hipSetDevice(b->eng->device));
    hipStream_t st = b->eng->stream;
    LPR_HIP(hipMemcpy2DAsync(out, (size_t)nd.cols * sizeof(double), nd.T,
                             (size_t)b->ld * sizeof(double), (size_t)nd.cols * sizeof(double),
                             nd.rows, hipMemcpyDeviceToHost, st));
    LPR_HIP(hipStreamSynchronize(st));
    return LPR_OK_OPTIMAL;
}

// ExecuteBranchAndBound (:1006-1233): DFS stack, lower child first, incumbent replaced on strict
// improvement only, optional pruning, node cap (20 in the reference).
int lpr_bb_run(lpr_bb* b, const lpr_bb_opts* opts, double* x, lpr_bb_result* res) {
    LPR_LIVE_BB(b);
    if (!res) return LPR_BAD_ARGUMENT;
    LPR_HIP(hipSetDevice(b->eng->device));
    lpr_bb_opts o;
    std::memset(&o, 0, sizeof o);
    if (opts) o = *opts;
    const int node_cap = o.node_cap > 0 ? o.node_cap : 20;  // :1038
    const int n = b->nvars;
    if (b->nodes.empty() || !b->nodes[0].live) {
        set_error("lpr_bb_run: the root node has been consumed; create a new handle");
        return LPR_BAD_ARGUMENT;
    }

    b->records.clear();
    b->pop_order.clear();
    b->piv_trace.clear();
    b->best_x.assign(n > 0 ? n : 1, 0.0);
    b->best_z = -INFINITY;  // :1024 (isMinimization is never set by the adapter)
    b->best_node = -1;
    b->found = false;
    b->total_pivots = 0;

    // initialTableaux = RoundAllTableaux(initialTableaux) :1021 -- the pop rounds again (:1047)
    std::vector<double> vals((size_t)(n > 0 ? n : 1));
    struct Item { int node; int depth; int rec; };
    std::vector<Item> stack;
    {
        int32_t root = 0;
        double z0 = 0;
        int rc = bb_node_info(b, &root, 1, &z0, vals.data());
        if (rc != LPR_OK_OPTIMAL) return rc;
        b->records.push_back({-1, 0, 0, -1, 0, 0.0, z0});
        stack.push_back({0, 0, 0});
    }

    int status = LPR_OK_OPTIMAL;
    int iteration = 0;
    int64_t processed = 0;
    while (!stack.empty()) {
        ++iteration;
        if (iteration > node_cap) {  // "Potential infinite loop detected" :1038-1042
            status = LPR_BB_NODE_CAP;
            break;
        }
        const Item it = stack.back();
        stack.pop_back();
        b->pop_order.push_back(it.rec);
        ++processed;
        int32_t nid = it.node;
        double objVal = 0;
        int rc = bb_node_info(b, &nid, 1, &objVal, vals.data());  // :1047, :892-897, :899-921
        if (rc != LPR_OK_OPTIMAL) return rc;

        if (o.enable_pruning && b->found && objVal <= b->best_z) {  // ShouldPrunebranch :985-1004
            bb_release_node(b, nid);
            continue;
        }
        bool allInt = true;  // UpdateOptimalSolution :935-983
        for (int i = 0; i < n; ++i)
            if (!is_integer(vals[i])) { allInt = false; break; }
        if (allInt && objVal > b->best_z) {
            b->best_z = objVal;
            for (int i = 0; i < n; ++i) b->best_x[i] = vals[i];
            b->found = true;
            b->best_node = it.rec;
        }
        int bestVar = -1;  // CheckIntegerBasicVar :829-847
        double bestValue = 0, minDist = INFINITY;
        for (int i = 0; i < n; ++i) {
            if (!is_integer(vals[i])) {
                const double frac = vals[i] - std::floor(vals[i]);
                const double dist = std::fabs(frac - 0.5);
                if (dist < minDist) {
                    minDist = dist;
                    bestVar = i;
                    bestValue = vals[i];
                }
            }
        }
        if (bestVar < 0) {  // integer node :1070-1076
            bb_release_node(b, nid);
            continue;
        }
        const int upperInt = dn_to_int32(std::ceil(bestValue));   // :870-871
        const int lowerInt = dn_to_int32(std::floor(bestValue));

        // both children in one batch: lower (<= floor, :1083-1148), upper (>= ceil, :1150-1208)
        int32_t parents[2] = {nid, nid};
        int32_t vars[2] = {bestVar, bestVar};
        double bounds[2] = {(double)lowerInt, (double)upperInt};
        int32_t kinds[2] = {0, 1};
        int32_t child[2] = {-1, -1}, cst[2] = {0, 0}, cpiv[2] = {0, 0};
        std::vector<int32_t> tr, off;
        rc = bb_expand(b, 2, parents, vars, bounds, kinds, child, cst, cpiv, &tr, &off, nullptr,
                       /*consume=*/true);
        if (rc != LPR_OK_OPTIMAL) {
            if (rc == LPR_BB_NODE_CAP) { status = rc; break; }
            return rc;
        }
        Item kids[2];
        int nk = 0;
        for (int side = 0; side < 2; ++side) {
            const int rid = (int)b->records.size();
            for (int q = off[side]; q < off[side + 1]; q += 3) {
                b->piv_trace.push_back(rid);
                b->piv_trace.push_back(tr[q]);
                b->piv_trace.push_back(tr[q + 1]);
                b->piv_trace.push_back(tr[q + 2]);
            }
            if (cst[side] == kBBSolved) {
                int32_t cid = child[side];
                double cz = 0;
                std::vector<double> cv((size_t)(n > 0 ? n : 1));
                rc = bb_node_info(b, &cid, 1, &cz, cv.data());
                if (rc != LPR_OK_OPTIMAL) return rc;
                b->records.push_back({it.rec, side + 1, it.depth + 1, bestVar, 0, bounds[side], cz});
                kids[nk++] = {cid, it.depth + 1, rid};
            } else {
                const int st2 = cst[side] == kBBInfeasible ? 1 : 2;
                b->records.push_back({it.rec, side + 1, it.depth + 1, bestVar, st2, bounds[side],
                                      0.0});
            }
        }
        for (int k = nk - 1; k >= 0; --k) stack.push_back(kids[k]);  // :1210-1213
        bb_release_node(b, nid);
    }
    for (const Item& it : stack) bb_release_node(b, it.node);

    if (x && b->found)
        for (int i = 0; i < n; ++i) x[i] = b->best_x[i];
    res->status = status;
    res->found = b->found ? 1 : 0;
    res->processed = processed;
    res->best_node = b->best_node;
    res->reserved = 0;
    res->z = b->best_z;
    res->pivots = b->total_pivots;
    res->nodes_created = (int64_t)b->records.size();
    return status;
}

// ---- level-synchronous, multi-rank form (comm_engine.hip holds the transport) ----------------

namespace {

// A node's branch path from the root: side k (0 lower, 1 upper) in bit k, `len` sides.  The
// reference's stack pops in pre-order, lower child first: lexicographic order of the paths, an
// ancestor before its descendants.
struct Path {
    uint64_t bits = 0;
    int len = 0;
};
bool dfs_before(const Path& a, const Path& b) {
    const int n = a.len < b.len ? a.len : b.len;
    for (int k = 0; k < n; ++k) {
        const int x = (int)((a.bits >> k) & 1u), y = (int)((b.bits >> k) & 1u);
        if (x != y) return x < y;
    }
    return a.len < b.len;
}
struct Front {
    int node;
    Path path;
};
// A rank-local lpr_status as a slot of the level's MAX all-reduce: 0 = fine, failures (< 0) above
// the non-fatal stops (> 0) so that the worst one wins.
double bb_encode_rc(int rc) { return rc == 0 ? 0.0 : (rc < 0 ? 1000.0 - (double)rc : (double)rc); }
int bb_decode_rc(double e) { return e > 1000.0 ? -(int)(e - 1000.0) : (int)e; }

}  // namespace

int lpr_bb_solve_level_sync(lpr_bb* b, lpr_comm* comm, const lpr_bb_sync_opts* opts, double* x,
                            lpr_bb_sync_result* res) {
    LPR_LIVE_BB(b);
    if (!res) return LPR_BAD_ARGUMENT;
    LPR_HIP(hipSetDevice(b->eng->device));
    const double t_begin = bb_now();
    b->prof = lpr_bb::Prof();
    lpr_bb_sync_opts o;
    std::memset(&o, 0, sizeof o);
    if (opts) o = *opts;
    const int n = b->nvars;
    int max_levels = o.max_levels > 0 ? o.max_levels : b->max_depth;
    if (max_levels > b->max_depth) max_levels = b->max_depth;  // node buffers are sized for it
    if (max_levels > 64) max_levels = 64;                      // the path is a 64-bit word
    const int64_t max_nodes = o.max_nodes > 0 ? o.max_nodes : ((int64_t)1 << 20);
    if (b->nodes.empty() || !b->nodes[0].live) {
        set_error("lpr_bb_solve_level_sync: the root node has been consumed; create a new handle");
        return LPR_BAD_ARGUMENT;
    }
    const int rank = comm_rank(comm), world = comm_world(comm);
    int split_level = 0;
    while ((1 << split_level) < world) ++split_level;

    {
        // the widest level has at most 2^levels children: size the per-child scratch ONCE (grown
        // by doubling level after level it cost 4.9 of 37 ms: nine synchronisations and ninety
        // allocations, two of them pinned)
        int64_t widest = (int64_t)1 << (max_levels < 10 ? max_levels : 10);
        if (widest > 2 * max_nodes) widest = 2 * max_nodes;
        if (widest < 2) widest = 2;
        int rc = bb_ensure_slots(b, (int)widest);
        if (rc != LPR_OK_OPTIMAL) return rc;
    }
    std::vector<Front> frontier{{0, Path()}};
    bool have_best = false;
    double best_z_local = -INFINITY;
    Path best_path;
    std::vector<double> best_x((size_t)(n > 0 ? n : 1), 0.0);
    double global_bound = -INFINITY;
    int64_t processed = 0, pivots = 0;
    int levels = 0;
    bool capped = false;
    std::vector<int32_t> ids, parents, var, kind, child, cst, cpiv;
    std::vector<double> zs, vals, bound;
    std::vector<Path> paths;

    // Test hook (fault injection, tests/test_bb_gpu.py): LPR_BB_INJECT_FAULT="<rank>:<level>" makes
    // that rank's batch of that level fail as a device error would.
    int fault_rank = -1, fault_level = -1;
    if (const char* fv = std::getenv("LPR_BB_INJECT_FAULT"))
        if (std::sscanf(fv, "%d:%d", &fault_rank, &fault_level) != 2) fault_rank = fault_level = -1;

    int agreed_rc = LPR_OK_OPTIMAL;  // a failure of ANY rank, learnt by all from the level's all-reduce
    bool depth_capped = false;
    for (;;) {
        // A level past max_levels is scored but not branched: its nodes were solved by the level
        // before and would otherwise be dropped unseen (an integer optimum sitting exactly at depth
        // max_levels).  Nodes left unbranched there make the result LPR_BB_DEPTH_CAP.
        const bool may_expand = levels < max_levels;
        const bool replicated = levels < split_level;  // every rank is doing the same nodes
        const bool count_here = !replicated || rank == 0;
        int local_rc = LPR_OK_OPTIMAL;  // never returned before the collective: the peers would hang
        bool unbranched = false;
        parents.clear(); var.clear(); bound.clear(); kind.clear(); paths.clear();
        if (!frontier.empty()) {
            const int cnt = (int)frontier.size();
            ids.resize(cnt);
            for (int q = 0; q < cnt; ++q) ids[q] = frontier[q].node;
            zs.assign(cnt, 0.0);
            vals.assign((size_t)cnt * (n > 0 ? n : 1), 0.0);
            const double t_i = bb_now();
            local_rc = bb_node_info(b, ids.data(), cnt, zs.data(), vals.data());
            b->prof.info += bb_now() - t_i;
            for (int q = 0; q < cnt && local_rc == LPR_OK_OPTIMAL; ++q) {
                const double* v = vals.data() + (size_t)q * n;
                const double z = zs[q];
                if (count_here) ++processed;
                if (o.enable_pruning && global_bound > -INFINITY && z <= global_bound)
                    continue;  // ShouldPrunebranch :995-1001 against the all-reduced bound
                bool allInt = true;  // UpdateOptimalSolution :943-981
                for (int i = 0; i < n; ++i)
                    if (!is_integer(v[i])) { allInt = false; break; }
                if (allInt && (!have_best || z > best_z_local ||
                               (z == best_z_local && dfs_before(frontier[q].path, best_path)))) {
                    have_best = true;
                    best_z_local = z;
                    best_path = frontier[q].path;
                    for (int i = 0; i < n; ++i) best_x[i] = v[i];
                }
                int bestVar = -1;  // CheckIntegerBasicVar :829-847
                double bestValue = 0, minDist = INFINITY;
                for (int i = 0; i < n; ++i) {
                    if (!is_integer(v[i])) {
                        const double dist = std::fabs((v[i] - std::floor(v[i])) - 0.5);
                        if (dist < minDist) {
                            minDist = dist;
                            bestVar = i;
                            bestValue = v[i];
                        }
                    }
                }
                if (bestVar < 0) continue;  // an integer node has no children (:1070-1076)
                if (!may_expand) { unbranched = true; continue; }
                for (int side = 0; side < 2; ++side) {  // CreateBranches :859-890
                    parents.push_back(frontier[q].node);
                    var.push_back(bestVar);
                    bound.push_back((double)dn_to_int32(side == 0 ? std::floor(bestValue)
                                                                  : std::ceil(bestValue)));
                    kind.push_back(side);
                    Path p = frontier[q].path;
                    if (side) p.bits |= (uint64_t)1 << p.len;
                    p.len += 1;
                    paths.push_back(p);
                }
            }
        }
        std::vector<Front> next;
        if (local_rc == LPR_OK_OPTIMAL && !parents.empty()) {
            const int cnt = (int)parents.size();
            child.assign(cnt, -1);
            cst.assign(cnt, 0);
            cpiv.assign(cnt, 0);
            const double t_e = bb_now();
            local_rc = bb_expand(b, cnt, parents.data(), var.data(), bound.data(), kind.data(),
                                 child.data(), cst.data(), cpiv.data(), nullptr, nullptr, nullptr,
                                 /*consume=*/true);
            b->prof.expand += bb_now() - t_e;
            for (int q = 0; q < cnt && local_rc == LPR_OK_OPTIMAL; ++q) {
                if (count_here) pivots += cpiv[q];
                if (cst[q] == kBBSolved) next.push_back({child[q], paths[q]});
            }
        }
        if (rank == fault_rank && levels == fault_level && local_rc == LPR_OK_OPTIMAL) {
            set_error("lpr_bb_solve_level_sync: injected fault (LPR_BB_INJECT_FAULT) on rank %d, "
                      "level %d", rank, levels);
            local_rc = LPR_DEVICE_ERROR;
        }
        for (const Front& f : frontier) bb_release_node(b, f.node);
        if (may_expand) ++levels;
        if (may_expand && levels == split_level && world > 1 && local_rc == LPR_OK_OPTIMAL) {
            // deal the frontier of this depth: identical on every rank, so nothing is exchanged
            std::stable_sort(next.begin(), next.end(),
                             [](const Front& a, const Front& c) { return dfs_before(a.path, c.path); });
            std::vector<Front> keep;
            for (size_t i = 0; i < next.size(); ++i) {
                if ((int)(i % (size_t)world) == rank) keep.push_back(next[i]);
                else bb_release_node(b, next[i].node);
            }
            next.swap(keep);
        }
        frontier.swap(next);
        // ---- the single collective of the level: incumbent bound, "someone has nodes left",
        //      "someone has hit max_nodes", "someone FAILED" (its status, so that every rank leaves
        //      on this level with it instead of waiting for ever in the next collective), "someone
        //      left nodes unbranched at max_levels" -- all decided by every rank from the same maxima
        double red[5] = {best_z_local, frontier.empty() ? 0.0 : 1.0,
                         processed > max_nodes ? 1.0 : 0.0, bb_encode_rc(local_rc),
                         unbranched ? 1.0 : 0.0};
        const double t_c = bb_now();
        int rc = comm_all_reduce_max(comm, red, 5);
        b->prof.comm += bb_now() - t_c;
        if (rc != LPR_OK_OPTIMAL) {  // the transport itself is gone: nothing left to agree through
            for (const Front& f : frontier) bb_release_node(b, f.node);
            return rc;
        }
        global_bound = red[0];
        if (red[3] > 0.5) {
            agreed_rc = bb_decode_rc(red[3]);
            if (local_rc == LPR_OK_OPTIMAL)
                set_error("lpr_bb_solve_level_sync: another rank failed with status %d at level %d "
                          "(this rank, %d of %d, was fine)", agreed_rc, levels, rank, world);
            break;
        }
        if (red[2] > 0.5) capped = true;
        if (red[4] > 0.5) depth_capped = true;
        if (red[1] < 0.5 || capped || !may_expand) break;
    }
    for (const Front& f : frontier) bb_release_node(b, f.node);
    if (agreed_rc != LPR_OK_OPTIMAL) {  // every rank is here on the same level: no gather
        res->status = agreed_rc;
        res->found = 0;
        res->processed = processed;
        res->pivots = pivots;
        res->levels = levels;
        res->path_len = 0;
        res->path_bits = 0;
        res->z = -INFINITY;
        return agreed_rc;
    }

    // winner identity: one gather at termination, ties by DFS order (:966 "first found wins")
    const int nx = n > 0 ? n : 1;
    const size_t rec_d = 6 + (size_t)nx;  // doubles per rank
    std::vector<double> mine(rec_d, 0.0), all(rec_d * (size_t)world, 0.0);
    mine[0] = have_best ? 1.0 : 0.0;
    mine[1] = best_z_local;
    std::memcpy(&mine[2], &best_path.bits, sizeof(uint64_t));
    mine[3] = (double)best_path.len;
    mine[4] = (double)processed;
    mine[5] = (double)pivots;
    for (int i = 0; i < n; ++i) mine[6 + i] = best_x[i];
    int rc = comm_all_gather(comm, mine.data(), all.data(), (int)(rec_d * sizeof(double)));
    if (rc != LPR_OK_OPTIMAL) return rc;
    bool found = false;
    double wz = -INFINITY;
    Path wp;
    int wr = -1;
    int64_t tot_proc = 0, tot_piv = 0;
    for (int r = 0; r < world; ++r) {
        const double* rr = all.data() + rec_d * (size_t)r;
        tot_proc += (int64_t)rr[4];
        tot_piv += (int64_t)rr[5];
        if (rr[0] < 0.5) continue;
        Path p;
        std::memcpy(&p.bits, &rr[2], sizeof(uint64_t));
        p.len = (int)rr[3];
        if (!found || rr[1] > wz || (rr[1] == wz && dfs_before(p, wp))) {
            found = true;
            wz = rr[1];
            wp = p;
            wr = r;
        }
    }
    if (found && x)
        for (int i = 0; i < n; ++i) x[i] = all[rec_d * (size_t)wr + 6 + i];
    if (const char* tv = std::getenv("LPR_BB_TIMING"); tv && tv[0] == '1')
        std::fprintf(stderr,
                     "lpr_bb timing: total %.3f ms | node_info %.3f | expand %.3f (of which buffer "
                     "allocation %.3f in %d hipMalloc) | slot scratch %.3f | collectives %.3f | "
                     "%d polls, %d pivot steps queued\n",
                     1e3 * (bb_now() - t_begin), 1e3 * b->prof.info, 1e3 * b->prof.expand,
                     1e3 * b->prof.alloc, b->prof.mallocs, 1e3 * b->prof.slots, 1e3 * b->prof.comm,
                     b->prof.polls, b->prof.steps);
    res->status = capped ? LPR_BB_NODE_CAP : (depth_capped ? LPR_BB_DEPTH_CAP : LPR_OK_OPTIMAL);
    res->found = found ? 1 : 0;
    res->processed = tot_proc;
    res->pivots = tot_piv;
    res->levels = levels;
    res->path_len = found ? wp.len : 0;
    res->path_bits = found ? wp.bits : 0;
    res->z = found ? wz : -INFINITY;
    return res->status;
}

int lpr_bb_records_read(lpr_bb* b, int32_t* parent, int32_t* kind, int32_t* depth, int32_t* var,
                        double* bound, int32_t* status, double* z, int64_t cap, int64_t* count) {
    if (!b || !count || cap < 0) return LPR_BAD_ARGUMENT;
    int64_t k = (int64_t)b->records.size();
    if (k > cap) k = cap;
    *count = k;
    for (int64_t q = 0; q < k; ++q) {
        const lpr_bb::Rec& r = b->records[q];
        if (parent) parent[q] = r.parent;
        if (kind) kind[q] = r.kind;
        if (depth) depth[q] = r.depth;
        if (var) var[q] = r.var;
        if (bound) bound[q] = r.bound;
        if (status) status[q] = r.status;
        if (z) z[q] = r.z;
    }
    return LPR_OK_OPTIMAL;
}

int lpr_bb_pop_order_read(lpr_bb* b, int32_t* ids, int64_t cap, int64_t* count) {
    if (!b || !count || cap < 0) return LPR_BAD_ARGUMENT;
    int64_t k = (int64_t)b->pop_order.size();
    if (k > cap) k = cap;
    *count = k;
    for (int64_t q = 0; q < k; ++q)
        if (ids) ids[q] = b->pop_order[q];
    return LPR_OK_OPTIMAL;
}

int lpr_bb_trace_read(lpr_bb* b, int32_t* quads, int64_t cap, int64_t* count) {
    if (!b || !count || cap < 0) return LPR_BAD_ARGUMENT;
    int64_t k = (int64_t)b->piv_trace.size() / 4;
    if (k > cap) k = cap;
    *count = k;
    if (quads) std::memcpy(quads, b->piv_trace.data(), (size_t)k * 4 * sizeof(int32_t));
    return LPR_OK_OPTIMAL;
}

}  // extern "C"
