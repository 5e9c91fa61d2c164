// comm_engine.hip -- C ABI of the multi-GPU communicator (include/lpr_engine.h, lpr_comm_*).
// One process per GPU; the only thing that crosses xGMI on the Branch & Bound path is the incumbent
// bound: ONE ncclAllReduce(ncclMax) of three doubles per level, called on RCCL
// (/opt/rocm/lib/librccl.so) directly from here, plus one all-gather of the winners at the end
// (reference loop being sharded: BranchBoundSimplexSolver.cs:1006-1233).
// A caller that brings its own transport (tests over gloo, a host with its own fabric) passes two
// callbacks instead; they are invoked on the calling thread only.
#include "engine_common.hpp"

#include <rccl/rccl.h>

#include <new>

struct lpr_comm {
    lpr_engine* eng = nullptr;   // RCCL form: the engine whose device / stream the collectives use
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;
    double* d_buf = nullptr;     // staging for the all-reduce (64 doubles)
    unsigned char* d_send = nullptr;  // staging for the all-gather
    unsigned char* d_recv = nullptr;
    size_t gather_cap = 0;
    lpr_allreduce_max_fn ar = nullptr;  // custom transport
    lpr_allgather_fn ag = nullptr;
    void* user = nullptr;
    int64_t allreduce_calls = 0, allgather_calls = 0;
    bool orphaned = false;              // its engine was closed first
};

namespace lpr {

#define LPR_NCCL(expr)                                                                    \
    do {                                                                                  \
        ncclResult_t _r = (expr);                                                         \
        if (_r != ncclSuccess) {                                                          \
            ::lpr::set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r),      \
                             __FILE__, __LINE__);                                         \
            return LPR_DEVICE_ERROR;                                                      \
        }                                                                                 \
    } while (0)

// lpr_engine_close with the communicator still alive: RCCL's resources go with the stream they
// were used on; the handle stays valid for lpr_comm_destroy / lpr_comm_info and refuses collectives.
static void comm_release_device(lpr_comm* c) {
    if (c->nccl) {
        hipSetDevice(c->eng->device);
        hipStreamSynchronize(c->eng->stream);
        ncclCommDestroy(c->nccl);
        c->nccl = nullptr;
    }
    hipFree(c->d_buf);
    hipFree(c->d_send);
    hipFree(c->d_recv);
    c->d_buf = nullptr;
    c->d_send = c->d_recv = nullptr;
    c->gather_cap = 0;
}
void comm_orphan(lpr_comm* c) {
    comm_release_device(c);
    c->eng = nullptr;
    c->orphaned = true;
}

int comm_rank(const lpr_comm* c) { return c ? c->rank : 0; }
int comm_world(const lpr_comm* c) { return c ? c->world : 1; }

// v[0..n) <- element-wise maximum over all ranks
int comm_all_reduce_max(lpr_comm* c, double* v, int n) {
    if (c && c->orphaned) {
        set_error("lpr_comm: the engine of this communicator has been closed");
        return LPR_BAD_ARGUMENT;
    }
    if (!c || (c->world == 1 && !c->nccl && !c->ar)) return LPR_OK_OPTIMAL;
    if (n < 1 || n > 64) return LPR_BAD_ARGUMENT;
    c->allreduce_calls += 1;
    if (c->ar) {
        if (c->ar(c->user, v, n) != 0) {
            set_error("lpr_comm: the caller's all-reduce callback failed");
            return LPR_DEVICE_ERROR;
        }
        return LPR_OK_OPTIMAL;
    }
    hipStream_t st = c->eng->stream;
    LPR_HIP(hipSetDevice(c->eng->device));
    LPR_HIP(hipMemcpyAsync(c->d_buf, v, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    LPR_NCCL(ncclAllReduce(c->d_buf, c->d_buf, (size_t)n, ncclDouble, ncclMax, c->nccl, st));
    LPR_HIP(hipMemcpyAsync(v, c->d_buf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    LPR_HIP(hipStreamSynchronize(st));
    return LPR_OK_OPTIMAL;
}

// recv[world * bytes] <- send[bytes] of every rank, in rank order
int comm_all_gather(lpr_comm* c, const void* send, void* recv, int bytes) {
    if (c && c->orphaned) {
        set_error("lpr_comm: the engine of this communicator has been closed");
        return LPR_BAD_ARGUMENT;
    }
    if (!c || (c->world == 1 && !c->nccl && !c->ag)) {
        std::memcpy(recv, send, (size_t)bytes);
        return LPR_OK_OPTIMAL;
    }
    c->allgather_calls += 1;
    if (c->ag) {
        if (c->ag(c->user, send, recv, bytes) != 0) {
            set_error("lpr_comm: the caller's all-gather callback failed");
            return LPR_DEVICE_ERROR;
        }
        return LPR_OK_OPTIMAL;
    }
    hipStream_t st = c->eng->stream;
    LPR_HIP(hipSetDevice(c->eng->device));
    const size_t need = (size_t)bytes * c->world;
    if (need > c->gather_cap) {
        hipFree(c->d_send);
        hipFree(c->d_recv);
        c->d_send = c->d_recv = nullptr;
        c->gather_cap = 0;
        LPR_HIP(hipMalloc(&c->d_send, (size_t)bytes));
        LPR_HIP(hipMalloc(&c->d_recv, need));
        c->gather_cap = need;
    }
    LPR_HIP(hipMemcpyAsync(c->d_send, send, (size_t)bytes, hipMemcpyHostToDevice, st));
    LPR_NCCL(ncclAllGather(c->d_send, c->d_recv, (size_t)bytes, ncclUint8, c->nccl, st));
    LPR_HIP(hipMemcpyAsync(recv, c->d_recv, need, hipMemcpyDeviceToHost, st));
    LPR_HIP(hipStreamSynchronize(st));
    return LPR_OK_OPTIMAL;
}

}  // namespace lpr

using namespace lpr;

extern "C" {

int lpr_comm_unique_id(uint8_t id[LPR_COMM_ID_BYTES]) {
    if (!id) return LPR_BAD_ARGUMENT;
    static_assert(LPR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size follows RCCL's");
    ncclUniqueId u;
    LPR_NCCL(ncclGetUniqueId(&u));
    std::memcpy(id, u.internal, LPR_COMM_ID_BYTES);
    return LPR_OK_OPTIMAL;
}

int lpr_comm_init(lpr_engine* e, int rank, int world, const uint8_t id[LPR_COMM_ID_BYTES],
                  lpr_comm** out) {
    if (!e || !out || !id || world < 1 || rank < 0 || rank >= world) {
        set_error("lpr_comm_init: bad arguments (rank %d of %d)", rank, world);
        return LPR_BAD_ARGUMENT;
    }
    *out = nullptr;
    LPR_HIP(hipSetDevice(e->device));
    lpr_comm* c = new (std::nothrow) lpr_comm();
    if (!c) return LPR_OUT_OF_MEMORY;
    c->eng = e;
    c->rank = rank;
    c->world = world;
    ncclUniqueId u;
    std::memcpy(u.internal, id, LPR_COMM_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&c->nccl, world, u, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, ncclGetErrorString(r));
        delete c;
        return LPR_DEVICE_ERROR;
    }
    if (hipMalloc(&c->d_buf, 64 * sizeof(double)) != hipSuccess) {
        ncclCommDestroy(c->nccl);
        delete c;
        return LPR_OUT_OF_MEMORY;
    }
    e->live_comm.push_back(c);
    *out = c;
    return LPR_OK_OPTIMAL;
}

int lpr_comm_init_custom(int rank, int world, lpr_allreduce_max_fn all_reduce_max,
                         lpr_allgather_fn all_gather, void* user, lpr_comm** out) {
    if (!out || world < 1 || rank < 0 || rank >= world || !all_reduce_max || !all_gather) {
        set_error("lpr_comm_init_custom: bad arguments (rank %d of %d)", rank, world);
        return LPR_BAD_ARGUMENT;
    }
    lpr_comm* c = new (std::nothrow) lpr_comm();
    if (!c) return LPR_OUT_OF_MEMORY;
    c->rank = rank;
    c->world = world;
    c->ar = all_reduce_max;
    c->ag = all_gather;
    c->user = user;
    *out = c;
    return LPR_OK_OPTIMAL;
}

int lpr_comm_destroy(lpr_comm* c) {
    if (!c) return LPR_BAD_ARGUMENT;
    if (c->eng) {
        comm_release_device(c);
        auto& lv = c->eng->live_comm;
        for (size_t k = 0; k < lv.size(); ++k)
            if (lv[k] == c) {
                lv.erase(lv.begin() + k);
                break;
            }
    }
    delete c;
    return LPR_OK_OPTIMAL;
}

int lpr_comm_info(const lpr_comm* c, int* rank, int* world, int64_t* allreduce_calls,
                  int64_t* allgather_calls) {
    if (!c) return LPR_BAD_ARGUMENT;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (allreduce_calls) *allreduce_calls = c->allreduce_calls;
    if (allgather_calls) *allgather_calls = c->allgather_calls;
    return LPR_OK_OPTIMAL;
}

int lpr_comm_all_reduce_max(lpr_comm* c, double* inout, int count) {
    if (!c || !inout) return LPR_BAD_ARGUMENT;
    return comm_all_reduce_max(c, inout, count);
}

}  // extern "C"
