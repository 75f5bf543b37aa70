// comm.hip — the multi-GPU exchange steps of the hot path behind the C ABI: RCCL over xGMI, one ss_ctx (= one GPU) per rank.
//
// The reference is single-process, single-threaded (ranking/pagerank.go:52 leaves even the topic loop sequential); the north
// star shards the doc range of the PageRank SpMV across the GPUs of a node with one collective per iteration.  With these
// entry points a host in any language (the cgo shim in go/, a C program, the Python test driver) runs the sharded sweep
// without a second communication stack: every collective is enqueued on the context's stream by the library itself.
//
//   rank 0:      ss_comm_unique_id(id)            -> hand the 128 bytes to every rank (file, pipe, env, MPI, ...)
//   every rank:  ss_comm_init(ctx, id, rank, world)
//                ss_graph_create(ctx, ..., rank, world) ; ss_pagerank_run_sharded(...)     (or the step-wise ss_pr_* + ss_pr_exchange)
//                ss_comm_destroy(ctx)
// A process may also hold several contexts (one per device) and call ss_comm_init for each from its own thread.
#include "common.hpp"

#include <rccl/rccl.h>

static_assert(SS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ss_comm_unique_id hands out an ncclUniqueId");

namespace ss {

int32_t comm_fail(ss_ctx* ctx, const char* what, ncclResult_t r) {
    return ctx->fail(SS_ERR_COMM, "%s: %s", what, ncclGetErrorString(r));
}
#define SS_NCCL(ctx, expr)                                              \
    do {                                                                \
        ncclResult_t _r = (expr);                                       \
        if (_r != ncclSuccess) return ss::comm_fail((ctx), #expr, _r);  \
    } while (0)

// all-gather of `bytes` per rank, device buffers, on the context's stream (enqueue only)
int32_t comm_allgather_on(ss_ctx* ctx, const void* send, void* recv, size_t bytes, hipStream_t st) {
    if (!ctx->comm) return ctx->fail(SS_ERR_STATE, "no communicator: call ss_comm_init first");
    SS_NCCL(ctx, ncclAllGather(send, recv, bytes, ncclChar, static_cast<ncclComm_t>(ctx->comm), st));
    return SS_OK;
}
int32_t comm_allreduce_f64_on(ss_ctx* ctx, const double* send, double* recv, size_t count, hipStream_t st) {
    if (!ctx->comm) return ctx->fail(SS_ERR_STATE, "no communicator: call ss_comm_init first");
    SS_NCCL(ctx, ncclAllReduce(send, recv, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(ctx->comm), st));
    return SS_OK;
}
int32_t comm_allgather(ss_ctx* ctx, const void* send, void* recv, size_t bytes) { return comm_allgather_on(ctx, send, recv, bytes, ctx->stream); }
int32_t comm_allreduce_f64(ss_ctx* ctx, const double* send, double* recv, size_t count) { return comm_allreduce_f64_on(ctx, send, recv, count, ctx->stream); }

}  // namespace ss

extern "C" {

int32_t ss_comm_unique_id(void* id_out) {
    if (!id_out) return SS_ERR_INVALID;
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) {
        ss::set_global_error(std::string("ss_comm_unique_id: ") + ncclGetErrorString(r));
        return SS_ERR_COMM;
    }
    std::memcpy(id_out, &id, sizeof(id));
    return SS_OK;
}

int32_t ss_comm_init(ss_ctx* ctx, const void* id, int32_t rank, int32_t world) {
    if (!ctx) return SS_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!id || world < 1 || rank < 0 || rank >= world) return ctx->fail(SS_ERR_INVALID, "ss_comm_init: id NULL or rank/world out of range");
    if (ctx->comm) return ctx->fail(SS_ERR_STATE, "ss_comm_init: this context already has a communicator");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    SS_NCCL(ctx, ncclCommInitRank(&comm, world, uid, rank));      // blocks until every rank has joined
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return SS_OK;
}

// 2-D decomposition of the sharded sweep (topic groups x doc shards): the `world` ranks split into groups by `color`; inside
// a group the ranks are renumbered by `key`.  The context's communicator becomes the group's (ss_comm_info then reports the
// rank and size inside the group); the parent is kept and released by ss_comm_destroy.  Collective over the parent.
int32_t ss_comm_split(ss_ctx* ctx, int32_t color, int32_t key) {
    if (!ctx) return SS_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!ctx->comm) return ctx->fail(SS_ERR_STATE, "ss_comm_split: no communicator");
    if (ctx->comm_parent) return ctx->fail(SS_ERR_STATE, "ss_comm_split: already split");
    if (color < 0 || key < 0) return ctx->fail(SS_ERR_INVALID, "ss_comm_split: color and key must be >= 0");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    ncclComm_t sub = nullptr;
    SS_NCCL(ctx, ncclCommSplit(static_cast<ncclComm_t>(ctx->comm), color, key, &sub, nullptr));
    int r = 0, w = 1;
    SS_NCCL(ctx, ncclCommUserRank(sub, &r));
    SS_NCCL(ctx, ncclCommCount(sub, &w));
    ctx->comm_parent = ctx->comm;
    ctx->comm = sub;
    ctx->comm_rank = r;
    ctx->comm_world = w;
    return SS_OK;
}

int32_t ss_comm_destroy(ss_ctx* ctx) {
    if (!ctx) return SS_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!ctx->comm) return SS_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->comm_stream);
    ncclResult_t r = ncclCommDestroy(static_cast<ncclComm_t>(ctx->comm));
    if (ctx->comm_parent) {
        const ncclResult_t r2 = ncclCommDestroy(static_cast<ncclComm_t>(ctx->comm_parent));
        if (r == ncclSuccess) r = r2;
        ctx->comm_parent = nullptr;
    }
    ctx->comm = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
    if (r != ncclSuccess) return ss::comm_fail(ctx, "ncclCommDestroy", r);
    return SS_OK;
}

int32_t ss_comm_info(ss_ctx* ctx, int32_t* rank_out, int32_t* world_out) {
    if (!ctx) return SS_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (rank_out) *rank_out = ctx->comm ? ctx->comm_rank : -1;
    if (world_out) *world_out = ctx->comm ? ctx->comm_world : 0;
    return SS_OK;
}

// Generic collectives of the two index-side exchange steps: whole-corpus document
// frequencies = all-reduce(sum) of the shards' list lengths; corpus top-k = all-gather of the shards' hit lists.
// Buffers may be host or device memory; the call returns when the result is in `recv`/`buf`.
int32_t ss_comm_allreduce_u64(ss_ctx* ctx, uint64_t* buf, uint64_t n) {
    if (!ctx) return SS_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!ctx->comm) return ctx->fail(SS_ERR_STATE, "ss_comm_allreduce_u64: no communicator");
    if (!buf && n) return ctx->fail(SS_ERR_INVALID, "ss_comm_allreduce_u64: buf is NULL");
    if (n == 0) return SS_OK;
    SS_HIP(ctx, hipSetDevice(ctx->device));
    ss::DevBuf<uint64_t> tmp;
    SS_HIP(ctx, tmp.alloc(n));
    SS_HIP(ctx, hipMemcpyAsync(tmp.p, buf, n * sizeof(uint64_t), hipMemcpyDefault, ctx->stream));
    SS_NCCL(ctx, ncclAllReduce(tmp.p, tmp.p, n, ncclUint64, ncclSum, static_cast<ncclComm_t>(ctx->comm), ctx->stream));
    SS_HIP(ctx, hipMemcpyAsync(buf, tmp.p, n * sizeof(uint64_t), hipMemcpyDefault, ctx->stream));
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

int32_t ss_comm_allgather(ss_ctx* ctx, const void* send, void* recv, uint64_t bytes_per_rank) {
    if (!ctx) return SS_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!ctx->comm) return ctx->fail(SS_ERR_STATE, "ss_comm_allgather: no communicator");
    if ((!send || !recv) && bytes_per_rank) return ctx->fail(SS_ERR_INVALID, "ss_comm_allgather: NULL buffer");
    if (bytes_per_rank == 0) return SS_OK;
    SS_HIP(ctx, hipSetDevice(ctx->device));
    // stage through device memory unless both buffers already live there
    hipPointerAttribute_t a1{}, a2{};
    const bool d1 = hipPointerGetAttributes(&a1, send) == hipSuccess && a1.type == hipMemoryTypeDevice;
    const bool d2 = hipPointerGetAttributes(&a2, recv) == hipSuccess && a2.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    if (d1 && d2) {
        SS_TRY(ss::comm_allgather(ctx, send, recv, bytes_per_rank));
        return SS_OK;                                      // device buffers: ordered on the context's stream, no wait
    }
    ss::DevBuf<unsigned char> ds, dr;
    SS_HIP(ctx, ds.alloc(bytes_per_rank));
    SS_HIP(ctx, dr.alloc(bytes_per_rank * (size_t)ctx->comm_world));
    SS_HIP(ctx, hipMemcpyAsync(ds.p, send, bytes_per_rank, hipMemcpyDefault, ctx->stream));
    SS_TRY(ss::comm_allgather(ctx, ds.p, dr.p, bytes_per_rank));
    SS_HIP(ctx, hipMemcpyAsync(recv, dr.p, bytes_per_rank * (size_t)ctx->comm_world, hipMemcpyDefault, ctx->stream));
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

}  // extern "C"
