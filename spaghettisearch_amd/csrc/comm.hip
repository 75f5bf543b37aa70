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

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <thread>

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
// ---- bounded waits --------------------------------------------------------------------------------------------------------
// A collective that a rank never joins, or that runs over a dead link, never completes: the stream behind it never drains and a
// plain hipStreamSynchronize turns a rendezvous mistake into a hang that only a watchdog ends.  Wherever the library itself waits
// for a stream that may carry a collective it polls instead, for at most "comm.timeout_ms" (default 120 s), and then fails with
// SS_ERR_COMM.  After such a time-out THAT device still holds the stuck work: its bit in `g_wedged` makes the memory pool and the
// sharded loop's clean-up skip their waits for that device (they would hang in turn), so that the error reaches the caller, who can
// report it and end the process.  Other devices of the process (in-process shard groups, the tests' contexts) are not affected:
// their blocks go back to the pool and their communicators are destroyed in the ordinary way (the flag used to be one
// for the whole process and was never cleared).
// The state is cleared again when the context whose wait timed out finds its streams drained after all (a slow rank that did
// join in the end): try_unwedge, called by its next bounded wait and by ss_shutdown.
static std::atomic<int> g_wedged[64];       // contexts with a timed-out wait, per device
bool device_wedged(int device) { return device >= 0 && device < 64 && g_wedged[device].load(std::memory_order_relaxed) > 0; }
static void mark_wedged(ss_ctx* ctx) {
    if (!ctx->wedged && ctx->device >= 0 && ctx->device < 64) {
        ctx->wedged = true;
        g_wedged[ctx->device].fetch_add(1);
    }
}
bool try_unwedge(ss_ctx* ctx) {
    if (!ctx->wedged) return true;
    for (hipStream_t st : {ctx->stream, ctx->comm_stream})
        if (st && hipStreamQuery(st) != hipSuccess) { (void)hipGetLastError(); return false; }
    ctx->wedged = false;
    g_wedged[ctx->device].fetch_sub(1);
    return true;
}

int32_t sync_bounded(ss_ctx* ctx, hipStream_t st, const char* what) {
    if (!ctx->comm) {                                        // nothing on this context can wait for another rank
        SS_HIP(ctx, hipStreamSynchronize(st));
        return SS_OK;
    }
    const int64_t timeout_ms = ctx->opt("comm.timeout_ms", 120000);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) { (void)try_unwedge(ctx); return SS_OK; }
        (void)hipGetLastError();
        if (e != hipErrorNotReady) return ctx->fail(SS_ERR_HIP, "%s: hipStreamQuery -> %s", what, hipGetErrorString(e));
        if (spins > 4096) std::this_thread::sleep_for(std::chrono::microseconds(50));     // the first milliseconds: spin
        if ((spins & 255) == 255 &&
            std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms) {
            mark_wedged(ctx);
            return ctx->fail(SS_ERR_COMM, "%s: rank %d of %d: the stream did not drain within %lld ms with a collective on it — a rank never "
                             "joined the collective or a link is down; this context cannot be used any more", what, ctx->comm_rank, ctx->comm_world,
                             (long long)timeout_ms);
        }
    }
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
    // ncclCommInitRank blocks until every rank has joined, with no time-out of its own: a rank that never arrives (wrong id,
    // wrong rank/world, a crashed peer, an unreachable bootstrap address) would hang every other rank for good.  The call runs
    // on a helper thread and this thread waits for it for at most "comm.timeout_ms" (default 120 s); after that the caller gets
    // SS_ERR_COMM and the helper is left behind (it owns everything it touches; should it ever finish, it aborts the
    // communicator nobody is waiting for).  The blocking call itself is RCCL's ordinary, most travelled initialisation path —
    // the non-blocking configuration (ncclCommInitRankConfig + ncclCommGetAsyncError) would also turn every later collective
    // into a poll loop.
    struct Job {
        std::mutex mu;
        std::condition_variable cv;
        bool done = false, abandoned = false;
        ncclResult_t r = ncclSuccess;
        ncclComm_t comm = nullptr;
    };
    auto job = std::make_shared<Job>();
    const int device = ctx->device;
    std::thread([job, uid, rank, world, device] {
        (void)hipSetDevice(device);
        ncclComm_t c = nullptr;
        const ncclResult_t r = ncclCommInitRank(&c, world, uid, rank);
        std::unique_lock<std::mutex> lk(job->mu);
        if (job->abandoned) {
            lk.unlock();
            if (r == ncclSuccess && c) (void)ncclCommAbort(c);
            return;
        }
        job->r = r;
        job->comm = c;
        job->done = true;
        job->cv.notify_all();
    }).detach();
    const int64_t timeout_ms = ctx->opt("comm.timeout_ms", 120000);
    {
        std::unique_lock<std::mutex> lk(job->mu);
        if (!job->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return job->done; })) {
            job->abandoned = true;
            return ctx->fail(SS_ERR_COMM, "ss_comm_init: rank %d of %d: the other ranks did not join within %lld ms (every rank needs rank 0's id, "
                             "its own rank, the same world size, and a reachable bootstrap address; multi-process GPU work on this host also "
                             "needs HSA_ENABLE_IPC_MODE_LEGACY=0)", rank, world, (long long)timeout_ms);
        }
    }
    if (job->r != ncclSuccess) return ss::comm_fail(ctx, "ncclCommInitRank", job->r);
    ncclComm_t comm = job->comm;
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
    const bool wedged = ss::device_wedged(ctx->device);                 // a collective timed out: nothing on the streams will ever finish
    if (!wedged) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->comm_stream);
    }
    ncclResult_t r = wedged ? ncclCommAbort(static_cast<ncclComm_t>(ctx->comm)) : ncclCommDestroy(static_cast<ncclComm_t>(ctx->comm));
    if (ctx->comm_parent) {
        const ncclResult_t r2 = wedged ? ncclCommAbort(static_cast<ncclComm_t>(ctx->comm_parent)) : ncclCommDestroy(static_cast<ncclComm_t>(ctx->comm_parent));
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
    SS_TRY(ss::sync_bounded(ctx, ctx->stream, "ss_comm_allreduce_u64"));
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
    SS_TRY(ss::sync_bounded(ctx, ctx->stream, "ss_comm_allgather"));
    return SS_OK;
}

}  // extern "C"
