// common.hpp — shared host-side infrastructure of libspaghetti_rank (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/spaghetti_rank.h"

namespace ss {

void set_global_error(const std::string& msg);

}  // namespace ss

struct ss_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // stream all work is enqueued on
    hipStream_t own_stream = nullptr;  // created by ss_init
    hipStream_t comm_stream = nullptr; // the exchange steps of the sharded sweep run here, beside the sweeps on `stream`
    static constexpr int N_WAVE_STREAMS = 3;
    hipStream_t wave_stream[N_WAVE_STREAMS] = {};   // option "score.pipeline" (default on): k_wave_prep + k_score_wave of a batch run here; the batch's k_merge_flat
                                        // follows on `stream` behind an event, so the NEXT batch's k_score_wave starts under it and the hits are still
                                        // complete in the order of `stream`.  Consecutive batches take different wave streams: batch i+1's
                                        // k_score_wave then also fills the slots that batch i's tail leaves idle.
    std::recursive_mutex mu;           // serialises calls on this ctx
    std::string last_error;
    // timing hook (ss_last_kernel_ms): [kind][0]=start, [1]=stop
    hipEvent_t ev[3][2] = {};
    bool ev_valid[3] = {false, false, false};
    int cu_count = 256;
    size_t total_mem = 0;
    int tfidf_scatter_lds = 0;         // k_scatter's dynamic LDS size granted on this device
    int tfidf_bucket_lds = 0;          // dynamic LDS size already granted to k_bucket_sum on this device
    void* comm = nullptr;              // RCCL communicator of this rank (ss_comm_init), ncclComm_t
    void* comm_parent = nullptr;       // the communicator `comm` was split from (ss_comm_split)
    int comm_rank = 0, comm_world = 1;
    bool wedged = false;               // a bounded wait of this context timed out (comm.hip: device_wedged / try_unwedge)
    // tuning / diagnostic options (ss_set_option); the defaults live at the point of use
    // Pinned host memory for what comes back from the device.  hipMemcpyAsync into PAGEABLE host memory pins the caller's page on
    // the fly — an mm-lock round trip that took 5-10 ms PER COPY on a loaded 256-core host (ss_graph_create: 7 ms on a quiet box,
    // 20-40 ms there; the time sat in the enqueue of four 8-byte read-backs).  h_pin: a bump-allocated scratch for small results
    // (reset at the entry of the call that uses it); pin_cache: free blocks for larger host copies (the graph's in-degrees).
    unsigned char* h_pin = nullptr;
    size_t pin_used = 0;
    static constexpr size_t PIN_SCRATCH = 8192;
    struct PinBlock { void* p; size_t cap; };
    std::vector<PinBlock> pin_cache;
    template <typename T>
    T* pin(size_t count = 1) {                       // scratch for `count` T; callers ask for bytes, not kilobytes
        const size_t bytes = (count * sizeof(T) + 15) & ~(size_t)15;
        if (!h_pin || bytes > PIN_SCRATCH) {                              // a request the scratch cannot hold is a bug in the caller
            fprintf(stderr, "libspaghetti_rank: ss_ctx::pin(%zu bytes) exceeds the %zu-byte scratch\n", bytes, PIN_SCRATCH);
            abort();
        }
        if (pin_used + bytes > PIN_SCRATCH) pin_used = 0;                 // (wraps: nothing is held across calls)
        T* r = reinterpret_cast<T*>(h_pin + pin_used);
        pin_used += bytes;
        return r;
    }
    void* pin_alloc(size_t bytes, size_t* cap_out) {
        size_t best = pin_cache.size();
        for (size_t i = 0; i < pin_cache.size(); i++)
            if (pin_cache[i].cap >= bytes && (best == pin_cache.size() || pin_cache[i].cap < pin_cache[best].cap)) best = i;
        if (best < pin_cache.size()) {
            PinBlock b = pin_cache[best];
            pin_cache.erase(pin_cache.begin() + best);
            *cap_out = b.cap;
            return b.p;
        }
        void* q = nullptr;
        const size_t cap = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(&q, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        *cap_out = cap;
        return q;
    }
    void pin_free(void* q, size_t cap) {
        if (!q) return;
        if (pin_cache.size() < 4) pin_cache.push_back({q, cap});
        else (void)hipHostFree(q);
    }
    std::map<std::string, int64_t> options;
    int64_t opt(const char* name, int64_t dflt) const {
        auto it = options.find(name);
        return it == options.end() ? dflt : it->second;
    }

    int32_t fail(int32_t code, const char* fmt, ...) {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        last_error = buf;
        ss::set_global_error(last_error);
        return code;
    }
};

namespace ss {
// Small results back from the device: into the context's pinned scratch, wait, copy out (see ss_ctx::h_pin for why never straight
// into the caller's pageable variables).  Up to two values per call; waits for `st`.  The sources may be host or device memory
// (the ABI's "host OR device" input arrays): the copy is ordered on `st`, so a device array produced on the context's stream —
// what ss_set_stream's contract allows — is read after its producer (a null-stream hipMemcpy is not ordered behind a
// non-blocking stream).
inline hipError_t fetch(ss_ctx* ctx, hipStream_t st, void* d1, const void* s1, size_t n1, void* d2 = nullptr, const void* s2 = nullptr, size_t n2 = 0) {
    ctx->pin_used = 0;
    unsigned char* const p1 = ctx->pin<unsigned char>(n1);
    unsigned char* const p2 = n2 ? ctx->pin<unsigned char>(n2) : nullptr;
    hipError_t e = hipMemcpyAsync(p1, s1, n1, hipMemcpyDefault, st);
    if (e == hipSuccess && n2) e = hipMemcpyAsync(p2, s2, n2, hipMemcpyDefault, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) {
        std::memcpy(d1, p1, n1);
        if (n2) std::memcpy(d2, p2, n2);
    }
    return e;
}
// An input array of the ABI ("host OR device memory") copied to a host vector.  Host memory: a plain memcpy.  Device memory:
// ordered on `st` (the context's stream is non-blocking, so a null-stream hipMemcpy would not wait for a producer on it) and
// waited for.
inline hipError_t copy_in(hipStream_t st, void* dst, const void* src, size_t bytes) {
    if (!bytes) return hipSuccess;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, src) != hipSuccess) {              // plain malloc'ed memory on older runtimes
        (void)hipGetLastError();
        std::memcpy(dst, src, bytes);
        return hipSuccess;
    }
    if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) {
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
        return e == hipSuccess ? hipStreamSynchronize(st) : e;
    }
    std::memcpy(dst, src, bytes);
    return hipSuccess;
}
}  // namespace ss

#define SS_HIP(ctx, expr)                                                                  \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            return (ctx)->fail(_e == hipErrorOutOfMemory ? SS_ERR_OOM : SS_ERR_HIP,        \
                               "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,               \
                               hipGetErrorString(_e));                                     \
        }                                                                                  \
    } while (0)

#define SS_TRY(expr)                   \
    do {                               \
        int32_t _rc = (expr);          \
        if (_rc != SS_OK) return _rc;  \
    } while (0)

namespace ss {

// Device memory pool (ctx.hip): freed blocks are kept and handed out again, by size class.  One call of the offline entry
// points allocates and frees some thirty temporaries; hipMalloc + hipFree cost ~0.1 ms a pair and hipFree waits for the
// device, which is most of what a caller of ss_graph_create + ss_pagerank_run waited for on a small graph.  The pool holds at
// most 64 GiB (of 288; option "mem.pool_mb" on any context changes it for the process, 0 switches it off).
hipError_t pool_alloc(void** p, size_t bytes);
void pool_free(void* p);
void pool_free_batch(void* const* blocks, size_t count);   // one device-wide wait for all of them
void pool_set_limit(size_t bytes);
void pool_trim();
void pool_context_count(int delta);
void pool_stats(uint64_t* misses, double* miss_ms);      // really free everything the pool holds

// Device allocation that frees itself; raw pointers are handed to kernels.
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) pool_free(p);
        p = nullptr;
        n = 0;
    }
    void* detach() {                                  // the block changes owner (ss_graph::defer)
        void* q = p;
        p = nullptr;
        n = 0;
        return q;
    }
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) count = 1;
        return pool_alloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    }
    // streamed-once data: uncached memory type, so it does not occupy L2 lines
    hipError_t alloc_streaming(size_t count) {
#ifdef SS_UNCACHED_STREAMS
        release();
        n = count;
        if (count == 0) count = 1;
        return hipExtMallocWithFlags(reinterpret_cast<void**>(&p), count * sizeof(T), hipDeviceMallocUncached);
#else
        return alloc(count);
#endif
    }
    size_t bytes() const { return n * sizeof(T); }
};

inline unsigned div_up(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

// comm.hip: collectives on the context's stream (enqueue only); SS_ERR_STATE without a communicator
int32_t comm_allgather(ss_ctx* ctx, const void* send, void* recv, size_t bytes);
int32_t comm_allreduce_f64(ss_ctx* ctx, const double* send, double* recv, size_t count);
// wait for a stream that may carry a collective: bounded by option "comm.timeout_ms" when the context has a communicator
// (SS_ERR_COMM instead of a hang), a plain hipStreamSynchronize otherwise; device_wedged(dev): such a wait has timed out on that device
int32_t sync_bounded(ss_ctx* ctx, hipStream_t st, const char* what);
bool device_wedged(int device);
bool try_unwedge(ss_ctx* ctx);      // the context's streams have drained after a timed-out wait: its device counts as usable again
// the same on a given stream (the pipelined sharded sweep puts its exchanges on ctx->comm_stream)
int32_t comm_allgather_on(ss_ctx* ctx, const void* send, void* recv, size_t bytes, hipStream_t st);
int32_t comm_allreduce_f64_on(ss_ctx* ctx, const double* send, double* recv, size_t count, hipStream_t st);

}  // namespace ss
