// ctx.hip — context, stream and error plumbing of the C ABI (include/spaghetti_rank.h).
#include "common.hpp"
#include <chrono>

#include <unordered_map>

namespace ss {

// ---- device memory pool ------------------------------------------------------------------------------------------
namespace {
struct Pool {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void*> free_blocks;        // (device, size class) -> block
    std::unordered_map<void*, std::pair<int, size_t>> live;           // block -> (device, size class) of the blocks handed out
    // default limit: 64 GiB of the 288 GB — the library's own working set at the benchmark sizes is ~20 GB, and a limit that the
    // working set exceeds turns every release into an eviction and every large allocation into a fresh hipMalloc (the scorer's
    // 8 GB of combined lists took 0.96 s to create that way, 40 ms from the pool); hipMalloc failures trim the pool and retry
    size_t held = 0, limit = (size_t)64 << 30;   // (all fields: under mu)
    double miss_ms = 0.0;            // time spent in hipMalloc on pool misses (pr.trace prints it)
    uint64_t misses = 0;
    int contexts = 0;                // live ss_ctx of the process: the last ss_shutdown gives the held blocks back
};
Pool& pool() { static Pool* p = new Pool(); return *p; }   // never destroyed: DevBufs of static objects may outlive main
// size classes: powers of two below 1 MiB, then eighths of the power of two (at most 12.5 % over)
size_t size_class(size_t b) {
    if (b <= 4096) return 4096;
    size_t p2 = 4096;
    while (p2 < b) p2 <<= 1;
    if (p2 <= ((size_t)1 << 20)) return p2;
    const size_t step = p2 >> 4;                         // p2/2 < b <= p2
    return (b + step - 1) / step * step;
}
}  // namespace

hipError_t pool_alloc(void** out, size_t bytes) {
    Pool& P = pool();
    const size_t cls = size_class(bytes);
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.free_blocks.find({dev, cls});
        if (it != P.free_blocks.end()) {
            *out = it->second;
            P.free_blocks.erase(it);
            P.held -= cls;
            P.live[*out] = {dev, cls};
            return hipSuccess;
        }
    }
    const auto tm0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(out, cls);
    {
        std::lock_guard<std::mutex> lk(P.mu);
        P.miss_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tm0).count();
        P.misses++;
    }
    if (e != hipSuccess) {                               // give the pool's memory back and try once more
        (void)hipGetLastError();
        pool_trim();
        e = hipMalloc(out, cls);
    }
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(P.mu);
        P.live[*out] = {dev, cls};
    }
    return e;
}

static thread_local bool t_device_idle = false;     // pool_free_batch: one device-wide wait covers the whole batch

void pool_free_batch(void* const* blocks, size_t count) {
    if (!count) return;
    {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (device_wedged(cur)) return;
    }
    // (one device at a time: the blocks of a batch belong to one graph)
    (void)hipDeviceSynchronize();
    t_device_idle = true;
    for (size_t i = 0; i < count; i++) pool_free(blocks[i]);
    t_device_idle = false;
}

void pool_free(void* p) {
    if (!p) return;
    Pool& P = pool();
    size_t cls = 0, limit = 0;
    int dev = 0;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.live.find(p);
        if (it != P.live.end()) { dev = it->second.first; cls = it->second.second; P.live.erase(it); }
        limit = P.limit;
    }
    // hipFree waits for the device before the memory goes; the pool keeps that guarantee (work of ANY stream that still uses
    // the block is over before somebody else can get it) and saves the unmap / map that follows.
    if (device_wedged(dev)) return;                        // a collective timed out on THIS device: a device-wide wait would hang; the block leaks
    if (cls && cls <= limit / 2) {                         // a block larger than half the limit is not worth holding
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != dev || !t_device_idle) {
            if (cur != dev) (void)hipSetDevice(dev);
            (void)hipDeviceSynchronize();
            if (cur != dev) (void)hipSetDevice(cur);
        }
        // over the limit: the LARGEST held blocks make room (a few big tables of an earlier, larger job must not keep every
        // small block of the current one out of the pool: that cost config 2's create + run 2 ms after config 4 had run)
        std::vector<void*> evict;
        {
            std::lock_guard<std::mutex> lk(P.mu);
            while (P.held + cls > P.limit && !P.free_blocks.empty()) {
                auto last = std::prev(P.free_blocks.end());
                if (last->first.second <= cls) break;                   // nothing larger than the newcomer is held
                evict.push_back(last->second);
                P.held -= last->first.second;
                P.free_blocks.erase(last);
            }
            if (P.held + cls <= P.limit) {
                P.free_blocks.emplace(std::make_pair(dev, cls), p);
                P.held += cls;
                p = nullptr;
            }
        }
        for (void* q : evict) (void)hipFree(q);
        if (!p) return;
    }
    (void)hipFree(p);
}

void pool_stats(uint64_t* misses, double* miss_ms) {
    std::lock_guard<std::mutex> lk(pool().mu);
    *misses = pool().misses;
    *miss_ms = pool().miss_ms;
}
// ss_init / ss_shutdown: when the last context of the process goes, so do the blocks the pool holds (a crawler or torch in the
// same process, or beside it on the same GPU, must not find 64 GiB parked by a library nobody is using)
void pool_context_count(int delta) {
    bool last = false;
    {
        std::lock_guard<std::mutex> lk(pool().mu);
        pool().contexts += delta;
        last = delta < 0 && pool().contexts <= 0;
    }
    if (last) pool_trim();
}

void pool_trim() {
    Pool& P = pool();
    std::multimap<std::pair<int, size_t>, void*> take;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        take.swap(P.free_blocks);
        P.held = 0;
    }
    for (auto& kv : take) (void)hipFree(kv.second);
}

void pool_set_limit(size_t bytes) {
    {
        std::lock_guard<std::mutex> lk(pool().mu);
        pool().limit = bytes;
    }
    pool_trim();
}

static std::mutex g_err_mu;
static std::string g_err;
void set_global_error(const std::string& msg) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err = msg;
}
}  // namespace ss

extern "C" {

int32_t ss_abi_version(void) { return SS_ABI_VERSION; }

int32_t ss_init(int32_t device_id, ss_ctx** out) {
    if (!out) {
        ss::set_global_error("ss_init: out is NULL");
        return SS_ERR_INVALID;
    }
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        // No CPU fallback: the product path requires a HIP device.
        ss::set_global_error(std::string("ss_init: no HIP device (") + hipGetErrorString(e) + ")");
        return SS_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n_dev) {
        ss::set_global_error("ss_init: device_id out of range");
        return SS_ERR_INVALID;
    }
    ss_ctx* ctx = new (std::nothrow) ss_ctx();
    if (!ctx) return SS_ERR_OOM;
    ctx->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess) {
        ss::set_global_error(std::string("ss_init: hipSetDevice: ") + hipGetErrorString(e));
        delete ctx;
        return SS_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) {
        ss::set_global_error(std::string("ss_init: hipGetDeviceProperties: ") + hipGetErrorString(e));
        delete ctx;
        return SS_ERR_NO_DEVICE;
    }
    // Kernels are built for gfx950 only (wave64, 160 KiB LDS, ds_add_f64).
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        ss::set_global_error(std::string("ss_init: device is ") + prop.gcnArchName +
                             ", this library is built for gfx950 (MI355X) only");
        delete ctx;
        return SS_ERR_NO_DEVICE;
    }
    ctx->cu_count = prop.multiProcessorCount;
    ctx->total_mem = prop.totalGlobalMem;
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        ss::set_global_error(std::string("ss_init: hipStreamCreate: ") + hipGetErrorString(e));
        delete ctx;
        return SS_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    // highest priority: what runs there is short and somebody waits for it (the next batch's plan upload and k_wave_prep beside the
    // current batch's kernels, a topic block's exchange beside the next block's sweep) — at equal priority its workgroups queue
    // behind every workgroup of the long kernel that was launched first
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if ((e = hipStreamCreateWithPriority(&ctx->comm_stream, hipStreamNonBlocking, prio_hi)) != hipSuccess) {
        ss::set_global_error(std::string("ss_init: hipStreamCreate (comm): ") + hipGetErrorString(e));
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return SS_ERR_HIP;
    }
    for (int k = 0; k < 3; k++)
        for (int j = 0; j < 2; j++)
            if ((e = hipEventCreate(&ctx->ev[k][j])) != hipSuccess) {
                ss::set_global_error(std::string("ss_init: hipEventCreate: ") + hipGetErrorString(e));
                delete ctx;
                return SS_ERR_HIP;
            }
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pin), ss_ctx::PIN_SCRATCH, hipHostMallocDefault)) != hipSuccess) {
        ss::set_global_error(std::string("ss_init: hipHostMalloc: ") + hipGetErrorString(e));
        delete ctx;
        return SS_ERR_HIP;
    }
    ss::pool_context_count(+1);
    *out = ctx;
    return SS_OK;
}

int32_t ss_shutdown(ss_ctx* ctx) {
    if (!ctx) return SS_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    const bool stuck = !ss::try_unwedge(ctx);              // a collective timed out and its stream still has not drained: wait for nothing
    (void)ss_comm_destroy(ctx);
    if (!stuck) (void)hipStreamSynchronize(ctx->stream);
    for (int k = 0; k < 3; k++)
        for (int j = 0; j < 2; j++)
            if (ctx->ev[k][j]) (void)hipEventDestroy(ctx->ev[k][j]);
    if (ctx->comm_stream && !stuck) { (void)hipStreamSynchronize(ctx->comm_stream); (void)hipStreamDestroy(ctx->comm_stream); }
    for (hipStream_t ws : ctx->wave_stream)
        if (ws && !stuck) { (void)hipStreamSynchronize(ws); (void)hipStreamDestroy(ws); }
    if (ctx->own_stream && !stuck) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    for (auto& b : ctx->pin_cache) (void)hipHostFree(b.p);
    delete ctx;
    ss::pool_context_count(-1);
    return SS_OK;
}

int32_t ss_set_stream(ss_ctx* ctx, void* hip_stream) {
    if (!ctx) return SS_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return SS_OK;
}

// Every option a call site reads; ss_set_option refuses anything else, so a typo cannot silently do nothing.
static const char* const k_option_names[] = {
    "comm.timeout_ms",      // longest wait for the other ranks: ss_comm_init, and every wait of the library for a stream that carries a collective (default 120000)
    "pr.force_narrow",      // 1: K <= 2 always runs the block-item kernel k_pr_step (tests reach it on small graphs)
    "pr.wire_f32",          // 1: the doc-range-sharded sweep exchanges its contribution slices as float32 (half the bytes per link; inside the 1e-6 gate, not the
                            //    reference's float64 arithmetic: opt-in, reported under bench.py's `decompositions` only).  Every rank must set it alike.
    "pr.narrow_wave",       // 0: K <= 2 as before round 4 (padded to the 8-wide sweep on small graphs, k_pr_step on large ones); default 1: k_pr_sweep_n
    "pr.t_quad",            // in-degree above which a row gets a wave of its own in k_pr_sweep (default 128 — with the class stagger 96 .. 192 measure alike, 256 0.5 % slower —; k_pr_sweep_n 256)
    "pr.blocks_per_cu",     // resident workgroups per CU of the sweep grid (default: the occupancy query)
    "mem.pool_mb",          // MiB of freed device blocks the library keeps for reuse (process-wide; default 65536, 0 = off)
    "pr.deal_snake",        // work items dealt to the waves in alternating direction (1) or least-loaded-first (0); default: 1 from 8 items per wave on and for k_pr_sweep_n
    "pr.affine_lag",        // two-vector form on shards: default 1 = ONE collective per iteration (the per-topic sums ride in spare tail rows of the next
                            //    iteration's all-gather, stop decisions one exchange late); 0 = a second, small all-gather per iteration (round 4)
    "pr.affine",            // 1: ss_pagerank_run computes every topic from TWO vectors (the reference's topics differ only in their start value 1/n_k,
                            //    and its recurrence maps (p*u + q) / (r*u + s) onto itself): opt-in, not the reference's operation order (~1e-13)
    "pr.items_per_wave",    // k_pr_sweep_n (K <= 2): the grid is cut so that every wave gets at least this many work items (default 4: a small graph's sweep costs per wave)
    "pr.persistent",        // K <= 2 on one rank: ss_pr_step's sweeps run inside ONE launch, the blocks waiting for each other between sweeps (k_pr_multi_n):
                            //    1 = write-through hand-offs, 2 = release / acquire fences.  Default 0 (measured slower than one launch per sweep: DESIGN K1c)
    "pr.persistent_blocks", // ... resident blocks per CU of that launch (default 4, at most half of what the occupancy query admits)
    "pr.n_class_order",     // k_pr_sweep_n (K <= 2): the order of its four phases, four digits (0 = long rows, 1 = mid rows, 2 = rows of <= 8 in-edges, 3 = edge-less rows)
    "pr.class_order",       // k_pr_sweep: the order in which a wave walks its work classes, six decimal digits naming the classes 0 = long rows, 1 = mid rows, 2 / 3 / 4 = rows of <= 2 / 4 / 8 in-edges, 5 = edge-less rows (default 235401: short rows first)
    "pr.stagger",           // default 0: every block of k_pr_sweep walks the work classes in the same order ("pr.class_order"); 1: the resident blocks of a CU
                            //    start at different positions of it (by arrival round; round 4's default); >= 10: 10 + the rounds' start positions as base-6 digits
    "pr.deal_global",       // 0: the work items are dealt chunk by chunk in table order, each chunk sorted by cost (before round 4); 1: all items by
                            //    falling cost first (one counting sort; default for k_pr_sweep_n<1>); 2: by falling cost inside each class, the classes in
                            //    table order (default for k_pr_sweep and k_pr_sweep_n<2>)
    "pr.item_turns",        // turns per V_DEG work item of k_pr_sweep (V_QUAD: twice that); default 4 up to 4M local rows, 8 beyond
    "graph.late_free",      // 0: ss_graph_create waits for its last kernels and frees its temporaries before it returns (default 1: they are freed
                            //    at the graph's next use, the caller's host work overlaps the row permutation)
    "pr.trace",             // 1: ss_graph_create / ss_pr_create print their phase times to stderr
    "score.collect_pinned", // ss_score_topk_collect: 1 = device -> pinned block on the copy engine, then a host memcpy (measured slower: 0.50 against 0.41 ms
                            //    per batch with three in flight); default 0 = hipMemcpy straight into the caller's memory
    "score.pipeline_slices",// 0: a batch that is all k_score_slices runs on the caller's stream with the fused merge (before late round 4); default 1: its
                            //    scoring kernel on an internal stream, k_merge_topk on the caller's stream behind an event (device outputs only)
    "score.timing",         // 0: ss_score_topk records no timing events (ss_last_kernel_ms(1) keeps its last value)
    "score.trace",          // 1: ss_score_topk prints the host phases of a call (copies in, plan, staging, launches) to stderr
    "pr.probe_hot",         // ss_pr_probe policies 3/4: rows below this index use the default cache policy
    "pr.topic_blocks",      // ss_pagerank_run_sharded: split K into this many topic blocks whose exchanges overlap the next block's sweep
    "tfidf.fused",          // 0: weight + count pass, then a scatter over the weighted postings (round 3); default 1: count pass over the doc ids, weights multiplied inside the scatter
    "tfidf.blocks",         // workgroups of the bucketed magnitude pass (default 4096)
    "tfidf.bucket_shift",   // log2 docs per bucket (default 13, 14 beyond 33M docs)
    "tfidf.head_min_run",   // head lists (summed bucket-major in place): average postings per bucket run, 0 = off (default 64)
    "tfidf.bucket_min",     // smallest table (postings) that takes the bucketed pass (default 4M)
    "score.wave",           // 0: never use the wave-per-slice kernel k_score_wave
    "score.wave_max_terms", // a query suits k_score_wave if it has at most this many terms (default 6, at most 12)
    "score.wave_min_list",  // a query suits k_score_wave if EVERY list of it has this many x k' postings (k' = k rounded up to 2^j; default 16)
    "score.wave_share_pct", // a batch uses k_score_wave if the queries that suit it carry at least this share of the batch's postings (default 90)
    "score.wave_slice_target", // postings per slice of k_score_wave (default: from the batch, 8k .. 48k)
    "score.wave_big_pct",   // graded slices: this share of a batch's postings goes into slices of wave_big_x100 % of the target, the rest into wave_small_x100 % (defaults 92 / 115 / 60; 0 = one size)
    "score.wave_big_x100",
    "score.small",          // k_score_small (one workgroup per query, every posting exactly, the hits written by the kernel itself) takes queries without a
                            //    phrase part, with at most "score.small_cap" postings (default and most: 1664) in at most 16 non-empty lists, k <= 256:
                            //    2 (default): only in calls of at most "score.small_max_batch" queries that are ALL such queries (one launch instead
                            //    of a slices kernel and a merge: a lone tail query 0.068 -> 0.050 ms host to host); 1: every such query of every call
                            //    (tests, A/B: a 1024-query tail batch is slower that way, 0.14 against 0.09 ms); 0: never.  Bit-identical hits (DESIGN K4c)
    "score.small_max_batch",// "score.small" = 2: longest call (queries) that may take k_score_small (default 64)
    "score.small_batch",    // 1 (with "score.small" = 2): longer calls with device outputs send their small queries to k_score_small too, on an internal stream, rows staged (default 0)
    "score.debug_floor",    // EXPERIMENT, only in a library built with -DSS_EXP_FLOOR (tools/floor_exp.py; no effect in the product): 1 = every host-output call records its
                            //    queries' k-th best FinalRank, and the next call of as many queries starts its filters from them (wrong hits if the batch changes)
    "score.small_cap",
    "score.pipeline",       // 0: every scoring kernel on the context's stream.  n >= 1: device-output batches that are all k_score_wave run k_wave_prep and
                            //    k_score_wave on one of n internal streams taken in turn (default 2, as include/spaghetti_rank.h says; at most 3) and
                            //    only k_merge_flat (behind an event) on the context's stream: the next batch's
                            //    k_score_wave starts under this batch's merge, and the hits are complete in stream order as before
    "score.grade_slices",   // 1: k_score_slices' slices are graded the same way (default 0)
    "score.wave_small_x100",
    "score.exact_all",      // 1: switch the upper-bound filter off (every record takes the exact stage)
    "score.slice_target",   // postings per (query, slice) workgroup (default: from the batch)
    "score.separate_merge", // 1: the per-query merge runs as its own launch (k_merge_topk)
};

int32_t ss_set_option(ss_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return SS_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    for (const char* n : k_option_names)
        if (std::strcmp(n, name) == 0) {
            if (value == SS_OPTION_DEFAULT) ctx->options.erase(name);
            else ctx->options[name] = value;
            if (std::strcmp(name, "mem.pool_mb") == 0) ss::pool_set_limit(value == SS_OPTION_DEFAULT ? (size_t)64 << 30 : (size_t)std::max<int64_t>(0, value) << 20);
            return SS_OK;
        }
    return ctx->fail(SS_ERR_INVALID, "ss_set_option: unknown option '%s'", name);
}

int32_t ss_synchronize(ss_ctx* ctx) {
    if (!ctx) return SS_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    SS_TRY(ss::sync_bounded(ctx, ctx->stream, "ss_synchronize"));       // (the wave stream's kernels are all in front of a merge on this stream)
    return SS_OK;
}

const char* ss_last_error(ss_ctx* ctx) {
    if (ctx) return ctx->last_error.c_str();
    // global: copy into a thread-local buffer so the pointer stays valid
    static thread_local std::string tl;
    {
        std::lock_guard<std::mutex> lk(ss::g_err_mu);
        tl = ss::g_err;
    }
    return tl.c_str();
}

int32_t ss_last_kernel_ms(ss_ctx* ctx, int32_t kind, float* ms_out) {
    if (!ctx || !ms_out || kind < 0 || kind > 2) return SS_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!ctx->ev_valid[kind]) return ctx->fail(SS_ERR_STATE, "ss_last_kernel_ms: no timed call of kind %d yet", kind);
    SS_HIP(ctx, hipEventSynchronize(ctx->ev[kind][1]));
    SS_HIP(ctx, hipEventElapsedTime(ms_out, ctx->ev[kind][0], ctx->ev[kind][1]));
    return SS_OK;
}

}  // extern "C"
