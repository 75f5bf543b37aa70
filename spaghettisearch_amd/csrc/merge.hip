// merge.hip — combine the per-shard top-k lists of a doc-range-sharded index (scoring row).
//
// Every shard (one GPU, one contiguous doc range) returns its own top-k per query in the library's result
// order; the k best of the union are the k best of the whole corpus, because a doc's postings all live in
// its shard.  The host gathers the lists (one all-gather of n_q*k ss_hit per rank) and calls ss_merge_hits,
// which reproduces the order of appendSort (retrieval/util.go:48-54) + the cut (main_retrieve.go:99-103)
// over the union.  Doc ids of shard p are shifted by doc_base[p] back to corpus ids.
//
// Kernel: one workgroup per query; each input row finds its output position directly — its index in its own
// (already ordered) list plus, for every other shard, the number of rows there that precede it (binary
// search).  Rows are distinct docs, so the order is total and positions never collide.  HBM-bound on
// n_parts*k*40 B per query; no LDS, no atomics.
#include "common.hpp"
#include "order.hpp"

namespace {

constexpr int TPB = 256;

__global__ __launch_bounds__(TPB) void k_merge_hits(int32_t n_q, int32_t n_parts, int32_t k, const ss_hit* __restrict__ parts,
                                                    const int32_t* __restrict__ n_hits, const uint32_t* __restrict__ doc_base,
                                                    ss_hit* __restrict__ out, int32_t* __restrict__ n_out) {
    const int32_t q = blockIdx.x;
    uint32_t total = 0;
    for (int32_t p = 0; p < n_parts; p++) total += (uint32_t)n_hits[(size_t)p * n_q + q];
    const uint32_t keep = total < (uint32_t)k ? total : (uint32_t)k;
    for (uint32_t e = threadIdx.x; e < (uint32_t)n_parts * (uint32_t)k; e += TPB) {
        const int32_t p = (int32_t)(e / (uint32_t)k);
        const uint32_t i = e % (uint32_t)k;
        if (i >= (uint32_t)n_hits[(size_t)p * n_q + q]) continue;
        ss_hit h = parts[((size_t)p * n_q + q) * k + i];
        h.doc += doc_base ? doc_base[p] : 0u;
        const uint64_t key = ss::fkey(h.final);
        uint32_t pos = i;
        for (int32_t o = 0; o < n_parts && pos < keep; o++) {
            if (o == p) continue;
            const ss_hit* __restrict__ lst = parts + ((size_t)o * n_q + q) * k;
            const uint32_t base = doc_base ? doc_base[o] : 0u;
            uint32_t lo = 0, hi = (uint32_t)n_hits[(size_t)o * n_q + q];
            while (lo < hi) {                         // first row of shard o that does not precede h
                const uint32_t mid = (lo + hi) >> 1;
                if (ss::better(ss::fkey(lst[mid].final), lst[mid].doc + base, key, h.doc)) lo = mid + 1; else hi = mid;
            }
            pos += lo;
        }
        if (pos < keep) out[(size_t)q * k + pos] = h;
    }
    for (uint32_t i = keep + threadIdx.x; i < (uint32_t)k; i += TPB) {
        ss_hit z;
        z.doc = 0; z._pad = 0; z.title = 0.0; z.body = 0.0; z.pagerank = 0.0; z.final = 0.0;
        out[(size_t)q * k + i] = z;
    }
    if (threadIdx.x == 0) n_out[q] = (int32_t)keep;
}

}  // namespace

extern "C" int32_t ss_merge_hits(ss_ctx* ctx, int32_t n_q, int32_t n_parts, int32_t k, const ss_hit* parts, const int32_t* n_hits,
                                 const uint32_t* doc_base, ss_hit* hits_out, int32_t* n_hits_out) {
    if (!ctx) return SS_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (n_q < 0 || n_parts < 1 || n_parts > SS_MAX_SHARDS || k < 1 || k > SS_MAX_TOPK)
        return ctx->fail(SS_ERR_INVALID, "ss_merge_hits: n_q=%d n_parts=%d (1..%d) k=%d (1..%d)", n_q, n_parts, SS_MAX_SHARDS, k, SS_MAX_TOPK);
    if (!parts || !n_hits || !hits_out || !n_hits_out) return ctx->fail(SS_ERR_INVALID, "ss_merge_hits: NULL argument");
    if (n_q == 0) return SS_OK;
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t rows = (size_t)n_parts * n_q;
    ss::DevBuf<ss_hit> d_parts, d_out;
    ss::DevBuf<int32_t> d_nh, d_nout;
    ss::DevBuf<uint32_t> d_base;
    SS_HIP(ctx, d_parts.alloc(rows * k));
    SS_HIP(ctx, d_out.alloc((size_t)n_q * k));
    SS_HIP(ctx, d_nh.alloc(rows));
    SS_HIP(ctx, d_nout.alloc(n_q));
    SS_HIP(ctx, hipMemcpyAsync(d_parts.p, parts, rows * k * sizeof(ss_hit), hipMemcpyDefault, st));
    SS_HIP(ctx, hipMemcpyAsync(d_nh.p, n_hits, rows * sizeof(int32_t), hipMemcpyDefault, st));
    if (doc_base) {
        SS_HIP(ctx, d_base.alloc(n_parts));
        SS_HIP(ctx, hipMemcpyAsync(d_base.p, doc_base, n_parts * sizeof(uint32_t), hipMemcpyDefault, st));
    }
    // the counts index the lists: check them before the kernel trusts them
    std::vector<int32_t> h_nh(rows);
    SS_HIP(ctx, hipMemcpyAsync(h_nh.data(), d_nh.p, rows * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    for (size_t i = 0; i < rows; i++)
        if (h_nh[i] < 0 || h_nh[i] > k) return ctx->fail(SS_ERR_INVALID, "ss_merge_hits: n_hits[%zu] = %d outside 0..k", i, h_nh[i]);
    hipLaunchKernelGGL(k_merge_hits, dim3((unsigned)n_q), dim3(TPB), 0, st, n_q, n_parts, k, d_parts.p, d_nh.p,
                       doc_base ? d_base.p : nullptr, d_out.p, d_nout.p);
    SS_HIP(ctx, hipGetLastError());
    SS_HIP(ctx, hipMemcpyAsync(hits_out, d_out.p, (size_t)n_q * k * sizeof(ss_hit), hipMemcpyDefault, st));
    SS_HIP(ctx, hipMemcpyAsync(n_hits_out, d_nout.p, (size_t)n_q * sizeof(int32_t), hipMemcpyDefault, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    return SS_OK;
}
