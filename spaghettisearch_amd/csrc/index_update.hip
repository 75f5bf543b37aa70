// index_update.hip — incremental update of a resident inverted table.
//
// Reference: indexer/indexer.go:420-641 (checkAndUpdate).  When a re-crawled page has changed, the reference removes
// the page from the posting row of every word of its old title (:455-485) and body (:487-531), removes the anchor-text
// postings of its children from inv[0] (:533-616), and then indexes the page again like a new one (Index, :107-408),
// which appends fresh postings.  Every step there is a BadgerDB Get + json.Unmarshal + Set of a WHOLE posting row.
// Here the table stays in HBM as a term-major CSR and a delta is merged into it on the device:
//   del_docs      every posting of these docs goes          (the changed page's old title/body words)
//   del (t, d)    single postings go                         (anchor words of the page's children)
//   add (t, d, w) postings arrive                            (the re-indexed page)
// One pass over the resident arrays: keep flags -> exclusive scan -> every surviving posting and every new posting
// computes its own slot (its rank among the survivors of its list + the number of new postings of that list with a
// smaller doc id, found by binary search in the sorted delta) and is written there.  No host re-flatten, no re-upload;
// the strictly-ascending-doc invariant of every list is re-validated before the new arrays replace the old ones
// (on failure the table is unchanged).
#include "index.hpp"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <memory>

namespace {

constexpr int TPB = 256;

inline unsigned grid_for(uint64_t n, unsigned cap = 1u << 16) { return (unsigned)std::min<uint64_t>(std::max<uint64_t>((n + TPB - 1) / TPB, 1), cap); }

__global__ void k_mark_docs(const uint32_t* __restrict__ del_docs, uint64_t n_del, uint64_t n_docs, uint32_t* __restrict__ bitmap, uint32_t* __restrict__ err) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_del; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t d = del_docs[i];
        if ((uint64_t)d >= n_docs) { atomicOr(err, 1u); continue; }
        atomicOr(&bitmap[d >> 5], 1u << (d & 31));
    }
}
// keep[i] = 1 unless the posting's doc is deleted
__global__ void k_keep_from_bitmap(const uint32_t* __restrict__ post_doc, uint64_t n_post, const uint32_t* __restrict__ bitmap, uint8_t* __restrict__ keep) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_post; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t d = post_doc[i];
        keep[i] = (bitmap[d >> 5] >> (d & 31)) & 1u ? 0 : 1;
    }
}
// single postings to delete: locate (term, doc) by binary search; a pair that does not exist is ignored, like the
// reference's delete(docP, docHash) on a map without the key
__global__ void k_unkeep_pairs(const uint64_t* __restrict__ term_ptr, const uint32_t* __restrict__ post_doc, uint64_t n_terms, uint64_t n_docs,
                               const uint32_t* __restrict__ del_term, const uint32_t* __restrict__ del_doc, uint64_t n_del,
                               uint8_t* __restrict__ keep, uint32_t* __restrict__ err) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_del; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t t = del_term[i], d = del_doc[i];
        if ((uint64_t)t >= n_terms || (uint64_t)d >= n_docs) { atomicOr(err, 2u); continue; }
        uint64_t lo = term_ptr[t], hi = term_ptr[t + 1];
        const uint64_t end = hi;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (post_doc[mid] < d) lo = mid + 1; else hi = mid;
        }
        if (lo < end && post_doc[lo] == d) keep[lo] = 0;
    }
}
__global__ void k_add_keys(const uint32_t* __restrict__ add_term, const uint32_t* __restrict__ add_doc, uint64_t n_add, uint64_t n_terms, uint64_t n_docs,
                           uint64_t* __restrict__ keys, uint32_t* __restrict__ err) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_add; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t t = add_term[i], d = add_doc[i];
        if ((uint64_t)t >= n_terms || (uint64_t)d >= n_docs) atomicOr(err, 4u);
        keys[i] = ((uint64_t)t << 32) | d;
    }
}
// first index in the sorted add keys whose key is >= `key`, inside [lo, hi)
__device__ __forceinline__ uint32_t add_lower(const uint64_t* __restrict__ keys, uint32_t lo, uint32_t hi, uint64_t key) {
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// add_ptr[t] = first new posting of term t in the sorted delta (a CSR over the terms: the merge looks a term's additions
// up with two loads instead of two searches per posting)
__global__ void k_add_ptr(const uint64_t* __restrict__ add_keys, uint32_t n_add, uint64_t n_terms, uint32_t* __restrict__ add_ptr) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_terms) return;
    add_ptr[t] = t == n_terms ? n_add : add_lower(add_keys, 0, n_add, t << 32);
}
// new list lengths: survivors + additions per term
__global__ void k_new_counts(const uint64_t* __restrict__ term_ptr, const uint32_t* __restrict__ kept_before /*[P+1]*/, uint64_t n_terms,
                             const uint32_t* __restrict__ add_ptr, uint64_t* __restrict__ cnt /*[T+1]*/) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_terms) return;
    if (t == n_terms) { cnt[t] = 0; return; }
    cnt[t] = (uint64_t)(kept_before[term_ptr[t + 1]] - kept_before[term_ptr[t]]) + (add_ptr[t + 1] - add_ptr[t]);
}
// every surviving posting writes itself to its slot in the merged list.  One block per CHUNK consecutive postings: the
// block finds the terms its chunk spans with two binary searches, each posting finds its own term inside that short
// range (a chunk of a long list is one term); a term without additions needs nothing more.
constexpr int PK_PT = 8;
constexpr int PK_CHUNK = TPB * PK_PT;
__global__ __launch_bounds__(TPB) void k_place_kept(const uint64_t* __restrict__ term_ptr, uint64_t n_terms, const uint32_t* __restrict__ post_doc,
                                                    const float* __restrict__ post_w, uint64_t n_post, const uint8_t* __restrict__ keep,
                                                    const uint32_t* __restrict__ kept_before, const uint64_t* __restrict__ add_keys,
                                                    const uint32_t* __restrict__ add_ptr, const uint64_t* __restrict__ new_ptr,
                                                    uint32_t* __restrict__ out_doc, float* __restrict__ out_w) {
    __shared__ uint64_t s_t[2];
    const uint64_t base = (uint64_t)blockIdx.x * PK_CHUNK;
    const uint64_t last = min(base + PK_CHUNK, n_post) - 1;
    if (threadIdx.x < 2) {
        const uint64_t target = threadIdx.x == 0 ? base : last;      // largest t with term_ptr[t] <= target
        uint64_t lo = 0, hi = n_terms;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (term_ptr[mid] <= target) lo = mid; else hi = mid;
        }
        s_t[threadIdx.x] = lo;
    }
    __syncthreads();
    const uint64_t t_lo = s_t[0], t_hi = s_t[1];
#pragma unroll 4
    for (int j = 0; j < PK_PT; j++) {
        const uint64_t i = base + (uint64_t)j * TPB + threadIdx.x;
        if (i > last) break;
        if (!keep[i]) continue;
        uint64_t lo = t_lo, hi = t_hi + 1;                            // term_ptr[lo] <= i < term_ptr[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (term_ptr[mid] <= i) lo = mid; else hi = mid;
        }
        const uint64_t t = lo;
        const uint32_t d = post_doc[i];
        const uint32_t rank_kept = kept_before[i] - kept_before[term_ptr[t]];
        const uint32_t a0 = add_ptr[t], a1 = add_ptr[t + 1];
        const uint32_t adds_below = a1 > a0 ? add_lower(add_keys, a0, a1, (t << 32) | d) - a0 : 0u;
        const uint64_t o = new_ptr[t] + rank_kept + adds_below;
        out_doc[o] = d;
        out_w[o] = post_w[i];
    }
}
// every new posting: its rank among the additions of its term + the survivors of the term with a smaller doc id
__global__ void k_place_adds(const uint64_t* __restrict__ term_ptr, const uint32_t* __restrict__ post_doc, const uint8_t* __restrict__ keep,
                             const uint32_t* __restrict__ kept_before, const uint64_t* __restrict__ add_keys, const uint32_t* __restrict__ add_order,
                             const float* __restrict__ add_w, uint64_t n_add, const uint32_t* __restrict__ add_ptr, const uint64_t* __restrict__ new_ptr,
                             uint32_t* __restrict__ out_doc, float* __restrict__ out_w, uint32_t* __restrict__ err) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_add) return;
    const uint64_t key = add_keys[j];
    if (j > 0 && add_keys[j - 1] == key) { atomicOr(err, 8u); return; }             // the same (term, doc) twice in the delta
    const uint64_t t = key >> 32;
    const uint32_t d = (uint32_t)key;
    uint64_t lo = term_ptr[t], hi = term_ptr[t + 1];
    const uint64_t end = hi;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (post_doc[mid] < d) lo = mid + 1; else hi = mid;
    }
    if (lo < end && post_doc[lo] == d && keep[lo]) { atomicOr(err, 16u); return; }   // the posting already exists and was not deleted
    const uint32_t kept_below = kept_before[lo] - kept_before[term_ptr[t]];
    const uint64_t o = new_ptr[t] + ((uint32_t)j - add_ptr[t]) + kept_below;
    out_doc[o] = d;
    out_w[o] = add_w[add_order[j]];
}
__global__ void k_iota(uint32_t* __restrict__ v, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) v[i] = (uint32_t)i;
}
// strictly ascending inside every list of the merged table: every posting looks at its predecessor (a term boundary is
// found like in k_place_kept)
__global__ __launch_bounds__(TPB) void k_check_merged(const uint64_t* __restrict__ new_ptr, uint64_t n_terms, const uint32_t* __restrict__ doc, uint64_t n_post,
                                                      uint32_t* __restrict__ err) {
    __shared__ uint64_t s_t[2];
    const uint64_t base = (uint64_t)blockIdx.x * PK_CHUNK;
    if (base >= n_post) return;
    const uint64_t last = min(base + PK_CHUNK, n_post) - 1;
    if (threadIdx.x < 2) {
        const uint64_t target = threadIdx.x == 0 ? base : last;
        uint64_t lo = 0, hi = n_terms;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (new_ptr[mid] <= target) lo = mid; else hi = mid;
        }
        s_t[threadIdx.x] = lo;
    }
    __syncthreads();
    const uint64_t t_lo = s_t[0], t_hi = s_t[1];
    bool bad = false;
    for (int j = 0; j < PK_PT; j++) {
        const uint64_t i = base + (uint64_t)j * TPB + threadIdx.x;
        if (i > last || i == 0) continue;
        uint64_t lo = t_lo, hi = t_hi + 1;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (new_ptr[mid] <= i) lo = mid; else hi = mid;
        }
        if (i > new_ptr[lo] && doc[i] <= doc[i - 1]) bad = true;      // not the first posting of its list
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(err, 32u);
}

template <typename In, typename Out>
int32_t exclusive_scan_u64(ss_ctx* ctx, In in, Out out, size_t n) {
    size_t tmp_bytes = 0;
    SS_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), ctx->stream));
    ss::DevBuf<char> tmp;
    SS_HIP(ctx, tmp.alloc(tmp_bytes));
    SS_HIP(ctx, rocprim::exclusive_scan(tmp.p, tmp_bytes, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), ctx->stream));
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SS_OK;
}
template <typename In, typename Out>
int32_t exclusive_scan_u32(ss_ctx* ctx, In in, Out out, size_t n) {
    size_t tmp_bytes = 0;
    SS_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, (uint32_t)0, n, rocprim::plus<uint32_t>(), ctx->stream));
    ss::DevBuf<char> tmp;
    SS_HIP(ctx, tmp.alloc(tmp_bytes));
    SS_HIP(ctx, rocprim::exclusive_scan(tmp.p, tmp_bytes, in, out, (uint32_t)0, n, rocprim::plus<uint32_t>(), ctx->stream));
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SS_OK;
}


}  // namespace

extern "C" {

int32_t ss_index_apply_delta(ss_index* idx, uint64_t n_del_docs, const uint32_t* del_docs, uint64_t n_del, const uint32_t* del_term,
                             const uint32_t* del_doc, uint64_t n_add, const uint32_t* add_term, const uint32_t* add_doc, const float* add_w) {
    if (!idx) return SS_ERR_INVALID;
    ss_ctx* ctx = idx->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (idx->users > 0) return ctx->fail(SS_ERR_STATE, "ss_index_apply_delta: %d scorer(s) still hold this table (destroy them, update, create them again)", idx->users);
    if ((n_del_docs && !del_docs) || (n_del && (!del_term || !del_doc)) || (n_add && (!add_term || !add_doc || !add_w)))
        return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: NULL array with a non-zero count");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t P = idx->n_post, T = idx->n_terms, N = idx->n_docs;
    if (P + n_add >= ((uint64_t)1 << 32)) return ctx->fail(SS_ERR_UNSUPPORTED, "ss_index_apply_delta: more than 2^32 postings");

    ss::DevBuf<uint32_t> bitmap, err, d_del_docs, d_del_term, d_del_doc, d_add_term, d_add_doc, order_in, order, kept_before, add_ptr;
    ss::DevBuf<uint8_t> keep;
    ss::DevBuf<float> d_add_w;
    ss::DevBuf<uint64_t> keys_in, keys, cnt, new_ptr;
    SS_HIP(ctx, bitmap.alloc((N + 31) / 32));
    SS_HIP(ctx, keep.alloc(P + 1));
    SS_HIP(ctx, kept_before.alloc(P + 1));
    SS_HIP(ctx, err.alloc(1));
    SS_HIP(ctx, hipMemsetAsync(bitmap.p, 0, std::max<size_t>(bitmap.bytes(), 4), st));
    SS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(uint32_t), st));
    SS_HIP(ctx, hipMemsetAsync(keep.p + P, 0, 1, st));
    if (n_del_docs) {
        SS_HIP(ctx, d_del_docs.alloc(n_del_docs));
        SS_HIP(ctx, hipMemcpyAsync(d_del_docs.p, del_docs, n_del_docs * sizeof(uint32_t), hipMemcpyDefault, st));
        hipLaunchKernelGGL(k_mark_docs, dim3(grid_for(n_del_docs)), dim3(TPB), 0, st, (const uint32_t*)d_del_docs.p, n_del_docs, N, bitmap.p, err.p);
    }
    if (P) hipLaunchKernelGGL(k_keep_from_bitmap, dim3(grid_for(P)), dim3(TPB), 0, st, (const uint32_t*)idx->post_doc.p, P, (const uint32_t*)bitmap.p, keep.p);
    if (n_del) {
        SS_HIP(ctx, d_del_term.alloc(n_del));
        SS_HIP(ctx, d_del_doc.alloc(n_del));
        SS_HIP(ctx, hipMemcpyAsync(d_del_term.p, del_term, n_del * sizeof(uint32_t), hipMemcpyDefault, st));
        SS_HIP(ctx, hipMemcpyAsync(d_del_doc.p, del_doc, n_del * sizeof(uint32_t), hipMemcpyDefault, st));
        hipLaunchKernelGGL(k_unkeep_pairs, dim3(grid_for(n_del)), dim3(TPB), 0, st, (const uint64_t*)idx->term_ptr.p, (const uint32_t*)idx->post_doc.p, T, N,
                           (const uint32_t*)d_del_term.p, (const uint32_t*)d_del_doc.p, n_del, keep.p, err.p);
    }
    // additions sorted by (term, doc); the order array carries the weights along
    SS_HIP(ctx, keys_in.alloc(n_add));
    SS_HIP(ctx, keys.alloc(n_add));
    SS_HIP(ctx, order_in.alloc(n_add));
    SS_HIP(ctx, order.alloc(n_add));
    SS_HIP(ctx, d_add_w.alloc(n_add));
    if (n_add) {
        SS_HIP(ctx, d_add_term.alloc(n_add));
        SS_HIP(ctx, d_add_doc.alloc(n_add));
        SS_HIP(ctx, hipMemcpyAsync(d_add_term.p, add_term, n_add * sizeof(uint32_t), hipMemcpyDefault, st));
        SS_HIP(ctx, hipMemcpyAsync(d_add_doc.p, add_doc, n_add * sizeof(uint32_t), hipMemcpyDefault, st));
        SS_HIP(ctx, hipMemcpyAsync(d_add_w.p, add_w, n_add * sizeof(float), hipMemcpyDefault, st));
        hipLaunchKernelGGL(k_add_keys, dim3(grid_for(n_add)), dim3(TPB), 0, st, (const uint32_t*)d_add_term.p, (const uint32_t*)d_add_doc.p, n_add, T, N, keys_in.p, err.p);
        hipLaunchKernelGGL(k_iota, dim3(grid_for(n_add)), dim3(TPB), 0, st, order_in.p, n_add);
        size_t tmp_bytes = 0;
        SS_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in.p, keys.p, order_in.p, order.p, (size_t)n_add, 0u, 64u, st));
        ss::DevBuf<char> tmp;
        SS_HIP(ctx, tmp.alloc(tmp_bytes));
        SS_HIP(ctx, rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys_in.p, keys.p, order_in.p, order.p, (size_t)n_add, 0u, 64u, st));
        SS_HIP(ctx, hipStreamSynchronize(st));
    }
    // Range errors stop the update HERE: the placement kernels below index term_ptr / add_ptr / new_ptr by the delta's
    // term ids, so an out-of-range id must never reach them.
    {
        uint32_t h_early = 0;
        SS_HIP(ctx, hipMemcpyAsync(&h_early, err.p, sizeof(h_early), hipMemcpyDeviceToHost, st));
        SS_HIP(ctx, hipStreamSynchronize(st));
        if (h_early & 1) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: del_docs holds a doc id >= n_docs (table unchanged)");
        if (h_early & 2) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: a (term, doc) to delete is out of range (table unchanged)");
        if (h_early & 4) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: a posting to add is out of range (table unchanged)");
    }
    // survivors before every posting, the delta as a CSR over the terms, new list lengths, new term_ptr
    {
        auto in_it = rocprim::make_transform_iterator(keep.p, [] __device__(uint8_t x) { return (uint32_t)x; });
        SS_TRY(exclusive_scan_u32(ctx, in_it, kept_before.p, (size_t)(P + 1)));
    }
    SS_HIP(ctx, add_ptr.alloc(T + 1));
    hipLaunchKernelGGL(k_add_ptr, dim3(grid_for(T + 1)), dim3(TPB), 0, st, (const uint64_t*)keys.p, (uint32_t)n_add, T, add_ptr.p);
    SS_HIP(ctx, cnt.alloc(T + 1));
    SS_HIP(ctx, new_ptr.alloc(T + 1));
    hipLaunchKernelGGL(k_new_counts, dim3(grid_for(T + 1)), dim3(TPB), 0, st, (const uint64_t*)idx->term_ptr.p, (const uint32_t*)kept_before.p, T,
                       (const uint32_t*)add_ptr.p, cnt.p);
    SS_TRY(exclusive_scan_u64(ctx, cnt.p, new_ptr.p, (size_t)(T + 1)));
    std::vector<uint64_t> h_new_ptr(T + 1);
    SS_HIP(ctx, hipMemcpyAsync(h_new_ptr.data(), new_ptr.p, (T + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    const uint64_t P2 = h_new_ptr[T];
    ss::DevBuf<uint32_t> out_doc;
    ss::DevBuf<float> out_w;
    SS_HIP(ctx, out_doc.alloc(P2));
    SS_HIP(ctx, out_w.alloc(P2));
    if (P) hipLaunchKernelGGL(k_place_kept, dim3(ss::div_up(P, PK_CHUNK)), dim3(TPB), 0, st, (const uint64_t*)idx->term_ptr.p, T, (const uint32_t*)idx->post_doc.p,
                              (const float*)idx->post_w.p, P, (const uint8_t*)keep.p, (const uint32_t*)kept_before.p, (const uint64_t*)keys.p,
                              (const uint32_t*)add_ptr.p, (const uint64_t*)new_ptr.p, out_doc.p, out_w.p);
    if (n_add) hipLaunchKernelGGL(k_place_adds, dim3(grid_for(n_add)), dim3(TPB), 0, st, (const uint64_t*)idx->term_ptr.p, (const uint32_t*)idx->post_doc.p,
                                  (const uint8_t*)keep.p, (const uint32_t*)kept_before.p, (const uint64_t*)keys.p, (const uint32_t*)order.p,
                                  (const float*)d_add_w.p, n_add, (const uint32_t*)add_ptr.p, (const uint64_t*)new_ptr.p, out_doc.p, out_w.p, err.p);
    if (P2) hipLaunchKernelGGL(k_check_merged, dim3(ss::div_up(P2, PK_CHUNK)), dim3(TPB), 0, st, (const uint64_t*)new_ptr.p, T, (const uint32_t*)out_doc.p, P2, err.p);
    SS_HIP(ctx, hipGetLastError());
    uint32_t h_err = 0;
    SS_HIP(ctx, hipMemcpyAsync(&h_err, err.p, sizeof(h_err), hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    if (h_err & 1) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: del_docs holds a doc id >= n_docs (table unchanged)");
    if (h_err & 2) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: a (term, doc) to delete is out of range (table unchanged)");
    if (h_err & 4) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: a posting to add is out of range (table unchanged)");
    if (h_err & 8) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: the same (term, doc) is added twice (table unchanged)");
    if (h_err & 16) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: a posting to add already exists and is not deleted by this delta (table unchanged)");
    if (h_err & 32) return ctx->fail(SS_ERR_UNSORTED, "ss_index_apply_delta: merged list not strictly ascending (table unchanged)");
    // commit
    idx->post_doc = std::move(out_doc);
    idx->post_w = std::move(out_w);
    idx->term_ptr = std::move(new_ptr);
    idx->h_term_ptr = std::move(h_new_ptr);
    idx->n_post = P2;
    idx->pos_ptr.release();            // positional postings no longer line up: set them again for phrase search
    idx->pos.release();
    return SS_OK;
}

}  // extern "C"
