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
// touched docs = every doc whose magnitude the delta can change (deleted docs, docs of deleted pairs, docs of new postings);
// ids out of range are reported by the kernels that own the array's error bit
__global__ void k_mark_touched(const uint32_t* __restrict__ docs, uint64_t n, uint64_t n_docs, uint32_t* __restrict__ touched) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t d = docs[i];
        if ((uint64_t)d < n_docs) atomicOr(&touched[d >> 5], 1u << (d & 31));
    }
}
// keep[i] = 1 unless the posting's doc is deleted; *n_touched += surviving postings of touched docs (an upper bound of what
// k_place_kept will hand to the magnitude pass: pair deletes only lower it)
__global__ void k_keep_from_bitmap(const uint32_t* __restrict__ post_doc, uint64_t n_post, const uint32_t* __restrict__ bitmap, uint8_t* __restrict__ keep,
                                   const uint32_t* __restrict__ touched /*nullable*/, unsigned long long* __restrict__ n_touched) {
    uint32_t mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_post; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t d = post_doc[i];
        const uint32_t k = (bitmap[d >> 5] >> (d & 31)) & 1u ? 0 : 1;
        keep[i] = (uint8_t)k;
        if (touched) mine += k & (touched[d >> 5] >> (d & 31));
    }
    if (touched) {
        for (int off = 32; off; off >>= 1) mine += __shfl_down(mine, off);
        if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_touched, (unsigned long long)mine);
    }
}
// single postings to delete: locate (term, doc) by binary search; a pair that does not exist is ignored, like the
// reference's delete(docP, docHash) on a map without the key
// (the keep flag is cleared with an atomic on its word: neighbouring flags belong to other threads' pairs)
__global__ void k_unkeep_pairs(const uint64_t* __restrict__ term_ptr, const uint32_t* __restrict__ post_doc,
                               uint64_t n_terms, uint64_t n_docs,
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
        if (lo < end && post_doc[lo] == d) {
            uint32_t* word = reinterpret_cast<uint32_t*>(keep + (lo & ~(uint64_t)3));
            const uint32_t sh = (uint32_t)(lo & 3) * 8u;
            atomicAnd(word, ~(0xFFu << sh));
        }
    }
}
// Magnitudes of the touched docs (term_weighting.go:44,72).  They are RECOMPUTED from the docs' postings in the merged table,
// never patched: a float64 sum of float32 squares is exact only while the squares span fewer than 29 binary orders, and
// tf-idf weights do not promise that (idf = log2(N/df) with N the PageRank node count, term_weighting.go:13-17,37: it can be
// 20 for one word and 1e-5, or negative, for another), so "subtract what left" can cancel to garbage or to a negative
// number.  k_place_kept / k_place_adds hand every posting of a touched doc to a list {doc << 32 | term, float32(w*w)}; the
// list is sorted and every doc's squares are summed by one thread in ascending term order — the order in which the
// reference's pass over the inverted table reaches a doc (term_weighting.go:29-46) — so the result is the full pass's, bit
// for bit, whatever the weights.
__global__ void k_mag_zero_docs(const uint32_t* __restrict__ docs, uint64_t n, double* __restrict__ mag2, double* __restrict__ mag) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { mag2[docs[i]] = 0.0; mag[docs[i]] = 0.0; }
}
__global__ void k_mag_segments(const uint64_t* __restrict__ keys, const float* __restrict__ sq, uint64_t n, double* __restrict__ mag2, double* __restrict__ mag) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t d = (uint32_t)(keys[i] >> 32);
        if (i > 0 && (uint32_t)(keys[i - 1] >> 32) == d) continue;          // not the first posting of its doc
        double sum = 0.0;
        for (uint64_t j = i; j < n && (uint32_t)(keys[j] >> 32) == d; j++) sum += (double)sq[j];   // :44 float32 product, float64 sum
        mag2[d] = sum;
        mag[d] = sqrt(sum);                                                  // :72
    }
}
__global__ void k_gather_f64(const double* __restrict__ v, const uint32_t* __restrict__ idx, uint64_t n, uint64_t limit, double* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = (uint64_t)idx[i] < limit ? v[idx[i]] : 0.0;
}
// positional postings of the merged table: every output posting copies its positions from the old array or from the delta's
__global__ void k_copy_positions(const uint64_t* __restrict__ new_pos_ptr, const uint64_t* __restrict__ src_start /* bit 63: from the delta */,
                                 uint64_t n_post, const float* __restrict__ old_pos, const float* __restrict__ add_pos, float* __restrict__ out) {
    for (uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_post; o += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = new_pos_ptr[o], n = new_pos_ptr[o + 1] - b, s = src_start[o];
        const float* from = (s >> 63) ? add_pos + (s & ~(1ull << 63)) : old_pos + s;
        for (uint64_t q = 0; q < n; q++) out[b + q] = from[q];
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
// where the placement kernels leave the postings of touched docs for the magnitude pass (k_mag_segments)
struct TouchedList {
    const uint32_t* bitmap;        // NULL: magnitudes are not maintained by this delta
    unsigned long long* cursor;
    uint64_t cap;
    uint64_t* keys;                // doc << 32 | term
    float* sq;                     // float32(w * w), term_weighting.go:44
    uint32_t* err;
    __device__ __forceinline__ void take(uint32_t t, uint32_t d, float w) const {
        if (!bitmap || !((bitmap[d >> 5] >> (d & 31)) & 1u)) return;
        const unsigned long long slot = atomicAdd(cursor, 1ull);
        if (slot >= cap) { atomicOr(err, 64u); return; }
        keys[slot] = ((uint64_t)d << 32) | t;
        sq[slot] = w * w;
    }
};
constexpr int PK_PT = 8;
constexpr int PK_CHUNK = TPB * PK_PT;
__global__ __launch_bounds__(TPB) void k_place_kept(const uint64_t* __restrict__ term_ptr, uint64_t n_terms, const uint32_t* __restrict__ post_doc,
                                                    const float* __restrict__ post_w, uint64_t n_post, const uint8_t* __restrict__ keep,
                                                    const uint32_t* __restrict__ kept_before, const uint64_t* __restrict__ add_keys,
                                                    const uint32_t* __restrict__ add_ptr, const uint64_t* __restrict__ new_ptr,
                                                    uint32_t* __restrict__ out_doc, float* __restrict__ out_w,
                                                    const uint64_t* __restrict__ pos_ptr /*nullable*/, uint64_t* __restrict__ len_out, uint64_t* __restrict__ src_start,
                                                    TouchedList tl) {
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
        const float w = post_w[i];
        out_doc[o] = d;
        out_w[o] = w;
        if (pos_ptr) { len_out[o] = pos_ptr[i + 1] - pos_ptr[i]; src_start[o] = pos_ptr[i]; }
        tl.take((uint32_t)t, d, w);
    }
}
// every new posting: its rank among the additions of its term + the survivors of the term with a smaller doc id
__global__ void k_place_adds(const uint64_t* __restrict__ term_ptr, const uint32_t* __restrict__ post_doc, const uint8_t* __restrict__ keep,
                             const uint32_t* __restrict__ kept_before, const uint64_t* __restrict__ add_keys, const uint32_t* __restrict__ add_order,
                             const float* __restrict__ add_w, uint64_t n_add, const uint32_t* __restrict__ add_ptr, const uint64_t* __restrict__ new_ptr,
                             uint32_t* __restrict__ out_doc, float* __restrict__ out_w, uint32_t* __restrict__ err,
                             const uint64_t* __restrict__ add_pos_ptr /*nullable*/, uint64_t* __restrict__ len_out /*nullable*/, uint64_t* __restrict__ src_start,
                             TouchedList tl) {
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
    const float w = add_w[add_order[j]];
    out_doc[o] = d;
    out_w[o] = w;
    tl.take((uint32_t)t, d, w);
    if (len_out) {
        const uint32_t a = add_order[j];
        len_out[o] = add_pos_ptr ? add_pos_ptr[a + 1] - add_pos_ptr[a] : 0ull;
        src_start[o] = (1ull << 63) | (add_pos_ptr ? add_pos_ptr[a] : 0ull);
    }
}
__global__ void k_check_mono_u64(const uint64_t* __restrict__ p, uint64_t n, uint32_t* __restrict__ err) {
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) bad = bad || p[i + 1] < p[i];
    if (bad) atomicOr(err, 1u);
}
__global__ void k_fill_u64(uint64_t* __restrict__ v, uint64_t lo, uint64_t hi, uint64_t x) {
    for (uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (uint64_t)gridDim.x * blockDim.x) v[i] = x;
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
    return ss_index_apply_delta_pos(idx, n_del_docs, del_docs, n_del, del_term, del_doc, n_add, add_term, add_doc, add_w, nullptr, nullptr);
}

int32_t ss_index_apply_delta_pos(ss_index* idx, uint64_t n_del_docs, const uint32_t* del_docs, uint64_t n_del, const uint32_t* del_term,
                                 const uint32_t* del_doc, uint64_t n_add, const uint32_t* add_term, const uint32_t* add_doc, const float* add_w,
                                 const uint64_t* add_pos_ptr, const float* add_pos) {
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
    ss::DevBuf<float> d_add_w, d_add_pos;
    ss::DevBuf<uint64_t> keys_in, keys, cnt, new_ptr, d_add_pos_ptr, len_out, src_start, new_pos_ptr;
    const bool has_pos = idx->pos_ptr.p != nullptr;
    uint64_t n_add_pos = 0;
    if (has_pos && add_pos_ptr && n_add) {
        uint64_t ends[2] = {0, 0};
        SS_HIP(ctx, ss::fetch(ctx, st, &ends[0], add_pos_ptr, sizeof(uint64_t), &ends[1], add_pos_ptr + n_add, sizeof(uint64_t)));
        if (ends[0] != 0 || (ends[1] && !add_pos)) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: add_pos_ptr[0] != 0 or add_pos NULL");
        n_add_pos = ends[1];
        SS_HIP(ctx, d_add_pos_ptr.alloc(n_add + 1));
        SS_HIP(ctx, d_add_pos.alloc(n_add_pos));
        SS_HIP(ctx, hipMemcpyAsync(d_add_pos_ptr.p, add_pos_ptr, (n_add + 1) * sizeof(uint64_t), hipMemcpyDefault, st));
        if (n_add_pos) SS_HIP(ctx, hipMemcpyAsync(d_add_pos.p, add_pos, n_add_pos * sizeof(float), hipMemcpyDefault, st));
        ss::DevBuf<uint32_t> perr;
        SS_HIP(ctx, perr.alloc(1));
        SS_HIP(ctx, hipMemsetAsync(perr.p, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_check_mono_u64, dim3(grid_for(n_add)), dim3(TPB), 0, st, (const uint64_t*)d_add_pos_ptr.p, n_add, perr.p);
        uint32_t h_perr = 0;
        SS_HIP(ctx, ss::fetch(ctx, st, &h_perr, perr.p, sizeof(uint32_t)));
        if (h_perr) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: add_pos_ptr is not non-decreasing");
    }
    const bool keep_mag = idx->mag2_valid;
    ss::DevBuf<uint32_t> touched;
    ss::DevBuf<unsigned long long> tcount;                        // [0] upper bound from k_keep_from_bitmap, [1] cursor of the placement kernels
    SS_HIP(ctx, bitmap.alloc((N + 31) / 32));
    SS_HIP(ctx, keep.alloc(((P + 1) + 3) & ~(uint64_t)3));
    SS_HIP(ctx, kept_before.alloc(P + 1));
    SS_HIP(ctx, err.alloc(1));
    SS_HIP(ctx, tcount.alloc(2));
    SS_HIP(ctx, hipMemsetAsync(bitmap.p, 0, std::max<size_t>(bitmap.bytes(), 4), st));
    SS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(uint32_t), st));
    SS_HIP(ctx, hipMemsetAsync(tcount.p, 0, 2 * sizeof(unsigned long long), st));
    SS_HIP(ctx, hipMemsetAsync(keep.p + P, 0, 1, st));
    if (keep_mag) {
        SS_HIP(ctx, touched.alloc((N + 31) / 32));
        SS_HIP(ctx, hipMemsetAsync(touched.p, 0, std::max<size_t>(touched.bytes(), 4), st));
    }
    // the delta's arrays go up first: the touched-doc bitmap needs all three doc arrays before the pass over the postings
    if (n_del_docs) {
        SS_HIP(ctx, d_del_docs.alloc(n_del_docs));
        SS_HIP(ctx, hipMemcpyAsync(d_del_docs.p, del_docs, n_del_docs * sizeof(uint32_t), hipMemcpyDefault, st));
        hipLaunchKernelGGL(k_mark_docs, dim3(grid_for(n_del_docs)), dim3(TPB), 0, st, (const uint32_t*)d_del_docs.p, n_del_docs, N, bitmap.p, err.p);
        if (keep_mag) hipLaunchKernelGGL(k_mark_touched, dim3(grid_for(n_del_docs)), dim3(TPB), 0, st, (const uint32_t*)d_del_docs.p, n_del_docs, N, touched.p);
    }
    if (n_del) {
        SS_HIP(ctx, d_del_term.alloc(n_del));
        SS_HIP(ctx, d_del_doc.alloc(n_del));
        SS_HIP(ctx, hipMemcpyAsync(d_del_term.p, del_term, n_del * sizeof(uint32_t), hipMemcpyDefault, st));
        SS_HIP(ctx, hipMemcpyAsync(d_del_doc.p, del_doc, n_del * sizeof(uint32_t), hipMemcpyDefault, st));
        if (keep_mag) hipLaunchKernelGGL(k_mark_touched, dim3(grid_for(n_del)), dim3(TPB), 0, st, (const uint32_t*)d_del_doc.p, n_del, N, touched.p);
    }
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
        if (keep_mag) hipLaunchKernelGGL(k_mark_touched, dim3(grid_for(n_add)), dim3(TPB), 0, st, (const uint32_t*)d_add_doc.p, n_add, N, touched.p);
    }
    if (P) hipLaunchKernelGGL(k_keep_from_bitmap, dim3(grid_for(P)), dim3(TPB), 0, st, (const uint32_t*)idx->post_doc.p, P, (const uint32_t*)bitmap.p, keep.p,
                              keep_mag ? (const uint32_t*)touched.p : nullptr, tcount.p);
    if (n_del)
        hipLaunchKernelGGL(k_unkeep_pairs, dim3(grid_for(n_del)), dim3(TPB), 0, st, (const uint64_t*)idx->term_ptr.p, (const uint32_t*)idx->post_doc.p,
                           T, N, (const uint32_t*)d_del_term.p, (const uint32_t*)d_del_doc.p, n_del, keep.p, err.p);
    // additions sorted by (term, doc); the order array carries the weights along
    if (n_add) {
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
        SS_HIP(ctx, ss::fetch(ctx, st, &h_early, err.p, sizeof(h_early)));
        if (h_early & 1) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: del_docs holds a doc id >= n_docs (table unchanged)");
        if (h_early & 2) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: a (term, doc) to delete is out of range (table unchanged)");
        if (h_early & 4) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: a posting to add is out of range (table unchanged)");
    }
    // room for the postings of the touched docs (their squares are summed again after the merge)
    TouchedList tl{nullptr, nullptr, 0, nullptr, nullptr, nullptr};
    ss::DevBuf<uint64_t> t_keys_in, t_keys;
    ss::DevBuf<float> t_sq_in, t_sq;
    if (keep_mag) {
        unsigned long long h_bound = 0;
        SS_HIP(ctx, ss::fetch(ctx, st, &h_bound, tcount.p, sizeof(h_bound)));
        const uint64_t cap = (uint64_t)h_bound + n_add;
        SS_HIP(ctx, t_keys_in.alloc(cap));
        SS_HIP(ctx, t_keys.alloc(cap));
        SS_HIP(ctx, t_sq_in.alloc(cap));
        SS_HIP(ctx, t_sq.alloc(cap));
        tl = TouchedList{touched.p, tcount.p + 1, cap, t_keys_in.p, t_sq_in.p, err.p};
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
    if (has_pos) {
        SS_HIP(ctx, len_out.alloc(P2 + 1));
        SS_HIP(ctx, src_start.alloc(P2 + 1));
        SS_HIP(ctx, new_pos_ptr.alloc(P2 + 1));
        SS_HIP(ctx, hipMemsetAsync(len_out.p, 0, (P2 + 1) * sizeof(uint64_t), st));
    }
    if (P) hipLaunchKernelGGL(k_place_kept, dim3(ss::div_up(P, PK_CHUNK)), dim3(TPB), 0, st, (const uint64_t*)idx->term_ptr.p, T, (const uint32_t*)idx->post_doc.p,
                              (const float*)idx->post_w.p, P, (const uint8_t*)keep.p, (const uint32_t*)kept_before.p, (const uint64_t*)keys.p,
                              (const uint32_t*)add_ptr.p, (const uint64_t*)new_ptr.p, out_doc.p, out_w.p,
                              has_pos ? (const uint64_t*)idx->pos_ptr.p : nullptr, len_out.p, src_start.p, tl);
    if (n_add) hipLaunchKernelGGL(k_place_adds, dim3(grid_for(n_add)), dim3(TPB), 0, st, (const uint64_t*)idx->term_ptr.p, (const uint32_t*)idx->post_doc.p,
                                  (const uint8_t*)keep.p, (const uint32_t*)kept_before.p, (const uint64_t*)keys.p, (const uint32_t*)order.p,
                                  (const float*)d_add_w.p, n_add, (const uint32_t*)add_ptr.p, (const uint64_t*)new_ptr.p, out_doc.p, out_w.p, err.p,
                                  (const uint64_t*)d_add_pos_ptr.p, has_pos ? len_out.p : nullptr, src_start.p, tl);
    if (P2) hipLaunchKernelGGL(k_check_merged, dim3(ss::div_up(P2, PK_CHUNK)), dim3(TPB), 0, st, (const uint64_t*)new_ptr.p, T, (const uint32_t*)out_doc.p, P2, err.p);
    SS_HIP(ctx, hipGetLastError());
    uint32_t h_err = 0;
    SS_HIP(ctx, ss::fetch(ctx, st, &h_err, err.p, sizeof(h_err)));
    if (h_err & 1) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: del_docs holds a doc id >= n_docs (table unchanged)");
    if (h_err & 2) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: a (term, doc) to delete is out of range (table unchanged)");
    if (h_err & 4) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: a posting to add is out of range (table unchanged)");
    if (h_err & 8) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: the same (term, doc) is added twice (table unchanged)");
    if (h_err & 16) return ctx->fail(SS_ERR_INVALID, "ss_index_apply_delta: a posting to add already exists and is not deleted by this delta (table unchanged)");
    if (h_err & 32) return ctx->fail(SS_ERR_UNSORTED, "ss_index_apply_delta: merged list not strictly ascending (table unchanged)");
    if (h_err & 64) return ctx->fail(SS_ERR_STATE, "ss_index_apply_delta: internal: more postings of touched docs than counted (table unchanged)");
    // positional postings follow their postings (listPos[1:] of every kept row entry; the re-indexed page brings its own)
    ss::DevBuf<float> new_pos;
    if (has_pos) {
        SS_TRY(exclusive_scan_u64(ctx, len_out.p, new_pos_ptr.p, (size_t)(P2 + 1)));
        uint64_t total = 0;
        SS_HIP(ctx, ss::fetch(ctx, st, &total, new_pos_ptr.p + P2, sizeof(uint64_t)));
        SS_HIP(ctx, new_pos.alloc(total));
        if (P2) hipLaunchKernelGGL(k_copy_positions, dim3(grid_for(P2)), dim3(TPB), 0, st, (const uint64_t*)new_pos_ptr.p, (const uint64_t*)src_start.p, P2,
                                   (const float*)idx->pos.p, (const float*)d_add_pos.p, new_pos.p);
        SS_HIP(ctx, hipGetLastError());
    }
    // magnitudes of the touched docs: zero (a doc may be left without postings), then every doc's squares summed in term order
    if (keep_mag) {
        unsigned long long h_n = 0;
        SS_HIP(ctx, ss::fetch(ctx, st, &h_n, tcount.p + 1, sizeof(h_n)));
        if (h_n) {
            size_t tmp_bytes = 0;
            SS_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, t_keys_in.p, t_keys.p, t_sq_in.p, t_sq.p, (size_t)h_n, 0u, 64u, st));
            ss::DevBuf<char> tmp;
            SS_HIP(ctx, tmp.alloc(tmp_bytes));
            SS_HIP(ctx, rocprim::radix_sort_pairs(tmp.p, tmp_bytes, t_keys_in.p, t_keys.p, t_sq_in.p, t_sq.p, (size_t)h_n, 0u, 64u, st));
            SS_HIP(ctx, hipStreamSynchronize(st));                 // tmp leaves scope
        }
        if (n_del_docs) hipLaunchKernelGGL(k_mag_zero_docs, dim3(grid_for(n_del_docs)), dim3(TPB), 0, st, (const uint32_t*)d_del_docs.p, n_del_docs, idx->mag2.p, idx->mag.p);
        if (n_del) hipLaunchKernelGGL(k_mag_zero_docs, dim3(grid_for(n_del)), dim3(TPB), 0, st, (const uint32_t*)d_del_doc.p, n_del, idx->mag2.p, idx->mag.p);
        if (n_add) hipLaunchKernelGGL(k_mag_zero_docs, dim3(grid_for(n_add)), dim3(TPB), 0, st, (const uint32_t*)d_add_doc.p, n_add, idx->mag2.p, idx->mag.p);
        if (h_n) hipLaunchKernelGGL(k_mag_segments, dim3(grid_for(h_n)), dim3(TPB), 0, st, (const uint64_t*)t_keys.p, (const float*)t_sq.p, (uint64_t)h_n, idx->mag2.p, idx->mag.p);
        SS_HIP(ctx, hipGetLastError());
    }
    SS_HIP(ctx, hipStreamSynchronize(st));
    // commit
    idx->post_doc = std::move(out_doc);
    idx->post_w = std::move(out_w);
    idx->term_ptr = std::move(new_ptr);
    idx->h_term_ptr = std::move(h_new_ptr);
    idx->n_post = P2;
    if (has_pos) {
        idx->pos_ptr = std::move(new_pos_ptr);
        idx->pos = std::move(new_pos);
    }
    return SS_OK;
}

int32_t ss_index_resize(ss_index* idx, uint64_t n_docs_new, uint64_t n_terms_new) {
    if (!idx) return SS_ERR_INVALID;
    ss_ctx* ctx = idx->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (idx->users > 0) return ctx->fail(SS_ERR_STATE, "ss_index_resize: %d scorer(s) still hold this table", idx->users);
    if (n_docs_new < idx->n_docs || n_terms_new < idx->n_terms) return ctx->fail(SS_ERR_INVALID, "ss_index_resize: a table only grows");
    if (n_docs_new >= 0xFFFFFFF0ull || n_terms_new >= 0xFFFFFFF0ull) return ctx->fail(SS_ERR_INVALID, "ss_index_resize: n_docs/n_terms out of range");
    if (idx->has_df_global && n_terms_new != idx->n_terms) return ctx->fail(SS_ERR_UNSUPPORTED, "ss_index_resize: a doc-range shard with whole-corpus document frequencies cannot grow its term space");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t T = idx->n_terms, N = idx->n_docs, P = idx->n_post;
    if (n_terms_new > T) {                                   // new terms: empty lists behind the last posting
        ss::DevBuf<uint64_t> tp;
        SS_HIP(ctx, tp.alloc(n_terms_new + 1));
        SS_HIP(ctx, hipMemcpyAsync(tp.p, idx->term_ptr.p, (T + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_fill_u64, dim3(grid_for(n_terms_new - T)), dim3(TPB), 0, st, tp.p, T + 1, n_terms_new + 1, P);
        SS_HIP(ctx, hipStreamSynchronize(st));
        idx->term_ptr = std::move(tp);
        idx->h_term_ptr.resize(n_terms_new + 1, P);
        idx->n_terms = n_terms_new;
    }
    if (n_docs_new > N) {                                    // new docs: no postings yet, magnitude 0
        ss::DevBuf<double> m, m2;
        SS_HIP(ctx, m.alloc(n_docs_new));
        SS_HIP(ctx, m2.alloc(n_docs_new));
        SS_HIP(ctx, hipMemsetAsync(m.p, 0, n_docs_new * sizeof(double), st));
        SS_HIP(ctx, hipMemsetAsync(m2.p, 0, n_docs_new * sizeof(double), st));
        SS_HIP(ctx, hipMemcpyAsync(m.p, idx->mag.p, N * sizeof(double), hipMemcpyDeviceToDevice, st));
        if (idx->mag2.p) SS_HIP(ctx, hipMemcpyAsync(m2.p, idx->mag2.p, N * sizeof(double), hipMemcpyDeviceToDevice, st));
        SS_HIP(ctx, hipStreamSynchronize(st));
        idx->mag = std::move(m);
        idx->mag2 = std::move(m2);
        idx->n_docs = n_docs_new;
    }
    return SS_OK;
}

int32_t ss_index_read_magnitudes(ss_index* idx, uint64_t n, const uint32_t* docs, double* mag_out) {
    if (!idx) return SS_ERR_INVALID;
    ss_ctx* ctx = idx->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (n && (!docs || !mag_out)) return ctx->fail(SS_ERR_INVALID, "ss_index_read_magnitudes: NULL argument");
    if (n == 0) return SS_OK;
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ss::DevBuf<uint32_t> d_docs;
    ss::DevBuf<double> d_out;
    SS_HIP(ctx, d_docs.alloc(n));
    SS_HIP(ctx, d_out.alloc(n));
    SS_HIP(ctx, hipMemcpyAsync(d_docs.p, docs, n * sizeof(uint32_t), hipMemcpyDefault, st));
    hipLaunchKernelGGL(k_gather_f64, dim3(grid_for(n)), dim3(TPB), 0, st, (const double*)idx->mag.p, (const uint32_t*)d_docs.p, n, idx->n_docs, d_out.p);
    SS_HIP(ctx, hipMemcpyAsync(mag_out, d_out.p, n * sizeof(double), hipMemcpyDefault, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    return SS_OK;
}

}  // extern "C"
