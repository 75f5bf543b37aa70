// tfidf.hip — inverted-index upload + TF-IDF weight/magnitude build for gfx950.
//
// Replaces ranking/term_weighting.go:10-57 (UpdateTermWeights) and the sqrt of
// saveMagnitude (:72,:97,:105):
//   idf   = float32(math.Log2(totalDocs / float64(len(val))))        (:37)
//   w     = tf * idf                      float32 multiply, in place  (:42)
//   mag2[doc] += float64(w * w)           float32 product, f64 sum    (:44)
//   mag   = sqrt(mag2)                                                (:72)
// HBM-bound: 12 B/posting (doc id, read w, write w) + 8 B per doc and term.
#include "index.hpp"
#include <cstdlib>
#include <algorithm>

#include <algorithm>
#include <memory>

namespace {

constexpr int TPB = 256;

// Go's math.Log / math.Log2 (go1.12 src/math/log.go, log10.go; FreeBSD e_log.c algorithm),
// restated for the device.  Same IEEE operation sequence as oracle/oracle.c:orc_go_log2
// (this file is compiled with -ffp-contract=off), so idf is bit-identical to the oracle's.
__device__ __forceinline__ double go_log(double x) {
    const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                 L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    const double Sqrt2Half = 0.70710678118654752440084436210484903928483593768847;
    if (x != x || (isinf(x) && x > 0)) return x;
    if (x < 0) return __builtin_nan("");
    if (x == 0) return -__builtin_inf();
    int ki;
    double f1 = frexp(x, &ki);
    if (f1 < Sqrt2Half) { f1 *= 2; ki--; }
    const double f = f1 - 1;
    const double k = (double)ki;
    const double s = f / (2 + f);
    const double s2 = s * s;
    const double s4 = s2 * s2;
    const double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    const double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}
__device__ __forceinline__ double go_log2(double x) {
    const double InvLn2 = 1.44269504088896340735992468100189214;
    int e;
    const double frac = frexp(x, &e);
    if (frac == 0.5) return (double)(e - 1);
    return go_log(frac) * InvLn2 + (double)e;
}

// df_global != nullptr: this table is one doc-range shard and len(docs) of term_weighting.go:37 is the length of
// the term's WHOLE list, summed over the shards by the host (one all-reduce)
__global__ void k_idf(const uint64_t* __restrict__ term_ptr, const uint64_t* __restrict__ df_global, uint64_t n_terms,
                      double total_docs, float* __restrict__ idf) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_terms) return;
    const double df = df_global ? (double)df_global[t] : (double)(term_ptr[t + 1] - term_ptr[t]);
    idf[t] = (float)go_log2(total_docs / df);                       // term_weighting.go:37
}

// validate: term_ptr non-decreasing, doc ids in range, count adjacent non-ascending pairs
__global__ void k_check_ptr(const uint64_t* __restrict__ term_ptr, uint64_t n_terms, const uint32_t* __restrict__ post_doc,
                            unsigned long long* __restrict__ n_boundary_desc, uint32_t* __restrict__ err) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_terms) return;
    const uint64_t a = term_ptr[t], b = term_ptr[t + 1];
    if (b < a) { atomicOr(err, 1u); return; }
    // a legitimate descent can only sit at the start of a non-empty list
    if (b > a && a > 0 && post_doc[a] <= post_doc[a - 1]) atomicAdd(n_boundary_desc, 1ull);
}
__global__ void k_check_df(const uint64_t* __restrict__ term_ptr, const uint64_t* __restrict__ df, uint64_t n_terms,
                           uint32_t* __restrict__ err) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_terms) return;
    if (df[t] < term_ptr[t + 1] - term_ptr[t]) atomicOr(err, 1u);
}
// positional postings: pos_ptr must be non-decreasing (k_phrase_match reads pos[pos_ptr[i] .. pos_ptr[i+1]))
__global__ void k_check_pos(const uint64_t* __restrict__ pos_ptr, uint64_t n_post, uint32_t* __restrict__ err) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (; i < n_post; i += stride) bad = bad || pos_ptr[i + 1] < pos_ptr[i];
    if (bad) atomicOr(err, 1u);
}
__global__ void k_check_docs(const uint32_t* __restrict__ post_doc, uint64_t n_post, uint64_t n_docs,
                             unsigned long long* __restrict__ n_desc, uint32_t* __restrict__ err) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long local = 0;
    bool bad = false;
    for (; i < n_post; i += stride) {
        const uint32_t d = post_doc[i];
        if (d >= n_docs) bad = true;
        if (i > 0 && d <= post_doc[i - 1]) local++;
    }
    if (bad) atomicOr(err, 2u);
    // wave-reduce then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(n_desc, local);
}

// One block per chunk of CH consecutive postings.  The block finds the terms its chunk spans
// with two binary searches, then every posting finds its own term inside that (short) range.
constexpr int CH = TPB * 16;
__global__ __launch_bounds__(TPB) void k_weight(const uint64_t* __restrict__ term_ptr, uint64_t n_terms,
                                                const uint32_t* __restrict__ post_doc, float* __restrict__ post_w,
                                                const float* __restrict__ idf, uint64_t n_post, double* __restrict__ mag2) {
    __shared__ uint64_t s_t[2];
    const uint64_t base = (uint64_t)blockIdx.x * CH;
    const uint64_t last = min(base + CH, n_post) - 1;
    if (threadIdx.x < 2) {
        // largest t with term_ptr[t] <= target
        const uint64_t target = threadIdx.x == 0 ? base : last;
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
    for (int j = 0; j < CH / TPB; j++) {
        const uint64_t i = base + (uint64_t)j * TPB + threadIdx.x;
        if (i >= n_post) break;
        uint64_t lo = t_lo, hi = t_hi + 1;   // term_ptr[lo] <= i < term_ptr[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (term_ptr[mid] <= i) lo = mid; else hi = mid;
        }
        const float w = post_w[i] * idf[lo];                          // term_weighting.go:42
        post_w[i] = w;
        const float sq = w * w;                                       // :44 (float32 product)
        unsafeAtomicAdd(&mag2[post_doc[i]], (double)sq);              // :44 (float64 accumulate)
    }
}

// ---- bucketed magnitude pass (large tables) -----------------------------------------------------------------
// One float64 atomic per posting into a random 8-byte word of mag2 costs a memory-side read-modify-write each
// (26 G/s on this part: 24.8 ms for the 641M-posting body table).  Instead the postings are partitioned once by
// doc range ("bucket" = 2^shift consecutive docs) and every bucket is summed in LDS:
//   k_weight_count   w = tf*idf in place, per-block LDS histogram of the buckets, one global add per touched bucket
//   k_bucket_offsets exclusive scan of the <= 4096 bucket counts
//   k_scatter        {doc, float32(w*w)} of every posting to its bucket's region (block claims a run per bucket,
//                    LDS ticket per posting)
//   k_bucket_sum     one workgroup per bucket: float64 LDS accumulators, sqrt, write mag
// float32 squares summed in float64 are exact, so the order inside a bucket does not matter (same as the atomics).
constexpr int NB_MAX = 4096;                 // most buckets (LDS histogram of a block)
constexpr int PER_THREAD = CH / TPB;         // postings per thread and block

__device__ __forceinline__ void chunk_term_range(const uint64_t* __restrict__ term_ptr, uint64_t n_terms, uint64_t base, uint64_t last,
                                                 uint64_t* s_t) {
    if (threadIdx.x < 2) {
        const uint64_t target = threadIdx.x == 0 ? base : last;      // largest t with term_ptr[t] <= target
        uint64_t lo = 0, hi = n_terms;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (term_ptr[mid] <= target) lo = mid; else hi = mid;
        }
        s_t[threadIdx.x] = lo;
    }
}

__global__ __launch_bounds__(TPB) void k_weight_count(const uint64_t* __restrict__ term_ptr, uint64_t n_terms,
                                                      const uint32_t* __restrict__ post_doc, float* __restrict__ post_w,
                                                      const float* __restrict__ idf, uint64_t n_post, int shift, uint32_t nb,
                                                      uint32_t* __restrict__ cnt) {
    __shared__ uint64_t s_t[2];
    __shared__ uint32_t s_hist[NB_MAX];
    const uint64_t base = (uint64_t)blockIdx.x * CH;
    const uint64_t last = min(base + CH, n_post) - 1;
    for (uint32_t b = threadIdx.x; b < nb; b += TPB) s_hist[b] = 0;
    chunk_term_range(term_ptr, n_terms, base, last, s_t);
    __syncthreads();
    const uint64_t t_lo = s_t[0], t_hi = s_t[1];
#pragma unroll 4
    for (int j = 0; j < PER_THREAD; j++) {
        const uint64_t i = base + (uint64_t)j * TPB + threadIdx.x;
        if (i >= n_post) break;
        uint64_t lo = t_lo, hi = t_hi + 1;   // term_ptr[lo] <= i < term_ptr[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (term_ptr[mid] <= i) lo = mid; else hi = mid;
        }
        post_w[i] = post_w[i] * idf[lo];                              // term_weighting.go:42
        atomicAdd(&s_hist[post_doc[i] >> shift], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += TPB)
        if (s_hist[b]) atomicAdd(&cnt[b], s_hist[b]);
}

// off[b] = sum of cnt[< b] (off[nb] = total); cursor[b] = off[b].  One block, nb <= NB_MAX.
__global__ __launch_bounds__(1024) void k_bucket_offsets(const uint32_t* __restrict__ cnt, uint32_t nb, uint32_t* __restrict__ off,
                                                         uint32_t* __restrict__ cursor) {
    __shared__ uint32_t s[NB_MAX];
    __shared__ uint32_t part[1024];
    const uint32_t per = (nb + 1023) / 1024;                          // consecutive buckets per thread
    uint32_t sum = 0;
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t b = threadIdx.x * per + j;
        const uint32_t c = b < nb ? cnt[b] : 0;
        if (b < nb) s[b] = sum;
        sum += c;
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int t = 0; t < 1024; t++) { const uint32_t v = part[t]; part[t] = run; run += v; }
        off[nb] = run;
    }
    __syncthreads();
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t b = threadIdx.x * per + j;
        if (b < nb) { const uint32_t o = s[b] + part[threadIdx.x]; off[b] = o; cursor[b] = o; }
    }
}

#ifndef SS_SC_TPB
#define SS_SC_TPB 1024
#endif
constexpr int SC_TPB = SS_SC_TPB;             // bigger blocks: longer runs per bucket where short lists spread a block over many buckets
constexpr int SC_CH = SC_TPB * PER_THREAD;
__global__ __launch_bounds__(SC_TPB) void k_scatter(const uint32_t* __restrict__ post_doc, const float* __restrict__ post_w, uint64_t n_post,
                                                 int shift, uint32_t nb, uint32_t* __restrict__ cursor, uint2* __restrict__ out) {
    __shared__ uint32_t s_hist[NB_MAX];
    __shared__ uint32_t s_base[NB_MAX];
    const uint64_t base = (uint64_t)blockIdx.x * SC_CH;
    for (uint32_t b = threadIdx.x; b < nb; b += SC_TPB) s_hist[b] = 0;
    __syncthreads();
    uint32_t doc[PER_THREAD];
#pragma unroll
    for (int j = 0; j < PER_THREAD; j++) {
        const uint64_t i = base + (uint64_t)j * SC_TPB + threadIdx.x;
        doc[j] = 0xFFFFFFFFu;
        if (i < n_post) {
            doc[j] = post_doc[i];
            atomicAdd(&s_hist[doc[j] >> shift], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += SC_TPB) {
        const uint32_t c = s_hist[b];
        if (c) { s_base[b] = atomicAdd(&cursor[b], c); s_hist[b] = 0; }    // this block's run inside bucket b
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER_THREAD; j++) {
        const uint64_t i = base + (uint64_t)j * SC_TPB + threadIdx.x;
        if (doc[j] != 0xFFFFFFFFu) {
            const uint32_t b = doc[j] >> shift;
            const float w = post_w[i];
            const float sq = w * w;                                    // term_weighting.go:44 (float32 product)
            const uint32_t r = atomicAdd(&s_hist[b], 1u);
            out[(uint64_t)s_base[b] + r] = make_uint2(doc[j], __float_as_uint(sq));
        }
    }
}

// magnitudes only (ss_index_refresh_magnitudes): the weights are already final
__global__ __launch_bounds__(TPB) void k_count_only(const uint32_t* __restrict__ post_doc, uint64_t n_post, int shift, uint32_t nb, uint32_t* __restrict__ cnt) {
    __shared__ uint32_t s_hist[NB_MAX];
    const uint64_t base = (uint64_t)blockIdx.x * CH;
    for (uint32_t b = threadIdx.x; b < nb; b += TPB) s_hist[b] = 0;
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < PER_THREAD; j++) {
        const uint64_t i = base + (uint64_t)j * TPB + threadIdx.x;
        if (i >= n_post) break;
        atomicAdd(&s_hist[post_doc[i] >> shift], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += TPB)
        if (s_hist[b]) atomicAdd(&cnt[b], s_hist[b]);
}
__global__ void k_sumsq_atomic(const uint32_t* __restrict__ post_doc, const float* __restrict__ post_w, uint64_t n_post, double* __restrict__ mag2) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_post; i += (uint64_t)gridDim.x * blockDim.x) {
        const float w = post_w[i];
        const float sq = w * w;                                       // term_weighting.go:44 (float32 product)
        unsafeAtomicAdd(&mag2[post_doc[i]], (double)sq);
    }
}

constexpr int TPB_B = 512;
__global__ __launch_bounds__(TPB_B) void k_bucket_sum(const uint2* __restrict__ packed, const uint32_t* __restrict__ off, uint64_t n_docs,
                                                      int shift, double* __restrict__ mag) {
    extern __shared__ double acc[];                                    // [1 << shift]
    const uint32_t bd = 1u << shift, b = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < bd; i += TPB_B) acc[i] = 0.0;
    __syncthreads();
    const uint32_t lo = off[b], hi = off[b + 1];
    for (uint32_t i = lo + threadIdx.x; i < hi; i += TPB_B) {
        const uint2 r = packed[i];
        atomicAdd(&acc[r.x & (bd - 1)], (double)__uint_as_float(r.y)); // :44 (float64 accumulate; LDS)
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bd; i += TPB_B) {
        const uint64_t d = (uint64_t)b * bd + i;
        if (d < n_docs) mag[d] = sqrt(acc[i]);                         // term_weighting.go:72
    }
}

__global__ void k_sqrt(double* __restrict__ v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = sqrt(v[i]);                                     // term_weighting.go:72
}

}  // namespace

extern "C" {

int32_t ss_index_create(ss_ctx* ctx, uint64_t n_docs, uint64_t n_terms, const uint64_t* term_ptr,
                        const uint32_t* post_doc, const float* post_tf, ss_index** out) {
    if (!ctx) return SS_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!out) return ctx->fail(SS_ERR_INVALID, "ss_index_create: out is NULL");
    *out = nullptr;
    if (!term_ptr) return ctx->fail(SS_ERR_INVALID, "ss_index_create: term_ptr is NULL");
    if (n_docs == 0 || n_docs >= 0xFFFFFFF0ull || n_terms >= 0xFFFFFFF0ull)
        return ctx->fail(SS_ERR_INVALID, "ss_index_create: n_docs/n_terms out of range");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    std::unique_ptr<ss_index> idx(new (std::nothrow) ss_index());
    if (!idx) return ctx->fail(SS_ERR_OOM, "ss_index_create: host OOM");
    idx->ctx = ctx;
    idx->n_docs = n_docs;
    idx->n_terms = n_terms;
    idx->h_term_ptr.resize(n_terms + 1);
    SS_HIP(ctx, idx->term_ptr.alloc(n_terms + 1));
    SS_HIP(ctx, hipMemcpyAsync(idx->term_ptr.p, term_ptr, (n_terms + 1) * sizeof(uint64_t), hipMemcpyDefault, st));
    SS_HIP(ctx, hipMemcpyAsync(idx->h_term_ptr.data(), idx->term_ptr.p, (n_terms + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    if (idx->h_term_ptr[0] != 0) return ctx->fail(SS_ERR_INVALID, "ss_index_create: term_ptr[0] != 0");
    const uint64_t P = idx->h_term_ptr[n_terms];
    idx->n_post = P;
    if (P && (!post_doc || !post_tf)) return ctx->fail(SS_ERR_INVALID, "ss_index_create: NULL postings");
    SS_HIP(ctx, idx->post_doc.alloc(P));
    SS_HIP(ctx, idx->post_w.alloc(P));
    SS_HIP(ctx, idx->mag.alloc(n_docs));
    if (P) {
        SS_HIP(ctx, hipMemcpyAsync(idx->post_doc.p, post_doc, P * sizeof(uint32_t), hipMemcpyDefault, st));
        SS_HIP(ctx, hipMemcpyAsync(idx->post_w.p, post_tf, P * sizeof(float), hipMemcpyDefault, st));
    }
    SS_HIP(ctx, hipMemsetAsync(idx->mag.p, 0, n_docs * sizeof(double), st));

    // validation (the scorer relies on strictly ascending doc ids inside a term)
    ss::DevBuf<unsigned long long> d_cnt;
    ss::DevBuf<uint32_t> d_err;
    SS_HIP(ctx, d_cnt.alloc(2));
    SS_HIP(ctx, d_err.alloc(1));
    SS_HIP(ctx, hipMemsetAsync(d_cnt.p, 0, 2 * sizeof(unsigned long long), st));
    SS_HIP(ctx, hipMemsetAsync(d_err.p, 0, sizeof(uint32_t), st));
    if (n_terms) hipLaunchKernelGGL(k_check_ptr, dim3(ss::div_up(n_terms, TPB)), dim3(TPB), 0, st, idx->term_ptr.p, n_terms,
                                    idx->post_doc.p, d_cnt.p, d_err.p);
    if (P) hipLaunchKernelGGL(k_check_docs, dim3(std::min<unsigned>(ss::div_up(P, TPB), 16384u)), dim3(TPB), 0, st,
                              idx->post_doc.p, P, n_docs, d_cnt.p + 1, d_err.p);
    unsigned long long h_cnt[2] = {0, 0};
    uint32_t h_err = 0;
    SS_HIP(ctx, hipMemcpyAsync(h_cnt, d_cnt.p, sizeof(h_cnt), hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipMemcpyAsync(&h_err, d_err.p, sizeof(h_err), hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    SS_HIP(ctx, hipGetLastError());
    if (h_err & 1) return ctx->fail(SS_ERR_INVALID, "ss_index_create: term_ptr is not non-decreasing");
    if (h_err & 2) return ctx->fail(SS_ERR_INVALID, "ss_index_create: posting holds a doc id >= n_docs");
    if (h_cnt[1] != h_cnt[0])
        return ctx->fail(SS_ERR_UNSORTED, "ss_index_create: %llu posting(s) not strictly ascending by doc id inside a term",
                         h_cnt[1] - h_cnt[0]);
    *out = idx.release();
    return SS_OK;
}

int32_t ss_index_destroy(ss_index* idx) {
    if (!idx) return SS_ERR_INVALID;
    ss_ctx* ctx = idx->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (idx->users > 0) return ctx->fail(SS_ERR_STATE, "ss_index_destroy: index still used by %d scorer(s)", idx->users);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    delete idx;
    return SS_OK;
}

int32_t ss_tfidf_build(ss_index* idx, uint64_t total_docs, float* w_out, double* mag_out, float* idf_out) {
    if (!idx) return SS_ERR_INVALID;
    ss_ctx* ctx = idx->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t P = idx->n_post, T = idx->n_terms, N = idx->n_docs;
    // every allocation happens BEFORE the start event: ss_last_kernel_ms(2) brackets device work only
    // (a 5 GB hipMalloc inside the window once put ~1 s of host-side allocator time into the "kernel" time)
    ss::DevBuf<float> idf;
    SS_HIP(ctx, idf.alloc(T));
    // large tables: bucketed magnitude pass (no global float64 atomics); small ones: one atomic per posting
    int shift = 13;                                                   // 8192 docs per bucket = 64 KB of float64 LDS accumulators
    if ((N >> shift) >= (uint64_t)NB_MAX) shift = 14;
    const uint64_t nb64 = ss::div_up(std::max<uint64_t>(N, 1), (uint64_t)1 << shift);
    // SS_TFIDF_BUCKET_MIN (tests, A/B): smallest table that takes the bucketed pass; a huge value forces the atomics
    uint64_t min_p = (uint64_t)1 << 22;
    if (const char* e = std::getenv("SS_TFIDF_BUCKET_MIN")) min_p = std::strtoull(e, nullptr, 10);
    const bool bucketed = P >= std::max<uint64_t>(min_p, 1) && P < ((uint64_t)1 << 32) && nb64 <= (uint64_t)NB_MAX;
    const uint32_t nb = (uint32_t)nb64;
    ss::DevBuf<uint32_t> b_cnt, b_off, b_cur;
    ss::DevBuf<uint2> b_packed;
    if (bucketed) {
        SS_HIP(ctx, b_cnt.alloc(nb));
        SS_HIP(ctx, b_off.alloc(nb + 1));
        SS_HIP(ctx, b_cur.alloc(nb));
        SS_HIP(ctx, b_packed.alloc(P));
        if (ctx->tfidf_bucket_lds < (1 << shift) * 8) {
            SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket_sum), hipFuncAttributeMaxDynamicSharedMemorySize, (1 << shift) * 8));
            ctx->tfidf_bucket_lds = (1 << shift) * 8;
        }
    }
    SS_HIP(ctx, hipStreamSynchronize(st));       // allocator work (and anything queued before) is over when the clock starts
    SS_HIP(ctx, hipEventRecord(ctx->ev[2][0], st));
    SS_HIP(ctx, hipMemsetAsync(idx->mag.p, 0, N * sizeof(double), st));
    if (T) hipLaunchKernelGGL(k_idf, dim3(ss::div_up(T, TPB)), dim3(TPB), 0, st, idx->term_ptr.p,
                              idx->has_df_global ? idx->df_global.p : nullptr, T, (double)total_docs, idf.p);
    if (bucketed) {
        SS_HIP(ctx, hipMemsetAsync(b_cnt.p, 0, nb * sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_weight_count, dim3(ss::div_up(P, CH)), dim3(TPB), 0, st, idx->term_ptr.p, T, idx->post_doc.p, idx->post_w.p,
                           idf.p, P, shift, nb, b_cnt.p);
        hipLaunchKernelGGL(k_bucket_offsets, dim3(1), dim3(1024), 0, st, b_cnt.p, nb, b_off.p, b_cur.p);
        hipLaunchKernelGGL(k_scatter, dim3(ss::div_up(P, SC_CH)), dim3(SC_TPB), 0, st, idx->post_doc.p, idx->post_w.p, P, shift, nb, b_cur.p, b_packed.p);
        hipLaunchKernelGGL(k_bucket_sum, dim3(nb), dim3(TPB_B), (size_t)(1 << shift) * 8, st, b_packed.p, b_off.p, N, shift, idx->mag.p);
    } else {
        if (P) hipLaunchKernelGGL(k_weight, dim3(ss::div_up(P, CH)), dim3(TPB), 0, st, idx->term_ptr.p, T, idx->post_doc.p,
                                  idx->post_w.p, idf.p, P, idx->mag.p);
        hipLaunchKernelGGL(k_sqrt, dim3(ss::div_up(N, TPB)), dim3(TPB), 0, st, idx->mag.p, N);
    }
    SS_HIP(ctx, hipEventRecord(ctx->ev[2][1], st));
    ctx->ev_valid[2] = true;
    SS_HIP(ctx, hipGetLastError());
    idx->weighted = true;
    if (w_out && P) SS_HIP(ctx, hipMemcpyAsync(w_out, idx->post_w.p, P * sizeof(float), hipMemcpyDefault, st));
    if (mag_out) SS_HIP(ctx, hipMemcpyAsync(mag_out, idx->mag.p, N * sizeof(double), hipMemcpyDefault, st));
    if (idf_out && T) SS_HIP(ctx, hipMemcpyAsync(idf_out, idf.p, T * sizeof(float), hipMemcpyDefault, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    return SS_OK;
}

int32_t ss_index_refresh_magnitudes(ss_index* idx, double* mag_out) {
    if (!idx) return SS_ERR_INVALID;
    ss_ctx* ctx = idx->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t P = idx->n_post, N = idx->n_docs;
    int shift = 13;
    if ((N >> shift) >= (uint64_t)NB_MAX) shift = 14;
    const uint64_t nb64 = ss::div_up(std::max<uint64_t>(N, 1), (uint64_t)1 << shift);
    const bool bucketed = P >= ((uint64_t)1 << 22) && P < ((uint64_t)1 << 32) && nb64 <= (uint64_t)NB_MAX;
    SS_HIP(ctx, hipMemsetAsync(idx->mag.p, 0, N * sizeof(double), st));
    if (bucketed) {
        const uint32_t nb = (uint32_t)nb64;
        ss::DevBuf<uint32_t> b_cnt, b_off, b_cur;
        ss::DevBuf<uint2> b_packed;
        SS_HIP(ctx, b_cnt.alloc(nb));
        SS_HIP(ctx, b_off.alloc(nb + 1));
        SS_HIP(ctx, b_cur.alloc(nb));
        SS_HIP(ctx, b_packed.alloc(P));
        if (ctx->tfidf_bucket_lds < (1 << shift) * 8) {
            SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket_sum), hipFuncAttributeMaxDynamicSharedMemorySize, (1 << shift) * 8));
            ctx->tfidf_bucket_lds = (1 << shift) * 8;
        }
        SS_HIP(ctx, hipMemsetAsync(b_cnt.p, 0, nb * sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_count_only, dim3(ss::div_up(P, CH)), dim3(TPB), 0, st, (const uint32_t*)idx->post_doc.p, P, shift, nb, b_cnt.p);
        hipLaunchKernelGGL(k_bucket_offsets, dim3(1), dim3(1024), 0, st, b_cnt.p, nb, b_off.p, b_cur.p);
        hipLaunchKernelGGL(k_scatter, dim3(ss::div_up(P, SC_CH)), dim3(SC_TPB), 0, st, idx->post_doc.p, idx->post_w.p, P, shift, nb, b_cur.p, b_packed.p);
        hipLaunchKernelGGL(k_bucket_sum, dim3(nb), dim3(TPB_B), (size_t)(1 << shift) * 8, st, b_packed.p, b_off.p, N, shift, idx->mag.p);
        SS_HIP(ctx, hipGetLastError());
        SS_HIP(ctx, hipStreamSynchronize(st));
    } else {
        if (P) hipLaunchKernelGGL(k_sumsq_atomic, dim3(std::min<unsigned>(ss::div_up(P, TPB), 16384u)), dim3(TPB), 0, st, (const uint32_t*)idx->post_doc.p,
                                  (const float*)idx->post_w.p, P, idx->mag.p);
        hipLaunchKernelGGL(k_sqrt, dim3(ss::div_up(N, TPB)), dim3(TPB), 0, st, idx->mag.p, N);
        SS_HIP(ctx, hipGetLastError());
    }
    idx->weighted = true;
    if (mag_out) SS_HIP(ctx, hipMemcpyAsync(mag_out, idx->mag.p, N * sizeof(double), hipMemcpyDefault, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    return SS_OK;
}

int32_t ss_index_get_info(const ss_index* idx, uint64_t* n_docs, uint64_t* n_terms, uint64_t* n_post) {
    if (!idx) return SS_ERR_INVALID;
    if (n_docs) *n_docs = idx->n_docs;
    if (n_terms) *n_terms = idx->n_terms;
    if (n_post) *n_post = idx->n_post;
    return SS_OK;
}

int32_t ss_index_read(ss_index* idx, uint64_t* term_ptr_out, uint32_t* post_doc_out, float* post_w_out) {
    if (!idx) return SS_ERR_INVALID;
    ss_ctx* ctx = idx->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (term_ptr_out) SS_HIP(ctx, hipMemcpyAsync(term_ptr_out, idx->term_ptr.p, (idx->n_terms + 1) * sizeof(uint64_t), hipMemcpyDefault, st));
    if (post_doc_out && idx->n_post) SS_HIP(ctx, hipMemcpyAsync(post_doc_out, idx->post_doc.p, idx->n_post * sizeof(uint32_t), hipMemcpyDefault, st));
    if (post_w_out && idx->n_post) SS_HIP(ctx, hipMemcpyAsync(post_w_out, idx->post_w.p, idx->n_post * sizeof(float), hipMemcpyDefault, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    return SS_OK;
}

int32_t ss_index_set_doc_freq(ss_index* idx, const uint64_t* df) {
    if (!idx) return SS_ERR_INVALID;
    ss_ctx* ctx = idx->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (idx->weighted) return ctx->fail(SS_ERR_STATE, "ss_index_set_doc_freq: table is already weighted");
    if (!df) { idx->has_df_global = false; idx->df_global.release(); return SS_OK; }
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t T = idx->n_terms;
    SS_HIP(ctx, idx->df_global.alloc(T));
    ss::DevBuf<uint32_t> err;
    SS_HIP(ctx, err.alloc(1));
    SS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(uint32_t), st));
    if (T) {
        SS_HIP(ctx, hipMemcpyAsync(idx->df_global.p, df, T * sizeof(uint64_t), hipMemcpyDefault, st));
        hipLaunchKernelGGL(k_check_df, dim3(ss::div_up(T, TPB)), dim3(TPB), 0, st, idx->term_ptr.p, idx->df_global.p, T, err.p);
    }
    uint32_t h_err = 0;
    SS_HIP(ctx, hipMemcpyAsync(&h_err, err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    if (h_err) {
        idx->df_global.release();
        idx->has_df_global = false;
        return ctx->fail(SS_ERR_INVALID, "ss_index_set_doc_freq: a document frequency is smaller than the local list");
    }
    idx->has_df_global = true;
    return SS_OK;
}

int32_t ss_index_set_positions(ss_index* idx, const uint64_t* pos_ptr, const float* pos) {
    if (!idx) return SS_ERR_INVALID;
    ss_ctx* ctx = idx->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!pos_ptr) return ctx->fail(SS_ERR_INVALID, "ss_index_set_positions: pos_ptr is NULL");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t P = idx->n_post;
    SS_HIP(ctx, idx->pos_ptr.alloc(P + 1));
    SS_HIP(ctx, hipMemcpyAsync(idx->pos_ptr.p, pos_ptr, (P + 1) * sizeof(uint64_t), hipMemcpyDefault, st));
    uint64_t ends[2] = {0, 0};
    SS_HIP(ctx, hipMemcpyAsync(&ends[0], idx->pos_ptr.p, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipMemcpyAsync(&ends[1], idx->pos_ptr.p + P, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    if (ends[0] != 0) { idx->pos_ptr.release(); return ctx->fail(SS_ERR_INVALID, "ss_index_set_positions: pos_ptr[0] != 0"); }
    if (P) {
        // monotone + first 0 + last = total  =>  every range lies inside pos[0, total)
        ss::DevBuf<uint32_t> err;
        SS_HIP(ctx, err.alloc(1));
        SS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_check_pos, dim3(std::min<unsigned>(ss::div_up(P, TPB), 16384u)), dim3(TPB), 0, st, idx->pos_ptr.p, P, err.p);
        uint32_t h_err = 0;
        SS_HIP(ctx, hipMemcpyAsync(&h_err, err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        SS_HIP(ctx, hipStreamSynchronize(st));
        if (h_err) { idx->pos_ptr.release(); return ctx->fail(SS_ERR_INVALID, "ss_index_set_positions: pos_ptr is not non-decreasing"); }
    }
    if (ends[1] && !pos) { idx->pos_ptr.release(); return ctx->fail(SS_ERR_INVALID, "ss_index_set_positions: pos is NULL"); }
    SS_HIP(ctx, idx->pos.alloc(ends[1]));
    if (ends[1]) SS_HIP(ctx, hipMemcpyAsync(idx->pos.p, pos, ends[1] * sizeof(float), hipMemcpyDefault, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    return SS_OK;
}

int32_t ss_index_set_weighted(ss_index* idx, const double* mag) {
    if (!idx) return SS_ERR_INVALID;
    ss_ctx* ctx = idx->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!mag) return ctx->fail(SS_ERR_INVALID, "ss_index_set_weighted: mag is NULL");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    SS_HIP(ctx, hipMemcpyAsync(idx->mag.p, mag, idx->n_docs * sizeof(double), hipMemcpyDefault, ctx->stream));
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    idx->weighted = true;
    return SS_OK;
}

}  // extern "C"
