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
//   k_weight_count   w = tf*idf in place; every block owns a contiguous range of postings and leaves its bucket histogram
//                    as one column of a [bucket][block] count matrix
//   k_bucket_rowscan + k_bucket_offsets   exclusive scans: inside a bucket over the blocks, then over the buckets
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

// ---- head lists: long posting lists are summed bucket-major straight from the table -------------------------------
// Every posting list is strictly ascending by doc (checked at ss_index_create, kept by the deltas), so the postings of a
// LONG list that fall into one bucket are one contiguous run of the list.  Lists whose average run is at least `min_run`
// postings ("head" lists: under Zipf a few hundred terms that hold about half of all postings) skip the partition
// altogether: k_weight_count and k_scatter pass over them, and k_bucket_sum reads bucket b's run of every head list straight
// from post_doc / post_w, weighting it on the way (w = tf*idf written once) — 12 bytes per head posting, the algorithmic
// minimum, instead of the 36 a partitioned posting costs.  Which lists are head only moves work between two exact paths.
#ifndef SS_HEAD_CAP
#define SS_HEAD_CAP 1024
#endif
constexpr int HEAD_CAP = SS_HEAD_CAP;               // most head lists (their run table sits in LDS beside the accumulators: 12 B each)
struct HeadArgs {
    const uint32_t* n;                       // number of head lists (device word; 0 = no head path)
    const uint32_t* term;                    // [HEAD_CAP] their term ids, ascending
    const uint64_t* hs;                      // [HEAD_CAP] term_ptr[term]      first posting
    const uint64_t* he;                      // [HEAD_CAP] term_ptr[term + 1]  one past the last
    const uint32_t* bounds;                  // [nb + 1][HEAD_CAP]: postings of list i with doc < b << shift
};
// Threshold: the smallest power-of-two multiple of `thr` that leaves at most HEAD_CAP lists (so that the LONGEST lists are the
// head lists when many qualify).  hist[j] counts the lists with length >= thr << j.
constexpr int HEAD_LEVELS = 24;
__global__ void k_head_hist(const uint64_t* __restrict__ term_ptr, uint64_t n_terms, uint64_t thr, uint32_t* __restrict__ hist) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_terms) return;
    const uint64_t len = term_ptr[t + 1] - term_ptr[t];
    for (int j = 0; j < HEAD_LEVELS && len >= (thr << j); j++) atomicAdd(&hist[j], 1u);    // a few thousand lists get here at all
}
__global__ void k_head_select(const uint64_t* __restrict__ term_ptr, uint64_t n_terms, uint64_t thr, const uint32_t* __restrict__ hist,
                              uint32_t* __restrict__ n_head, uint32_t* __restrict__ head_term) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_terms) return;
    int j = 0;
    while (j + 1 < HEAD_LEVELS && hist[j] > (uint32_t)HEAD_CAP) j++;
    if (term_ptr[t + 1] - term_ptr[t] >= (thr << j)) {
        const uint32_t slot = atomicAdd(n_head, 1u);
        if (slot < (uint32_t)HEAD_CAP) head_term[slot] = (uint32_t)t;     // (beyond the cap only if 2^23 * thr still leaves too many)
    }
}
// one workgroup: clamp the count, sort the term ids ascending (bitonic, LDS), derive the posting ranges
__global__ __launch_bounds__(1024) void k_head_sort(const uint64_t* __restrict__ term_ptr, uint32_t* __restrict__ n_head,
                                                    uint32_t* __restrict__ head_term, uint64_t* __restrict__ hs, uint64_t* __restrict__ he) {
    __shared__ uint32_t s[HEAD_CAP];
    const uint32_t n = min(*n_head, (uint32_t)HEAD_CAP);
    for (uint32_t i = threadIdx.x; i < (uint32_t)HEAD_CAP; i += 1024) s[i] = i < n ? head_term[i] : 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t k = 2; k <= (uint32_t)HEAD_CAP; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < (uint32_t)HEAD_CAP; i += 1024) {
                const uint32_t x = i ^ j;
                if (x > i) {
                    const uint32_t a = s[i], b = s[x];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s[i] = b; s[x] = a; }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < (uint32_t)HEAD_CAP; i += 1024) {
        const uint32_t t = s[i];
        head_term[i] = t;
        hs[i] = i < n ? term_ptr[t] : ~0ull;
        he[i] = i < n ? term_ptr[t + 1] : ~0ull;
    }
    if (threadIdx.x == 0) *n_head = n;
}
// bounds[b][i] = lower bound of doc (b << shift) inside head list i, relative to the list's start; b = 0 .. nb
__global__ void k_head_bounds(const uint32_t* __restrict__ post_doc, const uint32_t* __restrict__ n_head, const uint64_t* __restrict__ hs,
                              const uint64_t* __restrict__ he, int shift, uint32_t nb, uint32_t* __restrict__ bounds) {
    const uint32_t i = threadIdx.x + (blockIdx.x % (HEAD_CAP / 256)) * 256, b = blockIdx.x / (HEAD_CAP / 256);
    if (i >= *n_head) return;
    uint64_t lo = hs[i], hi = he[i];
    const uint64_t target = (uint64_t)b << shift;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((uint64_t)post_doc[mid] < target) lo = mid + 1; else hi = mid;
    }
    bounds[(size_t)b * HEAD_CAP + i] = (uint32_t)(lo - hs[i]);
}
// The head ranges that can overlap a chunk [base, base + n): a head list is longer than two chunks, so at most the one that
// is running at `base` and the one that starts inside the chunk.  `cur` is the caller's cursor (first head range that
// ends after `base`; only ever moves forward).  Everything here is uniform over the workgroup.
struct HeadSkip {
    uint64_t a_lo, a_hi, b_lo, b_hi;
    __device__ __forceinline__ bool hit(uint64_t i) const { return (i >= a_lo && i < a_hi) || (i >= b_lo && i < b_hi); }
};
// `k` is the caller's copy of the two ranges, kept across chunks: they only change when `base` passes the end of the running
// head list (a few times per block).  Read afresh for every chunk they were two rounds of dependent global loads in front
// of every chunk — and k_scatter, one workgroup per CU with barriers between its phases, has nothing to run beside them.
// Start with k = {0, 0, 0, 0} (stale: the first call loads).
__device__ __forceinline__ void head_skip(const HeadArgs& h, uint32_t nh, uint32_t& cur, uint64_t base, HeadSkip& k) {
    if (base < k.a_hi) return;                                            // still inside (or in front of) range `cur`
    while (cur < nh && h.he[cur] <= base) cur++;
    k.a_lo = cur < nh ? h.hs[cur] : ~0ull;
    k.a_hi = cur < nh ? h.he[cur] : ~0ull;
    k.b_lo = cur + 1 < nh ? h.hs[cur + 1] : ~0ull;
    k.b_hi = cur + 1 < nh ? h.he[cur + 1] : ~0ull;
}
__device__ __forceinline__ uint32_t head_first(const HeadArgs& h, uint32_t nh, uint64_t r0) {
    uint32_t lo = 0, hi = nh;                                             // first range with he > r0
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (h.he[mid] <= r0) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Pass 1.  Block b owns the contiguous postings [b*per, (b+1)*per) (per is a multiple of the chunk sizes of both passes):
// WEIGHT: w = tf*idf in place; the block's bucket histogram is kept in LDS over its whole range and written ONCE, as column
// b of the [bucket][block] count matrix — no global atomics (the first version added every block's counts of every touched
// bucket to global counters and claimed its output runs the same way: ~150M returning atomics on ~1200 words, most of
// the 9 ms the two passes took).
constexpr int WIN = CH + 2;                   // term window of a chunk: a chunk of CH postings spans at most CH + 1 non-empty terms
template <bool WEIGHT>
__global__ __launch_bounds__(TPB) void k_weight_count(const uint64_t* __restrict__ term_ptr, uint64_t n_terms,
                                                      const uint32_t* __restrict__ post_doc, float* __restrict__ post_w,
                                                      const float* __restrict__ idf, uint64_t n_post, uint64_t per, int shift, uint32_t nb,
                                                      uint32_t nblk, uint32_t* __restrict__ mat, HeadArgs head) {
    __shared__ uint32_t s_hist[NB_MAX];
    __shared__ uint32_t s_tp[WEIGHT ? WIN : 1];   // term_ptr[t0 + k] - base, clamped to [0, 2^32-1]: where the chunk's terms start
    __shared__ uint64_t s_t0;
    __shared__ uint32_t s_need;
    for (uint32_t b = threadIdx.x; b < nb; b += TPB) s_hist[b] = 0;
    const uint64_t r0 = (uint64_t)blockIdx.x * per, r1 = min(n_post, r0 + per);
    if (WEIGHT && threadIdx.x == 0) {
        uint64_t lo = 0, hi = n_terms;                                // largest t with term_ptr[t] <= r0 (once per block)
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (term_ptr[mid] <= r0) lo = mid; else hi = mid;
        }
        s_t0 = lo;
    }
    const uint32_t nh = head.n ? *head.n : 0u;
    uint32_t hcur = nh ? head_first(head, nh, r0) : 0u;
    __syncthreads();
    HeadSkip hk = nh ? HeadSkip{0ull, 0ull, 0ull, 0ull} : HeadSkip{~0ull, ~0ull, ~0ull, ~0ull};
    for (uint64_t base = r0; base < r1; base += CH) {
        const uint32_t n_here = (uint32_t)min((uint64_t)CH, r1 - base);
        if (nh) {
            head_skip(head, nh, hcur, base, hk);
            if (hk.a_lo <= base && hk.a_hi >= base + n_here) {            // the whole chunk belongs to a head list: nothing to do here
                if (WEIGHT) {
                    __syncthreads();
                    if (threadIdx.x == 0) s_t0 = head.term[hcur];         // where the next chunk's terms start
                    __syncthreads();
                }
                continue;
            }
        }
        // every thread takes PER_THREAD CONSECUTIVE postings (vector loads), so it finds the term of its first posting
        // once and walks from there; the chunk's term starts are staged in LDS relative to `base`
        const uint32_t x0 = threadIdx.x * PER_THREAD;
        uint32_t k = 0;
        uint64_t t0 = 0;
        if (WEIGHT) {
            t0 = s_t0;                                                // term of posting `base` or an earlier one
            if (threadIdx.x == 0) s_need = 0;
            __syncthreads();
            // coarse probe: how far do the chunk's terms reach?  thread q looks at term t0 + 1 + 16 q
            {
                const uint64_t t = t0 + 1 + (uint64_t)threadIdx.x * 16;
                const uint64_t v = t <= n_terms ? term_ptr[t] : ~0ull;
                if (v < base + n_here) atomicMax(&s_need, threadIdx.x + 1);
            }
            __syncthreads();
            const uint32_t need = min((uint32_t)WIN, s_need * 16 + 18);   // window entries that can matter
            for (uint32_t q = threadIdx.x; q < need; q += TPB) {
                const uint64_t t = t0 + q;
                const uint64_t v = t <= n_terms ? term_ptr[t] : ~0ull;
                s_tp[q] = v <= base ? 0u : (v - base > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(v - base));
            }
            __syncthreads();
            // largest k with s_tp[k] <= x0 (s_tp[0] = 0); a window that ends before the chunk does (runs of empty terms) is
            // finished from global memory below
            uint32_t lo = 0, hi = need;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_tp[mid] <= x0) lo = mid; else hi = mid;
            }
            k = lo;
        }
        if (x0 < n_here) {
            uint32_t doc[PER_THREAD];
            float w[PER_THREAD];
            const uint64_t i0 = base + x0;
            const bool full = x0 + PER_THREAD <= n_here;
            if (full) {
#pragma unroll
                for (int v4 = 0; v4 < PER_THREAD / 4; v4++) {
                    const uint4 d = *reinterpret_cast<const uint4*>(&post_doc[i0 + v4 * 4]);
                    doc[v4 * 4] = d.x; doc[v4 * 4 + 1] = d.y; doc[v4 * 4 + 2] = d.z; doc[v4 * 4 + 3] = d.w;
                    if (WEIGHT) {
                        const float4 f = *reinterpret_cast<const float4*>(&post_w[i0 + v4 * 4]);
                        w[v4 * 4] = f.x; w[v4 * 4 + 1] = f.y; w[v4 * 4 + 2] = f.z; w[v4 * 4 + 3] = f.w;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < PER_THREAD; j++) {
                    const bool ok = x0 + j < n_here;
                    doc[j] = ok ? post_doc[i0 + j] : 0u;
                    if (WEIGHT) w[j] = ok ? post_w[i0 + j] : 0.f;
                }
            }
            if (WEIGHT) {
                const uint32_t need = min((uint32_t)WIN, s_need * 16 + 18);
                uint64_t t = t0 + k;
#pragma unroll
                for (int j = 0; j < PER_THREAD; j++) {
                    const uint32_t x = x0 + j;
                    if (x >= n_here) continue;
                    if (nh && hk.hit(i0 + j)) continue;                  // head posting: weighted by k_bucket_sum (w[j] goes back as it came)
                    while (k + 1 < need && s_tp[k + 1] <= x) k++;
                    t = t0 + k;
                    if (k + 1 >= need) {                              // past the staged window: rare (long runs of empty terms)
                        uint64_t lo = t, hi = n_terms;
                        const uint64_t i = i0 + j;
                        if (term_ptr[hi] <= i) lo = hi;
                        while (hi - lo > 1) {
                            const uint64_t mid = (lo + hi) >> 1;
                            if (term_ptr[mid] <= i) lo = mid; else hi = mid;
                        }
                        t = lo;
                    }
                    w[j] = w[j] * idf[t];                             // term_weighting.go:42
                }
                if (full) {
#pragma unroll
                    for (int v4 = 0; v4 < PER_THREAD / 4; v4++)
                        *reinterpret_cast<float4*>(&post_w[i0 + v4 * 4]) = make_float4(w[v4 * 4], w[v4 * 4 + 1], w[v4 * 4 + 2], w[v4 * 4 + 3]);
                } else {
#pragma unroll
                    for (int j = 0; j < PER_THREAD; j++)
                        if (x0 + j < n_here) post_w[i0 + j] = w[j];
                }
                // the thread holding the chunk's last posting knows where the next chunk's terms start
                if (x0 + PER_THREAD >= n_here) s_t0 = t;
            }
#pragma unroll
            for (int j = 0; j < PER_THREAD; j++)
                if (x0 + j < n_here && !(nh && hk.hit(i0 + j))) atomicAdd(&s_hist[doc[j] >> shift], 1u);
        }
        if (WEIGHT) __syncthreads();                                  // s_t0 / s_tp are rewritten by the next chunk
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += TPB) mat[(size_t)b * nblk + blockIdx.x] = s_hist[b];
}

// Per bucket (one wave each): exclusive scan of the bucket's row of the count matrix over the blocks, in place; the
// row's total goes to cnt[bucket].
__global__ __launch_bounds__(64) void k_bucket_rowscan(uint32_t* __restrict__ mat, uint32_t nblk, uint32_t* __restrict__ cnt) {
    uint32_t* row = mat + (size_t)blockIdx.x * nblk;
    const uint32_t per = (nblk + 63) / 64, lane = threadIdx.x;
    uint32_t sum = 0;
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t i = lane * per + j;
        if (i < nblk) sum += row[i];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += v;
    }
    uint32_t run = incl - sum;
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t i = lane * per + j;
        if (i < nblk) { const uint32_t c = row[i]; row[i] = run; run += c; }
    }
    if (lane == 63) cnt[blockIdx.x] = incl;
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

// k_scatter: {doc, float32(w*w)} of every posting to its bucket's region.  A wave's 64 postings of a short list
// belong to 64 different buckets: written directly they are 64 partial-line stores per instruction (5.9 ms for the
// 641M-posting table, 1.9x write amplification).  The block therefore sorts its SC_CH records by bucket in LDS first
// (histogram -> exclusive scan -> LDS ticket per record) and then writes them out in LDS order: consecutive lanes write
// consecutive records of a bucket's run.
#ifndef SS_SC_TPB
#define SS_SC_TPB 1024
#endif
constexpr int SC_TPB = SS_SC_TPB;
#ifndef SS_SC_PT
#define SS_SC_PT (8192 / SS_SC_TPB)
#endif
constexpr int SC_PT = SS_SC_PT;                // records per thread
constexpr int SC_CH = SC_TPB * SC_PT;         // records per chunk (staged in LDS)
constexpr int SC_BPT_MAX = NB_MAX / SC_TPB;
constexpr int SC_PF = (1024 + SC_TPB - 1) / SC_TPB;   // entries of the next chunk's term window a thread prefetches (windows up to 1024 terms)
static_assert(SC_CH <= 65535, "staging positions are 16-bit");
// LDS: rec[SC_CH] (8 B: doc, float32 square), then per bucket (nbt = the bucket count rounded up to even): the chunk's count (16 bits,
// two buckets to a word — the returning ds_add of the count is also the record's rank), the first staging position (16 bits) and the
// output position of the chunk's run (32 bits); the block's output cursors live in the registers of the buckets' owner threads and a
// record's bucket is read off its doc id.  10M docs (1222 buckets): 64 + 9.6 KB (before late round 4: 64 + 32 + 16 KB with 32-bit
// tables, a bucket word per record and the cursors in LDS).  ONE workgroup of 1024 threads per CU all the same: the smaller layout
// was made so that two fit, and two measured SLOWER — 2 x 512 threads with 16 postings each 6.6 ms for the build, 2 x 1024 capped at
// 64 VGPRs 6.6, one workgroup of 1024 5.4-5.6 (5.85 with the old layout; 4096-record chunks in two workgroups had measured 5 % slower
// in round 3).  Every resident block keeps a half-written line open per bucket, and twice the blocks are twice the lines an XCD's L2
// has to hold until the block's next chunk completes them (DESIGN K2/K3).
inline size_t scatter_lds_bytes(uint32_t nbt) { return (size_t)SC_CH * 8 + (size_t)nbt * 8 + 64; }
// WEIGHT (round 4: weight and scatter in ONE pass over the postings): w = tf * idf is computed here, written back in place and
// squared on the way into the record — the separate weighting pass read and wrote every tail posting once more (12 of its 36
// bytes; the count pass in front of this kernel now reads doc ids only).  A thread takes SC_PT CONSECUTIVE postings, finds the
// term of its first one in the chunk's term window (term starts relative to the chunk and their idf, staged in the LDS that holds
// the sorted records later in the chunk) and walks on from there.
#ifdef SS_SC_NO_NT
#define SC_NT_LOAD(p) (*(p))
#else
#define SC_NT_LOAD(p) __builtin_nontemporal_load(p)
#endif
#ifdef SS_SC_PHASES
// variant build (tools/build_variant.sh scph -DSS_SC_PHASES): cycles wave 0 of every block spends in each phase of a chunk, summed over
// the grid and printed by the build (s_memtime; the kernel's time is unchanged within 1 %)
__device__ unsigned long long g_sc_ph[8];
#define SC_PH(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); ph_acc[i] += t_ - ph_t; ph_t = t_; } } while (0)
#else
#define SC_PH(i) do { } while (0)
#endif
constexpr int SC_WIN = SC_CH;                  // term-window entries staged per chunk (a chunk spans at most SC_CH + 1 non-empty terms; beyond: global search)
template <bool WEIGHT, int BPT>
#ifndef SS_SC_MINW
#define SS_SC_MINW 1
#endif
__global__ __launch_bounds__(SC_TPB, SS_SC_MINW) void k_scatter(const uint32_t* __restrict__ post_doc, float* __restrict__ post_w, uint64_t n_post,
                                                    uint64_t per, int shift, uint32_t nb, uint32_t nblk, uint32_t bpt, uint32_t nbt,
                                                    const uint32_t* __restrict__ mat, const uint32_t* __restrict__ off, uint2* __restrict__ out,
                                                    HeadArgs head, const uint64_t* __restrict__ term_ptr, uint64_t n_terms, const float* __restrict__ idf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sc_smem[];
    __shared__ uint64_t s_t0;
    __shared__ uint32_t s_need;
    __shared__ uint32_t s_nstaged;
    uint2* const L_rec = reinterpret_cast<uint2*>(sc_smem);
    uint32_t* const s_tp = reinterpret_cast<uint32_t*>(sc_smem);            // [SC_WIN] WEIGHT: the chunk's term starts, relative to `base` ...
    float* const s_idf = reinterpret_cast<float*>(sc_smem) + SC_WIN;         // [SC_WIN] ... and their idf (both in L_rec's bytes: used before the records are staged)
    uint32_t* const L_gout = reinterpret_cast<uint32_t*>(L_rec + SC_CH);   // [nbt] where this chunk's run of the bucket starts in the output MINUS where it starts in the staging area
    uint32_t* const L_hist = L_gout + nbt;                                 // [nbt / 2] records of the chunk per bucket, 16 bits each (bucket b: word b >> 1, half b & 1)
    uint16_t* const L_loff = reinterpret_cast<uint16_t*>(L_hist + nbt / 2);// [nbt] first staging position of the bucket
    uint32_t* const L_part = reinterpret_cast<uint32_t*>(L_loff + nbt);    // [16]
    // thread t owns buckets t * bpt .. t * bpt + bpt - 1 (bpt is even: whole words of L_hist): it scans their counts, keeps their
    // output cursors (next output position of THIS block in the bucket) in registers and zeroes their counts
    uint32_t cur[BPT];
#pragma unroll
    for (int q = 0; q < BPT; q++) {
        const uint32_t b = threadIdx.x * bpt + q;
        cur[q] = ((uint32_t)q < bpt && b < nb) ? off[b] + mat[(size_t)b * nblk + blockIdx.x] : 0u;
    }
    for (uint32_t i = threadIdx.x; i < nbt / 2; i += SC_TPB) L_hist[i] = 0;
    const uint64_t r0 = (uint64_t)blockIdx.x * per, r1 = min(n_post, r0 + per);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (WEIGHT && threadIdx.x == 0) {
        uint64_t lo = 0, hi = n_terms;                                // largest t with term_ptr[t] <= r0 (once per block)
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (term_ptr[mid] <= r0) lo = mid; else hi = mid;
        }
        s_t0 = lo;
    }
    // The next chunk's postings are requested when the current chunk's records are staged (phase 3), into the same registers: the
    // write-out (phase 4) and the window staging run under the loads.  (Round 3/4 loaded them a whole chunk ahead into a second set
    // of registers; the late loads measured no slower and leave the registers to the compiler.)
    uint32_t doc[SC_PT];
    float w[SC_PT];
    uint32_t head_mask = 0u;                                               // postings of doc[] that belong to a head list (see loads / apply_head)
    // head lists are not partitioned (k_bucket_sum reads them in place): their postings count as absent here
    const uint32_t nh = head.n ? *head.n : 0u;
    uint32_t hcur = nh ? head_first(head, nh, r0) : 0u;
    bool n_empty = false;                                                  // the chunk whose head ranges were looked up last holds head postings only (uniform)
    HeadSkip hk = nh ? HeadSkip{0ull, 0ull, 0ull, 0ull} : HeadSkip{~0ull, ~0ull, ~0ull, ~0ull};
    const uint32_t x0 = threadIdx.x * SC_PT;                               // this thread's first posting inside a chunk
    // meta: the head ranges over the chunk at `base` (moves hcur / hk on); loads: its postings, after meta(base)
    auto meta = [&](uint64_t base) __attribute__((always_inline)) {
        n_empty = base >= r1;
        if (nh && base < r1) {
            head_skip(head, nh, hcur, base, hk);
            n_empty = hk.a_lo <= base && hk.a_hi >= min(r1, base + SC_CH);
        }
    };
    auto loads = [&](uint64_t base) __attribute__((always_inline)) {
        const uint64_t i0 = base + x0;
        if (!n_empty && i0 + SC_PT <= r1) {                                // whole group inside the range: vector loads
#pragma unroll
            for (int v4 = 0; v4 < SC_PT / 4; v4++) {
                // streamed once: non-temporal, so that the postings do not push the half-written lines of the 1221 bucket runs this
                // block keeps open out of the XCD's L2 before the next chunk completes them
                typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
                typedef float f4_t __attribute__((ext_vector_type(4)));
                const u4_t d = SC_NT_LOAD(reinterpret_cast<const u4_t*>(&post_doc[i0 + v4 * 4]));
                const f4_t f = SC_NT_LOAD(reinterpret_cast<const f4_t*>(&post_w[i0 + v4 * 4]));
                doc[v4 * 4] = d.x; doc[v4 * 4 + 1] = d.y; doc[v4 * 4 + 2] = d.z; doc[v4 * 4 + 3] = d.w;
                w[v4 * 4] = f.x; w[v4 * 4 + 1] = f.y; w[v4 * 4 + 2] = f.z; w[v4 * 4 + 3] = f.w;
            }
            // Which of these postings belong to a head list is a matter of their POSITIONS: noted here, applied to doc[] where the chunk
            // is first looked at (apply_head), so that nothing in this phase depends on the loads just issued.  [Round 5 suspected the
            // select that used to stand here of making every wave sit out the next chunk's memory latency inside the staging phase —
            // 32-35 % of a chunk's cycles in the phase clocks.  Moving it changed nothing (5.13 ms either way): the phase is long because
            // its barrier waits for the slowest of sixteen waves on a shared LDS pipe, not because of a stalled load.]
            head_mask = 0u;
#pragma unroll
            for (int j = 0; j < SC_PT; j++)
                if (nh && hk.hit(i0 + j)) head_mask |= 1u << j;
        } else {
            head_mask = 0u;
#pragma unroll
            for (int j = 0; j < SC_PT; j++) {
                const uint64_t i = i0 + j;
                const bool ok = !n_empty && i < r1 && !(nh && hk.hit(i));
                doc[j] = ok ? post_doc[i] : 0xFFFFFFFFu;
                w[j] = ok ? post_w[i] : 0.f;
            }
        }
    };
    auto apply_head = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < SC_PT; j++)
            if ((head_mask >> j) & 1u) doc[j] = 0xFFFFFFFFu;
        head_mask = 0u;
    };
    meta(r0);
    loads(r0);
    bool cur_empty = n_empty;
    __syncthreads();
    // WEIGHT: the term window of the NEXT chunk is fetched while the current chunk goes through its LDS phases (two dependent loads —
    // how far the terms reach, then their starts and idf — that cost every chunk ~4k cycles in front of the weight phase when they
    // were made there).  One entry per thread: windows above SC_TPB terms (lists of under eight postings on average) are staged
    // the slow way.  pf_ok: pf_tp / pf_idf hold the window of the chunk that starts at `base`, pf_need entries, first term pf_t0.
    bool pf_ok = false;
    uint32_t pf_need = 0;
    uint64_t pf_t0 = 0, pf_tp[SC_PF] = {}, pf_probe = 0;
    float pf_idf[SC_PF] = {};
    __shared__ uint32_t s_need_n;
#ifdef SS_SC_PHASES
    unsigned long long ph_acc[8] = {}, ph_t = __builtin_readcyclecounter();
#endif
    for (uint64_t base = r0; base < r1; base += SC_CH) {
        uint32_t rank2[SC_PT / 2];                                        // two 16-bit ranks to a register
        SC_PH(7);
        const bool empty = cur_empty;
        apply_head();                                                      // (the chunk's postings have had the previous write-out to arrive)
        const uint32_t h_here = hcur;                                      // the head range running at `base` (before meta moves on)
        meta(base + SC_CH);
        cur_empty = n_empty;                                               // ... of the next chunk, for the next turn
        const bool next_live = WEIGHT && !n_empty && base + SC_CH < r1;     // the next chunk will want a window
        if (empty) {
            if (WEIGHT) {                                                  // a chunk inside a head list: the next chunk's terms start at that list
                __syncthreads();
                if (threadIdx.x == 0) s_t0 = head.term[h_here];
                __syncthreads();
                pf_ok = false;
            }
            loads(base + SC_CH);
            continue;
        }
#if defined(SS_EXP_SC) && SS_EXP_SC == 3      // timing experiments only (wrong results): loads alone
        if (doc[0] == 0x12345678u) out[0] = make_uint2(0u, __float_as_uint(w[0]));
        loads(base + SC_CH);
        continue;
#endif
        if (WEIGHT) {
            // (0) the chunk's term window, then w = tf * idf (term_weighting.go:42) for this thread's postings, written back in place
            const uint32_t n_here = (uint32_t)min((uint64_t)SC_CH, r1 - base);
            __syncthreads();                                               // the previous chunk's records (same bytes) have been written out
            SC_PH(0);
            uint64_t t0;
            uint32_t need;
            if (pf_ok) {
                t0 = pf_t0;
                need = pf_need;
#pragma unroll
                for (int u = 0; u < SC_PF; u++) {
                    const uint32_t e = threadIdx.x + (uint32_t)u * SC_TPB;
                    if (e < need) {
                        s_tp[e] = pf_tp[u] <= base ? 0u : (pf_tp[u] - base > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(pf_tp[u] - base));
                        s_idf[e] = pf_idf[u];
                    }
                }
                if (threadIdx.x == 0) s_need_n = 0;
            } else {
                t0 = s_t0;                                                 // term of posting `base` or an earlier one
                if (threadIdx.x == 0) { s_need = 0; s_need_n = 0; }
                __syncthreads();
                {                                                          // coarse probe: how far do the chunk's terms reach?
                    const uint64_t t = t0 + 1 + (uint64_t)threadIdx.x * SC_PT;
                    const uint64_t v = t <= n_terms ? term_ptr[t] : ~0ull;
                    if (v < base + n_here) atomicMax(&s_need, threadIdx.x + 1);
                }
                __syncthreads();
                need = min((uint32_t)SC_WIN, s_need * SC_PT + SC_PT + 2);
                for (uint32_t q = threadIdx.x; q < need; q += SC_TPB) {
                    const uint64_t t = t0 + q;
                    const uint64_t v = t <= n_terms ? term_ptr[t] : ~0ull;
                    s_tp[q] = v <= base ? 0u : (v - base > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(v - base));
                    s_idf[q] = t < n_terms ? idf[t] : 0.f;
                }
            }
            __syncthreads();
            SC_PH(1);
            if (x0 < n_here) {
                uint32_t k = 0;
                {
                    uint32_t lo = 0, hi = need;                            // largest k with s_tp[k] <= x0 (s_tp[0] = 0)
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_tp[mid] <= x0) lo = mid; else hi = mid;
                    }
                    k = lo;
                }
                uint64_t t = t0 + k;
                bool any = false;
#pragma unroll
                for (int j = 0; j < SC_PT; j++) {
                    const uint32_t x = x0 + j;
                    if (x >= n_here || doc[j] == 0xFFFFFFFFu) continue;    // past the range, or a head posting (weighted by k_bucket_sum)
                    while (k + 1 < need && s_tp[k + 1] <= x) k++;
                    t = t0 + k;
                    float f = s_idf[k];
                    if (k + 1 >= need) {                                   // past the staged window: rare (long runs of empty terms)
                        uint64_t lo = t, hi = n_terms;
                        const uint64_t i = base + x;
                        if (term_ptr[hi] <= i) lo = hi;
                        while (hi - lo > 1) {
                            const uint64_t mid = (lo + hi) >> 1;
                            if (term_ptr[mid] <= i) lo = mid; else hi = mid;
                        }
                        t = lo;
                        f = idf[t];
                    }
                    w[j] = w[j] * f;                                       // term_weighting.go:42
                    any = true;
                }
                const uint64_t i0 = base + x0;
                bool all = x0 + SC_PT <= n_here;
#pragma unroll
                for (int j = 0; j < SC_PT; j++) all = all && doc[j] != 0xFFFFFFFFu;
                if (all) {
#pragma unroll
                    for (int v4 = 0; v4 < SC_PT / 4; v4++)
                    {
#ifdef SS_SC_W_PLAIN_STORE
                        *reinterpret_cast<float4*>(&post_w[i0 + v4 * 4]) = make_float4(w[v4 * 4], w[v4 * 4 + 1], w[v4 * 4 + 2], w[v4 * 4 + 3]);
#else
                        // (streamed out, never read again here: non-temporal, so that the weights do not push the half-written lines of
                        //  the block's bucket runs out of the L2)
                        typedef float f4s_t __attribute__((ext_vector_type(4)));
                        f4s_t wv; wv.x = w[v4 * 4]; wv.y = w[v4 * 4 + 1]; wv.z = w[v4 * 4 + 2]; wv.w = w[v4 * 4 + 3];
                        __builtin_nontemporal_store(wv, reinterpret_cast<f4s_t*>(&post_w[i0 + v4 * 4]));
#endif
                    }
                } else if (any) {
#pragma unroll
                    for (int j = 0; j < SC_PT; j++)
                        if (x0 + j < n_here && doc[j] != 0xFFFFFFFFu) post_w[i0 + j] = w[j];
                }
                // the thread that holds the chunk's last posting knows where the next chunk's terms start (a lower bound is enough)
                if (x0 + SC_PT >= n_here) s_t0 = t;
            }
            __syncthreads();                                               // the window's bytes become the record staging area
            SC_PH(2);
            // the next chunk's window, first load: how far do its terms reach? (s_t0 now names its first term)
            pf_ok = false;
            if (next_live) {
                pf_t0 = s_t0;
                const uint64_t t = pf_t0 + 1 + (uint64_t)threadIdx.x * SC_PT;
                pf_probe = t <= n_terms ? term_ptr[t] : ~0ull;
            }
        }
        // (1) count; the returned value is the record's rank inside its bucket
#pragma unroll
        for (int j = 0; j < SC_PT; j++) {
            const uint32_t b = doc[j] >> shift, sh = (b & 1u) << 4;
            const uint32_t r = doc[j] != 0xFFFFFFFFu ? (atomicAdd(&L_hist[b >> 1], 1u << sh) >> sh) & 0xFFFFu : 0u;
            rank2[j >> 1] = (j & 1) ? (rank2[j >> 1] | (r << 16)) : r;
        }
        if (WEIGHT && next_live) {
            const uint64_t base_n = base + SC_CH;
            if (pf_probe < base_n + min((uint64_t)SC_CH, r1 - base_n)) atomicMax(&s_need_n, threadIdx.x + 1);
        }
        __syncthreads();
        SC_PH(3);
        // (2) exclusive scan of the counts -> staging offsets; claim this chunk's runs from the block's cursors
        uint32_t c[BPT], run = 0;
#pragma unroll
        for (int q = 0; q < BPT; q++) {
            const uint32_t b = threadIdx.x * bpt + q;
            c[q] = ((uint32_t)q < bpt && b < nbt) ? (L_hist[b >> 1] >> ((b & 1u) << 4)) & 0xFFFFu : 0u;
            run += c[q];
        }
        uint32_t incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) L_part[wv] = incl;
        __syncthreads();
        uint32_t o = incl - run;
        {
            // the earlier waves' totals: all SC_TPB / 64 words with wide loads issued together instead of a loop of `wv` dependent
            // 4-byte reads (up to 15 LDS latencies for the last wave; no measurable change in the build time)
            constexpr int NW4 = (SC_TPB / 64 + 3) / 4;
            uint4 pw[NW4];
#pragma unroll
            for (int q4 = 0; q4 < NW4; q4++) pw[q4] = reinterpret_cast<const uint4*>(L_part)[q4];
#pragma unroll
            for (int q4 = 0; q4 < NW4; q4++) {
                o += (4 * q4 + 0 < wv ? pw[q4].x : 0u) + (4 * q4 + 1 < wv ? pw[q4].y : 0u) + (4 * q4 + 2 < wv ? pw[q4].z : 0u) + (4 * q4 + 3 < wv ? pw[q4].w : 0u);
            }
        }
#pragma unroll
        for (int q = 0; q < BPT; q++) {
            const uint32_t b = threadIdx.x * bpt + q;
            if ((uint32_t)q < bpt && b < nbt) {
                L_loff[b] = (uint16_t)o;
                L_gout[b] = cur[q] - o;                                    // (output position of a staged record = this + its staging position)
                o += c[q];
                cur[q] += c[q];
                if ((q & 1) == 0) L_hist[b >> 1] = 0;                      // (bpt and nbt are even: the word is this thread's alone)
            }
        }
        if (threadIdx.x == SC_TPB - 1) s_nstaged = o;                      // (the last thread's running offset ends behind the last bucket)
        if (WEIGHT && next_live) {
            // ... second load: the window itself (s_need_n is complete since the barrier behind the count phase)
            pf_need = s_need_n * SC_PT + SC_PT + 2;
            pf_ok = pf_need <= (uint32_t)(SC_PF * SC_TPB);
            if (pf_ok) {
#pragma unroll
                for (int u = 0; u < SC_PF; u++) {
                    const uint32_t e = threadIdx.x + (uint32_t)u * SC_TPB;
                    if (e < pf_need) {
                        const uint64_t t = pf_t0 + e;
                        pf_tp[u] = t <= n_terms ? term_ptr[t] : ~0ull;
                        pf_idf[u] = t < n_terms ? idf[t] : 0.f;
                    }
                }
            }
        }
        __syncthreads();
        SC_PH(4);
#if defined(SS_EXP_SC) && SS_EXP_SC == 2      // ... loads, count and scan
        loads(base + SC_CH);
        continue;
#endif
        // (3) records to their staging positions
#pragma unroll
        for (int j = 0; j < SC_PT; j++) {
            if (doc[j] != 0xFFFFFFFFu) {
                const uint32_t b = doc[j] >> shift;
                const float sq = w[j] * w[j];                          // term_weighting.go:44 (float32 product)
                const uint32_t pos = (uint32_t)L_loff[b] + ((rank2[j >> 1] >> ((j & 1) << 4)) & 0xFFFFu);
                L_rec[pos] = make_uint2(doc[j], __float_as_uint(sq));
            }
        }
        loads(base + SC_CH);                                               // (doc / w are free: the next chunk's postings travel under phase 4)
        __syncthreads();
        SC_PH(5);
        // (4) out, in staging order: consecutive lanes write consecutive records of a bucket's run
        // (the staged records end where the last bucket's run ends; head postings were never staged)
        const uint32_t n_staged = s_nstaged;
#if defined(SS_EXP_SC) && SS_EXP_SC == 1      // ... everything but the stores
        if (n_staged == 0x12345678u) out[0] = L_rec[0];
        continue;
#endif
#if defined(SS_EXP_SC) && SS_EXP_SC == 4      // ... the stores as one coalesced stream (what perfectly joined bucket runs would cost)
        for (uint32_t pos = threadIdx.x; pos < n_staged; pos += SC_TPB) out[base + pos] = L_rec[pos];
        continue;
#endif
        for (uint32_t pos = threadIdx.x; pos < n_staged; pos += SC_TPB) {
            const uint2 r = L_rec[pos];
            const uint32_t b = r.x >> shift;
            out[(uint64_t)(L_gout[b] + pos)] = r;                          // (L_gout = run start - staging start, modulo 2^32: the sum is exact)
        }
        // the next chunk's phase (1) only touches hist; its phase (2) rewrites loff / gout after the barrier that follows (1),
        // by which time every thread has left (4)  [WEIGHT: the barrier at the top of the weight phase comes before the window is staged]
        SC_PH(6);
    }
#ifdef SS_SC_PHASES
    if (threadIdx.x == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&g_sc_ph[i], ph_acc[i]);
#endif
}

// magnitudes only (ss_index_refresh_magnitudes): small tables
__global__ void k_sumsq_atomic(const uint32_t* __restrict__ post_doc, const float* __restrict__ post_w, uint64_t n_post, double* __restrict__ mag2) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_post; i += (uint64_t)gridDim.x * blockDim.x) {
        const float w = post_w[i];
        const float sq = w * w;                                       // term_weighting.go:44 (float32 product)
        unsafeAtomicAdd(&mag2[post_doc[i]], (double)sq);
    }
}

#ifndef SS_TPB_B
#define SS_TPB_B 1024
#endif
constexpr int TPB_B = SS_TPB_B;
#ifndef SS_HEAD_PIECE
#define SS_HEAD_PIECE 2048
#endif
#ifndef SS_HEAD_U
#define SS_HEAD_U 16
#endif
#ifndef SS_BS_U
#define SS_BS_U 8
#endif
constexpr int BS_U = SS_BS_U;                // partitioned records of a thread in flight
constexpr int HEAD_PIECE = SS_HEAD_PIECE;    // head postings a wave takes at a time
constexpr int HEAD_U = SS_HEAD_U;            // loads of a lane in flight per array (4: 1.19 ms for the head part of the 641M table, 8: see DESIGN)
inline size_t bucket_lds_bytes(int shift) { return ((size_t)1 << shift) * 8 + (size_t)(3 * HEAD_CAP + 4) * 4; }
// One workgroup per bucket: float64 accumulators in LDS take (A) the bucket's partitioned records and (B) the bucket's run of
// every head list, read in place — WEIGHT: tf in, w = tf*idf out (term_weighting.go:42), otherwise the weights as they stand.
template <bool WEIGHT>
__global__ __launch_bounds__(TPB_B) void k_bucket_sum(const uint2* __restrict__ packed, const uint32_t* __restrict__ off, uint64_t n_docs,
                                                      int shift, double* __restrict__ mag, double* __restrict__ mag2,
                                                      const uint32_t* __restrict__ post_doc, float* __restrict__ post_w,
                                                      const float* __restrict__ idf, HeadArgs head) {
    extern __shared__ double acc[];                                    // [1 << shift], then the head run table
    __shared__ uint32_t s_wsum[TPB_B / 64];
    const uint32_t bd = 1u << shift, b = blockIdx.x;
    uint32_t* const pre = reinterpret_cast<uint32_t*>(acc + bd);       // [HEAD_CAP + 1] exclusive prefix of the run lengths
    uint32_t* const r_pos = pre + HEAD_CAP + 4;                        // [HEAD_CAP] first posting of the run (the bucketed pass has P < 2^32)
    float* const r_idf = reinterpret_cast<float*>(r_pos + HEAD_CAP);   // [HEAD_CAP] the list's idf
    for (uint32_t i = threadIdx.x; i < bd; i += TPB_B) acc[i] = 0.0;
    __syncthreads();
    const uint32_t lo = off[b], hi = off[b + 1];
    // (A) BS_U records per thread in flight (two workgroups of 1024 per CU: 512 threads each measured 6.8 ms for the build, 1024 6.3)
    for (uint32_t i0 = lo; i0 < hi; i0 += BS_U * TPB_B) {
        uint2 r[BS_U];
#pragma unroll
        for (int u = 0; u < BS_U; u++) {
            const uint32_t i = i0 + u * TPB_B + threadIdx.x;
            r[u] = i < hi ? packed[i] : make_uint2(0u, 0u);            // a zero square adds nothing
        }
#pragma unroll
        for (int u = 0; u < BS_U; u++)
            if (i0 + u * TPB_B + threadIdx.x < hi) atomicAdd(&acc[r[u].x & (bd - 1)], (double)__uint_as_float(r[u].y));   // :44 (float64 accumulate; LDS)
    }
    // (B) head lists
    const uint32_t nh = head.n ? *head.n : 0u;
    if (nh) {
        const uint32_t* const b0 = head.bounds + (size_t)b * HEAD_CAP;
        const uint32_t* const b1 = b0 + HEAD_CAP;
        // exclusive prefix of the run lengths: HEAD_CAP / TPB_B consecutive lists per thread
        constexpr int PT = HEAD_CAP / TPB_B;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        uint32_t len[PT], run = 0;
#pragma unroll
        for (int q = 0; q < PT; q++) {
            const uint32_t i = threadIdx.x * PT + q;
            const uint32_t lo_i = i < nh ? b0[i] : 0u;
            len[q] = i < nh ? b1[i] - lo_i : 0u;
            run += len[q];
            if (i < nh) {
                r_pos[i] = (uint32_t)(head.hs[i] + lo_i);
                if (WEIGHT) r_idf[i] = idf[head.term[i]];
            }
        }
        uint32_t incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) s_wsum[wv] = incl;
        __syncthreads();
        uint32_t o = incl - run;
        for (int q = 0; q < wv; q++) o += s_wsum[q];
#pragma unroll
        for (int q = 0; q < PT; q++) {
            pre[threadIdx.x * PT + q] = o;
            o += len[q];
        }
        if (threadIdx.x == TPB_B - 1) pre[HEAD_CAP] = o;
        __syncthreads();
        const uint32_t total = pre[HEAD_CAP];
        // the runs laid end to end are cut into pieces of HEAD_PIECE postings; wave w takes pieces w, w + 8, ...
        for (uint32_t v0 = (uint32_t)wv * HEAD_PIECE; v0 < total; v0 += (TPB_B / 64) * HEAD_PIECE) {
            const uint32_t v1 = min(total, v0 + (uint32_t)HEAD_PIECE);
            uint32_t rl = 0, rh = nh;                                  // largest r with pre[r] <= v0
            while (rh - rl > 1) {
                const uint32_t mid = (rl + rh) >> 1;
                if (pre[mid] <= v0) rl = mid; else rh = mid;
            }
            uint32_t v = v0;
            for (uint32_t r = rl; r < nh && v < v1; r++) {
                const uint32_t r_end = pre[r + 1];
                if (r_end <= v) continue;                              // empty run
                const uint32_t n = min(r_end, v1) - v;
                const uint64_t s0 = (uint64_t)r_pos[r] + (v - pre[r]);
                const float f = WEIGHT ? r_idf[r] : 1.f;
                for (uint32_t j0 = 0; j0 < n; j0 += HEAD_U * 64) {
                    uint32_t d[HEAD_U];
                    float x[HEAD_U];
#pragma unroll
                    for (int u = 0; u < HEAD_U; u++) {
                        const uint32_t j = j0 + u * 64 + lane;
                        d[u] = j < n ? post_doc[s0 + j] : 0u;
                        x[u] = j < n ? post_w[s0 + j] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < HEAD_U; u++) {
                        const uint32_t j = j0 + u * 64 + lane;
                        if (j < n) {
                            float w = x[u];
                            if (WEIGHT) {
                                w = w * f;                             // term_weighting.go:42
                                post_w[s0 + j] = w;
                            }
                            const float sq = w * w;                    // :44 (float32 product)
                            atomicAdd(&acc[d[u] & (bd - 1)], (double)sq);
                        }
                    }
                }
                v += n;
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bd; i += TPB_B) {
        const uint64_t d = (uint64_t)b * bd + i;
        if (d < n_docs) { mag[d] = sqrt(acc[i]); mag2[d] = acc[i]; }   // term_weighting.go:72
    }
}

__global__ void k_sqrt(double* __restrict__ v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = sqrt(v[i]);                                     // term_weighting.go:72
}


// ---- bucketed magnitude pass: buffers (allocated before the clock starts) and launches ------------------------
struct BucketPass {
    ss::DevBuf<uint32_t> mat, cnt, off, cur;
    ss::DevBuf<uint2> packed;
    ss::DevBuf<uint32_t> h_n, h_term, h_bounds, h_hist;  // head lists (see HeadArgs)
    ss::DevBuf<uint64_t> h_hs, h_he;
    uint64_t head_thr = 0;                       // shortest list that takes the head path; 0 = no head path
    uint32_t nb = 0, nblk = 0, bpt = 2, nbt = 2;
    uint64_t per = 0;
    int shift = 13;
};
int32_t bucket_pass_prepare(ss_ctx* ctx, uint64_t P, uint32_t nb, int shift, BucketPass& bp) {
    bp.nb = nb;
    bp.shift = shift;
    // blocks own contiguous ranges whose length is a multiple of both passes' chunk sizes; ~1024 ranges
    const uint64_t unit = std::max<uint64_t>(SC_CH, CH);
    static_assert(SC_CH % CH == 0 || CH % SC_CH == 0, "a range must be whole chunks of both passes");
    const uint64_t target_blocks = (uint64_t)std::max<int64_t>(1, ctx->opt("tfidf.blocks", 4096));
    bp.per = std::max<uint64_t>(unit, ss::div_up(ss::div_up(P, target_blocks), unit) * unit);
    bp.nblk = (uint32_t)ss::div_up(P, bp.per);
    SS_HIP(ctx, bp.mat.alloc((size_t)nb * bp.nblk));
    SS_HIP(ctx, bp.cnt.alloc(nb));
    SS_HIP(ctx, bp.off.alloc(nb + 1));
    SS_HIP(ctx, bp.cur.alloc(nb));
    SS_HIP(ctx, bp.packed.alloc(P));
    // head lists: average run of at least "tfidf.head_min_run" postings per bucket (0 switches the head path off), and longer
    // than two chunks of either pass (head_skip relies on it)
    const int64_t min_run = ctx->opt("tfidf.head_min_run", 64);
    bp.head_thr = min_run > 0 ? std::max<uint64_t>((uint64_t)min_run * nb, 2 * unit + 1) : 0;
    if (bp.head_thr) {
        SS_HIP(ctx, bp.h_n.alloc(1));
        SS_HIP(ctx, bp.h_hist.alloc(HEAD_LEVELS));
        SS_HIP(ctx, bp.h_term.alloc(HEAD_CAP));
        SS_HIP(ctx, bp.h_hs.alloc(HEAD_CAP));
        SS_HIP(ctx, bp.h_he.alloc(HEAD_CAP));
        SS_HIP(ctx, bp.h_bounds.alloc((size_t)(nb + 1) * HEAD_CAP));
    }
    bp.bpt = (ss::div_up(nb, (uint32_t)SC_TPB) + 1u) & ~1u;                  // even: a thread's buckets are whole words of the packed counts
    bp.nbt = (nb + 1u) & ~1u;
    if (bp.bpt > (uint32_t)SC_BPT_MAX) return ctx->fail(SS_ERR_UNSUPPORTED, "tfidf: %u buckets exceed the partition's %d", nb, NB_MAX);
    if (ctx->tfidf_scatter_lds < (int)scatter_lds_bytes(bp.nbt)) {
        const int lds = (int)scatter_lds_bytes(bp.nbt);
        SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter<true, SC_BPT_MAX>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_scatter<false, SC_BPT_MAX>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        ctx->tfidf_scatter_lds = lds;
    }
    if (ctx->tfidf_bucket_lds < (int)bucket_lds_bytes(shift)) {
        SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket_sum<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bucket_lds_bytes(shift)));
        SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_bucket_sum<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bucket_lds_bytes(shift)));
        ctx->tfidf_bucket_lds = (int)bucket_lds_bytes(shift);
    }
    return SS_OK;
}
// weight: also w = tf*idf in place (ss_tfidf_build); otherwise the magnitudes of the weights as they stand
void bucket_pass_launch(ss_index* idx, hipStream_t st, BucketPass& bp, bool weight, const float* idf) {
    const uint64_t P = idx->n_post, N = idx->n_docs, T = idx->n_terms;
    HeadArgs head{nullptr, nullptr, nullptr, nullptr, nullptr};
    if (bp.head_thr && T) {
        (void)hipMemsetAsync(bp.h_n.p, 0, sizeof(uint32_t), st);
        (void)hipMemsetAsync(bp.h_hist.p, 0, HEAD_LEVELS * sizeof(uint32_t), st);
        hipLaunchKernelGGL(k_head_hist, dim3(ss::div_up(T, TPB)), dim3(TPB), 0, st, idx->term_ptr.p, T, bp.head_thr, bp.h_hist.p);
        hipLaunchKernelGGL(k_head_select, dim3(ss::div_up(T, TPB)), dim3(TPB), 0, st, idx->term_ptr.p, T, bp.head_thr, (const uint32_t*)bp.h_hist.p,
                           bp.h_n.p, bp.h_term.p);
        hipLaunchKernelGGL(k_head_sort, dim3(1), dim3(1024), 0, st, idx->term_ptr.p, bp.h_n.p, bp.h_term.p, bp.h_hs.p, bp.h_he.p);
        hipLaunchKernelGGL(k_head_bounds, dim3((bp.nb + 1) * (HEAD_CAP / 256)), dim3(256), 0, st, (const uint32_t*)idx->post_doc.p, (const uint32_t*)bp.h_n.p,
                           (const uint64_t*)bp.h_hs.p, (const uint64_t*)bp.h_he.p, bp.shift, bp.nb, bp.h_bounds.p);
        head = HeadArgs{bp.h_n.p, bp.h_term.p, bp.h_hs.p, bp.h_he.p, bp.h_bounds.p};
    }
    // the count pass reads doc ids only; the weights are multiplied by k_scatter<true> on its way through the postings (option
    // "tfidf.fused" = 0: the round-3 order — weight + count, then a scatter that reads the weighted postings again)
    const bool fused = weight && idx->ctx->opt("tfidf.fused", 1) != 0;
    if (weight && !fused) hipLaunchKernelGGL(k_weight_count<true>, dim3(bp.nblk), dim3(TPB), 0, st, idx->term_ptr.p, T, idx->post_doc.p, idx->post_w.p,
                                             idf, P, bp.per, bp.shift, bp.nb, bp.nblk, bp.mat.p, head);
    else hipLaunchKernelGGL(k_weight_count<false>, dim3(bp.nblk), dim3(TPB), 0, st, idx->term_ptr.p, T, idx->post_doc.p, idx->post_w.p,
                            idf, P, bp.per, bp.shift, bp.nb, bp.nblk, bp.mat.p, head);
    hipLaunchKernelGGL(k_bucket_rowscan, dim3(bp.nb), dim3(64), 0, st, bp.mat.p, bp.nblk, bp.cnt.p);
    hipLaunchKernelGGL(k_bucket_offsets, dim3(1), dim3(1024), 0, st, bp.cnt.p, bp.nb, bp.off.p, bp.cur.p);
    // (the owner threads' cursor registers are sized by the buckets per thread: 2 up to 1024 buckets, 4 up to 2048, 8 beyond)
#define SS_SCATTER_LAUNCH(W, B)                                                                                                                     \
    hipLaunchKernelGGL((k_scatter<W, B>), dim3(bp.nblk), dim3(SC_TPB), scatter_lds_bytes(bp.nbt), st, (const uint32_t*)idx->post_doc.p, idx->post_w.p, P, \
                       bp.per, bp.shift, bp.nb, bp.nblk, bp.bpt, bp.nbt, (const uint32_t*)bp.mat.p, (const uint32_t*)bp.off.p, bp.packed.p, head,   \
                       (const uint64_t*)idx->term_ptr.p, T, idf)
    if (fused) {
        if (bp.bpt <= 2) SS_SCATTER_LAUNCH(true, 2); else if (bp.bpt <= 4) SS_SCATTER_LAUNCH(true, 4); else SS_SCATTER_LAUNCH(true, SC_BPT_MAX);
    } else {
        if (bp.bpt <= 2) SS_SCATTER_LAUNCH(false, 2); else if (bp.bpt <= 4) SS_SCATTER_LAUNCH(false, 4); else SS_SCATTER_LAUNCH(false, SC_BPT_MAX);
    }
#undef SS_SCATTER_LAUNCH
#ifdef SS_SC_PHASES
    {
        unsigned long long h[8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sc_ph), sizeof(h));
        unsigned long long tot = 0;
        for (int i = 0; i < 8; i++) tot += h[i];
        fprintf(stderr, "[k_scatter phases, %u blocks, cycles of wave 0 summed] wait-for-previous-write-out+barrier %.1f%%  window staged %.1f%%  weights %.1f%%  count %.1f%%  scan %.1f%%  "
                        "stage+next loads issued %.1f%%  write-out %.1f%%  loop top %.1f%%  (total %.3g cycles = %.1f k per block)\n", bp.nblk,
                100.0 * h[0] / tot, 100.0 * h[1] / tot, 100.0 * h[2] / tot, 100.0 * h[3] / tot, 100.0 * h[4] / tot, 100.0 * h[5] / tot, 100.0 * h[6] / tot, 100.0 * h[7] / tot,
                (double)tot, (double)tot / bp.nblk / 1e3);
        unsigned long long z[8] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sc_ph), z, sizeof(z));
    }
#endif
    if (weight) hipLaunchKernelGGL(k_bucket_sum<true>, dim3(bp.nb), dim3(TPB_B), bucket_lds_bytes(bp.shift), st, bp.packed.p, bp.off.p, N, bp.shift,
                                   idx->mag.p, idx->mag2.p, (const uint32_t*)idx->post_doc.p, idx->post_w.p, idf, head);
    else hipLaunchKernelGGL(k_bucket_sum<false>, dim3(bp.nb), dim3(TPB_B), bucket_lds_bytes(bp.shift), st, bp.packed.p, bp.off.p, N, bp.shift,
                            idx->mag.p, idx->mag2.p, (const uint32_t*)idx->post_doc.p, idx->post_w.p, idf, head);
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
    SS_HIP(ctx, idx->mag2.alloc(n_docs));
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
    SS_HIP(ctx, ss::fetch(ctx, st, h_cnt, d_cnt.p, sizeof(h_cnt), &h_err, d_err.p, sizeof(h_err)));
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
    shift = (int)std::max<int64_t>(10, std::min<int64_t>(14, ctx->opt("tfidf.bucket_shift", shift)));
    if ((N >> shift) >= (uint64_t)NB_MAX) shift = 14;
    const uint64_t nb64 = ss::div_up(std::max<uint64_t>(N, 1), (uint64_t)1 << shift);
    // "tfidf.bucket_min" (tests, A/B): smallest table that takes the bucketed pass; a huge value forces the atomics
    const uint64_t min_p = (uint64_t)ctx->opt("tfidf.bucket_min", (int64_t)1 << 22);
    const bool bucketed = P >= std::max<uint64_t>(min_p, 1) && P < ((uint64_t)1 << 32) && nb64 <= (uint64_t)NB_MAX;
    const uint32_t nb = (uint32_t)nb64;
    BucketPass bp;
    if (bucketed) SS_TRY(bucket_pass_prepare(ctx, P, nb, shift, bp));
    SS_HIP(ctx, hipStreamSynchronize(st));       // allocator work (and anything queued before) is over when the clock starts
    SS_HIP(ctx, hipEventRecord(ctx->ev[2][0], st));
    SS_HIP(ctx, hipMemsetAsync(idx->mag.p, 0, N * sizeof(double), st));
    if (T) hipLaunchKernelGGL(k_idf, dim3(ss::div_up(T, TPB)), dim3(TPB), 0, st, idx->term_ptr.p,
                              idx->has_df_global ? idx->df_global.p : nullptr, T, (double)total_docs, idf.p);
    if (bucketed) {
        bucket_pass_launch(idx, st, bp, true, idf.p);
    } else {
        if (P) hipLaunchKernelGGL(k_weight, dim3(ss::div_up(P, CH)), dim3(TPB), 0, st, idx->term_ptr.p, T, idx->post_doc.p,
                                  idx->post_w.p, idf.p, P, idx->mag.p);
        SS_HIP(ctx, hipMemcpyAsync(idx->mag2.p, idx->mag.p, N * sizeof(double), hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_sqrt, dim3(ss::div_up(N, TPB)), dim3(TPB), 0, st, idx->mag.p, N);
    }
    idx->mag2_valid = true;
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
    const uint64_t min_p = (uint64_t)ctx->opt("tfidf.bucket_min", (int64_t)1 << 22);     // tests, A/B (as in ss_tfidf_build)
    const bool bucketed = P >= std::max<uint64_t>(min_p, 1) && P < ((uint64_t)1 << 32) && nb64 <= (uint64_t)NB_MAX;
    SS_HIP(ctx, hipMemsetAsync(idx->mag.p, 0, N * sizeof(double), st));
    if (bucketed) {
        BucketPass bp;
        SS_TRY(bucket_pass_prepare(ctx, P, (uint32_t)nb64, shift, bp));
        bucket_pass_launch(idx, st, bp, false, nullptr);
        SS_HIP(ctx, hipGetLastError());
        SS_HIP(ctx, hipStreamSynchronize(st));
    } else {
        if (P) hipLaunchKernelGGL(k_sumsq_atomic, dim3(std::min<unsigned>(ss::div_up(P, TPB), 16384u)), dim3(TPB), 0, st, (const uint32_t*)idx->post_doc.p,
                                  (const float*)idx->post_w.p, P, idx->mag.p);
        SS_HIP(ctx, hipMemcpyAsync(idx->mag2.p, idx->mag.p, N * sizeof(double), hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_sqrt, dim3(ss::div_up(N, TPB)), dim3(TPB), 0, st, idx->mag.p, N);
        SS_HIP(ctx, hipGetLastError());
    }
    idx->mag2_valid = true;
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

int32_t ss_index_read_positions(ss_index* idx, uint64_t* pos_ptr_out, float* pos_out) {
    if (!idx) return SS_ERR_INVALID;
    ss_ctx* ctx = idx->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!idx->pos_ptr.p) return ctx->fail(SS_ERR_STATE, "ss_index_read_positions: the table holds no positional postings");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (pos_ptr_out) SS_HIP(ctx, hipMemcpyAsync(pos_ptr_out, idx->pos_ptr.p, (idx->n_post + 1) * sizeof(uint64_t), hipMemcpyDefault, st));
    if (pos_out && idx->pos.n) SS_HIP(ctx, hipMemcpyAsync(pos_out, idx->pos.p, idx->pos.n * sizeof(float), hipMemcpyDefault, st));
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
    SS_HIP(ctx, ss::fetch(ctx, st, &h_err, err.p, sizeof(uint32_t)));
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
    SS_HIP(ctx, ss::fetch(ctx, st, &ends[0], idx->pos_ptr.p, sizeof(uint64_t), &ends[1], idx->pos_ptr.p + P, sizeof(uint64_t)));
    if (ends[0] != 0) { idx->pos_ptr.release(); return ctx->fail(SS_ERR_INVALID, "ss_index_set_positions: pos_ptr[0] != 0"); }
    if (P) {
        // monotone + first 0 + last = total  =>  every range lies inside pos[0, total)
        ss::DevBuf<uint32_t> err;
        SS_HIP(ctx, err.alloc(1));
        SS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_check_pos, dim3(std::min<unsigned>(ss::div_up(P, TPB), 16384u)), dim3(TPB), 0, st, idx->pos_ptr.p, P, err.p);
        uint32_t h_err = 0;
        SS_HIP(ctx, ss::fetch(ctx, st, &h_err, err.p, sizeof(uint32_t)));
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
    idx->mag2_valid = false;       // given magnitudes: their squares are not the exact sums an incremental update needs
    return SS_OK;
}

}  // extern "C"
