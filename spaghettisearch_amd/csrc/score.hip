// score.hip — batched OR-query cosine scorer + PageRank blend + top-k for gfx950 (MI355X).
//
// Replaces, per query (retrieval/main_retrieve.go:50-103):
//   getFromInverted  :204-247   fetch body+title postings of every query token
//   aggregation      :61-69     OR-union by doc, weights appended per token (duplicates count twice)
//   genAggrDocs      :170-187   TitleRank = sum float64(w_title), BodyRank = sum float64(w_body)
//   computeFinalRank get_metadata.go:31-69
//                               Body  /= mag_body  * sqrt(queryLength)   (:53,:57)
//                               Title /= mag_title * sqrt(queryLength)   (:58)
//                               NaN -> 0                                 (:61-66)
//                               sqd   = sum_t topicProbs[t]*PR[doc][t]   (:39-42)
//                               Final = (0.33*sqd+0.38*Title+0.29*Body)*100 (:69)
//   appendSort + cut util.go:48-54, main_retrieve.go:99-103  descending FinalRank, first k
//
// Device design (HBM-bound: 8 B per posting streamed once, 16 B magnitudes per candidate):
//   * postings stay resident, term-major, doc-sorted; the host plans (it keeps df per term)
//     and cuts every query's doc range into slices of ~SLICE_TARGET postings -> one workgroup
//     per (query, slice), thousands of workgroups per 1024-query batch;
//   * k_score_slices: the workgroup walks its slice in windows of <= CAP postings taken
//     proportionally from all (term, field) lists, stages the doc ids in LDS, cuts the window at a
//     common doc bound so that every posting of a doc lands in the same window, accumulates
//     (title, body) per doc in an LDS hash table with ds atomics (float32 addends summed in float64
//     are exact, so order does not matter), then scores every touched doc in registers and keeps a
//     running top-k in LDS (threshold filter + bitonic compaction);
//   * k_merge_topk: one workgroup per query merges its slices' top-k lists, then re-derives
//     title/body/pagerank of the k winners by binary search and writes ss_hit rows.
//   Ties: ascending doc id (Q10); NaN finals last.
#include "index.hpp"

#include <algorithm>
#include <cmath>
#include <memory>

namespace {

constexpr int TPB = 256;
constexpr int CAP = 1024;          // postings per window
constexpr int HT = 2048;           // hash slots (load factor <= 0.5)
constexpr int CB = 2048;           // candidate buffer entries (>= 2*k_max... k <= 1024)
constexpr int MAXL = 2 * SS_MAX_QUERY_TERMS;   // (term, field) lists per query
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint64_t SLICE_TARGET = 32768;
constexpr uint32_t MAX_SLICES_PER_Q = 256;

struct SliceDesc {
    uint32_t q;
    uint32_t dlo, dhi;   // doc range [dlo, dhi)
    uint32_t pad;
};

struct ScoreParams {
    // index
    const uint64_t* t_ptr; const uint32_t* t_doc; const float* t_w;
    const uint64_t* b_ptr; const uint32_t* b_doc; const float* b_w;
    const double* mag2;        // [n_docs][2] = (title, body)
    const double* prior;       // [n_docs][k_topics] or null
    int32_t k_topics;
    // batch
    const uint32_t* q_off;     // [n_q+1] into dterm/dmult
    const uint32_t* dterm;     // distinct known terms per query, first-occurrence order
    const uint32_t* dmult;     // multiplicity of each
    const double* qmag;        // [n_q] sqrt(queryLength)
    const double* probs;       // [n_q][k_topics] or null
    const uint32_t* slice_base;// [n_q+1]
    const SliceDesc* slices;
    int32_t k;
    // scratch / outputs
    uint64_t* so_key; uint32_t* so_doc; uint32_t* so_cnt;   // per slice top-k
    ss_hit* hits; int32_t* n_hits;
};

// total order: larger key = better; NaN lowest
__device__ __forceinline__ uint64_t fkey(double f) {
    if (f != f) return 0ull;
    const uint64_t b = (uint64_t)__double_as_longlong(f);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ bool better(uint64_t ka, uint32_t da, uint64_t kb, uint32_t db) {
    return ka > kb || (ka == kb && da < db);
}

__device__ __forceinline__ uint64_t lower_bound_g(const uint32_t* __restrict__ a, uint64_t lo, uint64_t hi, uint32_t v) {
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// get_metadata.go:53-69 for one candidate
__device__ __forceinline__ void final_rank(double T, double B, double mt, double mb, double qmag, double sqd,
                                           double& title, double& body, double& fin) {
    body = B / (mb * qmag);                          // :57
    title = T / (mt * qmag);                         // :58
    if (body != body) body = 0.0;                    // :61-63
    if (title != title) title = 0.0;                 // :64-66
    fin = (0.33 * sqd + 0.38 * title + 0.29 * body) * 100.0;   // :69
}

__device__ __forceinline__ double topic_dot(const double* __restrict__ prior, const double* __restrict__ probs, int K, uint32_t doc) {
    double sqd = 0.0;                                // get_metadata.go:39-42, topic order
    const double* pr = prior + (size_t)doc * K;
    for (int t = 0; t < K; t++) sqd += probs[t] * pr[t];
    return sqd;
}

// ---- running top-k in LDS ------------------------------------------------------
struct TopK {
    uint64_t* key;    // [CB]
    uint32_t* doc;    // [CB]
    uint32_t* count;  // shared scalar
    uint64_t* thr;    // shared scalar: admit keys >= thr
};

// Sort the candidate buffer descending (better first) and keep the k best. All threads call.
__device__ void topk_compact(const TopK& tk, int k) {
    __syncthreads();
    const uint32_t n = min(*tk.count, (uint32_t)CB);
    uint32_t n2 = 64;
    while (n2 < n) n2 <<= 1;
    for (uint32_t i = n + threadIdx.x; i < n2; i += TPB) { tk.key[i] = 0ull; tk.doc[i] = EMPTY; }   // worst sentinels
    __syncthreads();
    for (uint32_t size = 2; size <= n2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t i = threadIdx.x; i < (n2 >> 1); i += TPB) {
                const uint32_t lo = 2 * i - (i & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool desc = ((lo & size) == 0);     // this run sorted best-first
                const uint64_t ka = tk.key[lo], kb = tk.key[hi];
                const uint32_t da = tk.doc[lo], db = tk.doc[hi];
                const bool swap = desc ? better(kb, db, ka, da) : better(ka, da, kb, db);
                if (swap) { tk.key[lo] = kb; tk.key[hi] = ka; tk.doc[lo] = db; tk.doc[hi] = da; }
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        const uint32_t keep = min(n, (uint32_t)k);
        *tk.count = keep;
        *tk.thr = keep == (uint32_t)k ? tk.key[k - 1] : 0ull;
    }
    __syncthreads();
}

__device__ __forceinline__ void topk_admit(const TopK& tk, uint64_t key, uint32_t doc) {
    if (key >= *tk.thr) {
        const uint32_t i = atomicAdd(tk.count, 1u);
        if (i < (uint32_t)CB) { tk.key[i] = key; tk.doc[i] = doc; }   // room is guaranteed by the callers
    }
}

// ---- K4: score one (query, doc-range slice) --------------------------------------
__global__ __launch_bounds__(TPB) void k_score_slices(ScoreParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ht_T = reinterpret_cast<double*>(smem);                    // [HT]
    double* ht_B = ht_T + HT;                                          // [HT]
    uint64_t* cd_key = reinterpret_cast<uint64_t*>(ht_B + HT);         // [CB]
    uint64_t* l_cur = cd_key + CB;                                     // [MAXL]
    uint64_t* l_end = l_cur + MAXL;                                    // [MAXL]
    double* l_mult = reinterpret_cast<double*>(l_end + MAXL);          // [MAXL]
    const uint32_t** l_docp = reinterpret_cast<const uint32_t**>(l_mult + MAXL);   // [MAXL]
    const float** l_wp = reinterpret_cast<const float**>(l_docp + MAXL);           // [MAXL]
    uint64_t* sc64 = reinterpret_cast<uint64_t*>(l_wp + MAXL);         // [4] scalars: total_rem, thr
    uint32_t* ht_key = reinterpret_cast<uint32_t*>(sc64 + 4);          // [HT]
    uint32_t* cd_doc = ht_key + HT;                                    // [CB]
    uint32_t* s_doc = cd_doc + CB;                                     // [CAP]
    uint32_t* l_off = s_doc + CAP;                                     // [MAXL+4]
    uint32_t* l_share = l_off + MAXL + 4;                              // [MAXL]
    uint32_t* l_cnt = l_share + MAXL;                                  // [MAXL]
    uint32_t* l_field = l_cnt + MAXL;                                  // [MAXL]
    uint32_t* sc32 = l_field + MAXL;                                   // [8] scalars

    uint64_t& total_rem = sc64[0];
    uint32_t& cand_count = sc32[0];
    uint32_t& dw = sc32[1];
    uint32_t& wave0_total = sc32[2];
    TopK tk{cd_key, cd_doc, &sc32[0], &sc64[1]};

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const SliceDesc sd = p.slices[blockIdx.x];
    const uint32_t q = sd.q;
    const uint32_t t0 = p.q_off[q], nd = p.q_off[q + 1] - t0;
    const int L = (int)(2 * nd);
    const double qmag = p.qmag[q];
    const double* probs = (p.probs && p.prior) ? p.probs + (size_t)q * p.k_topics : nullptr;

    for (int i = tid; i < HT; i += TPB) { ht_key[i] = EMPTY; ht_T[i] = 0.0; ht_B[i] = 0.0; }
    if (tid == 0) { cand_count = 0; sc64[1] = 0ull; }
    if (tid < L) {
        const uint32_t term = p.dterm[t0 + (tid >> 1)];
        const int field = tid & 1;                     // 0 = body, 1 = title
        const uint64_t* ptr = field ? p.t_ptr : p.b_ptr;
        const uint32_t* docs = field ? p.t_doc : p.b_doc;
        const uint64_t p0 = ptr[term], p1 = ptr[term + 1];
        l_cur[tid] = lower_bound_g(docs, p0, p1, sd.dlo);
        l_end[tid] = sd.dhi == 0xFFFFFFFFu ? p1 : lower_bound_g(docs, p0, p1, sd.dhi);
        l_mult[tid] = (double)p.dmult[t0 + (tid >> 1)];
        l_docp[tid] = docs;
        l_wp[tid] = field ? p.t_w : p.b_w;
        l_field[tid] = field;
    }
    __syncthreads();

    for (;;) {
        // (1) remaining postings per list, proportional shares of the window
        if (tid == 0) { total_rem = 0ull; dw = sd.dhi; }
        __syncthreads();
        uint64_t rem = 0;
        if (tid < L) {
            rem = l_end[tid] - l_cur[tid];
            if (rem) atomicAdd(reinterpret_cast<unsigned long long*>(&total_rem), (unsigned long long)rem);
        }
        __syncthreads();
        const uint64_t tot = total_rem;
        if (tot == 0) break;
        uint32_t share = 0;
        if (tid < L && rem) share = (uint32_t)min(rem, 1ull + ((uint64_t)(CAP - L) * rem) / tot);
        // (2) exclusive prefix of the shares over the (<=128) lists: two waves scan
        uint32_t incl = share;
        if (wave < 2) {
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            if (wave == 0 && lane == 63) wave0_total = incl;
        }
        __syncthreads();
        if (wave == 1) incl += wave0_total;
        if (tid < MAXL) {
            if (tid < L) { l_share[tid] = share; l_off[tid] = incl - share; }
            if (tid == L - 1 || (L == 0 && tid == 0)) l_off[L] = incl;
        }
        __syncthreads();
        const uint32_t n_stage = l_off[L];

        // (3) stage doc ids in LDS (weights stay in registers)
        uint32_t my_l[CAP / TPB], my_doc[CAP / TPB];
        float my_w[CAP / TPB];
#pragma unroll
        for (int j = 0; j < CAP / TPB; j++) {
            const uint32_t i = tid + j * TPB;
            my_l[j] = EMPTY;
            if (i < n_stage) {
                int lo = 0, hi = L;            // largest l with l_off[l] <= i
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (l_off[mid] <= i) lo = mid; else hi = mid;
                }
                const uint64_t pp = l_cur[lo] + (i - l_off[lo]);
                const uint32_t d = l_docp[lo][pp];
                my_l[j] = lo;
                my_doc[j] = d;
                my_w[j] = l_wp[lo][pp];
                s_doc[i] = d;
            }
        }
        __syncthreads();
        // (4) common doc bound: the window holds every posting with doc < dw
        if (tid < L && l_share[tid]) {
            const uint64_t e = l_cur[tid] + l_share[tid];
            if (e < l_end[tid]) atomicMin(&dw, s_doc[l_off[tid] + l_share[tid] - 1] + 1u);
        }
        __syncthreads();
        const uint32_t bound = dw;
        // (5) how much of each staged chunk is inside the window; advance the cursors
        if (tid < L) {
            const uint32_t o = l_off[tid];
            uint32_t lo = 0, hi = l_share[tid];
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_doc[o + mid] < bound) lo = mid + 1; else hi = mid;
            }
            l_cnt[tid] = lo;
            l_cur[tid] += lo;
        }
        __syncthreads();
        // (6) accumulate per doc (main_retrieve.go:61-69,170-187); float32 addends in float64: exact
#pragma unroll
        for (int j = 0; j < CAP / TPB; j++) {
            const uint32_t l = my_l[j];
            if (l != EMPTY && (tid + j * TPB - l_off[l]) < l_cnt[l]) {
                const uint32_t d = my_doc[j];
                uint32_t h = (d * 2654435761u) >> (32 - 11);
                for (;;) {
                    const uint32_t prev = atomicCAS(&ht_key[h], EMPTY, d);
                    if (prev == EMPTY || prev == d) break;
                    h = (h + 1) & (HT - 1);
                }
                const double v = (double)my_w[j] * l_mult[l];
                atomicAdd(l_field[l] ? &ht_T[h] : &ht_B[h], v);
            }
        }
        // (7) room for every doc of this window in the candidate buffer?
        if (cand_count > (uint32_t)(CB - CAP)) topk_compact(tk, p.k);
        __syncthreads();
        // (8) score every touched doc (get_metadata.go:31-69), filter, reset the table
        uint32_t e_doc[HT / TPB];
        double e_mt[HT / TPB], e_mb[HT / TPB];
#pragma unroll
        for (int j = 0; j < HT / TPB; j++) {
            const uint32_t d = ht_key[tid + j * TPB];
            e_doc[j] = d;
            if (d != EMPTY) {
                const double2 m = *reinterpret_cast<const double2*>(p.mag2 + 2 * (size_t)d);
                e_mt[j] = m.x;
                e_mb[j] = m.y;
            }
        }
#pragma unroll
        for (int j = 0; j < HT / TPB; j++) {
            const uint32_t d = e_doc[j];
            if (d != EMPTY) {
                const int h = tid + j * TPB;
                const double T = ht_T[h], B = ht_B[h];
                ht_key[h] = EMPTY; ht_T[h] = 0.0; ht_B[h] = 0.0;
                const double sqd = probs ? topic_dot(p.prior, probs, p.k_topics, d) : 0.0;
                double title, body, fin;
                final_rank(T, B, e_mt[j], e_mb[j], qmag, sqd, title, body, fin);
                topk_admit(tk, fkey(fin), d);
            }
        }
        __syncthreads();
    }

    topk_compact(tk, p.k);
    const uint32_t n_out = cand_count;
    for (uint32_t i = tid; i < n_out; i += TPB) {
        p.so_key[(size_t)blockIdx.x * p.k + i] = cd_key[i];
        p.so_doc[(size_t)blockIdx.x * p.k + i] = cd_doc[i];
    }
    if (tid == 0) p.so_cnt[blockIdx.x] = n_out;
}

constexpr size_t SCORE_LDS = (size_t)HT * 16 + (size_t)CB * 8 + (size_t)MAXL * 8 * 5 + 4 * 8 +
                             ((size_t)HT + CB + CAP + (MAXL + 4) + 4 * (size_t)MAXL + 8) * 4;

// ---- K5: merge a query's slices, explain the winners ------------------------------
__global__ __launch_bounds__(TPB) void k_merge_topk(ScoreParams p) {
    __shared__ uint64_t cd_key[CB];
    __shared__ uint32_t cd_doc[CB];
    __shared__ double accT[SS_MAX_TOPK], accB[SS_MAX_TOPK];
    __shared__ uint32_t sc32[2];
    __shared__ uint64_t sc64[1];
    TopK tk{cd_key, cd_doc, &sc32[0], &sc64[0]};
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const int k = p.k;
    if (tid == 0) { sc32[0] = 0; sc64[0] = 0ull; }
    __syncthreads();
    for (uint32_t s = p.slice_base[q]; s < p.slice_base[q + 1]; s++) {
        if (sc32[0] > (uint32_t)(CB - k)) topk_compact(tk, k);
        __syncthreads();
        const uint32_t n = p.so_cnt[s];
        for (uint32_t i = tid; i < n; i += TPB) topk_admit(tk, p.so_key[(size_t)s * k + i], p.so_doc[(size_t)s * k + i]);
        __syncthreads();
    }
    topk_compact(tk, k);
    const uint32_t n_out = sc32[0];

    // explain: TitleRank/BodyRank of the winners, re-derived from the posting lists
    for (uint32_t i = tid; i < n_out; i += TPB) { accT[i] = 0.0; accB[i] = 0.0; }
    __syncthreads();
    const uint32_t t0 = p.q_off[q], nd = p.q_off[q + 1] - t0;
    const uint32_t L = 2 * nd;
    for (uint32_t task = tid; task < n_out * L; task += TPB) {
        const uint32_t i = task / L, l = task % L;
        const uint32_t term = p.dterm[t0 + (l >> 1)];
        const int field = l & 1;
        const uint64_t* ptr = field ? p.t_ptr : p.b_ptr;
        const uint32_t* docs = field ? p.t_doc : p.b_doc;
        const uint32_t d = cd_doc[i];
        const uint64_t p1 = ptr[term + 1];
        const uint64_t pos = lower_bound_g(docs, ptr[term], p1, d);
        if (pos < p1 && docs[pos] == d) {
            const double v = (double)(field ? p.t_w : p.b_w)[pos] * (double)p.dmult[t0 + (l >> 1)];
            atomicAdd(field ? &accT[i] : &accB[i], v);
        }
    }
    __syncthreads();
    const double qmag = p.qmag[q];
    const double* probs = (p.probs && p.prior) ? p.probs + (size_t)q * p.k_topics : nullptr;
    for (uint32_t i = tid; i < (uint32_t)k; i += TPB) {
        ss_hit h;
        h.doc = 0; h._pad = 0; h.title = 0.0; h.body = 0.0; h.pagerank = 0.0; h.final = 0.0;
        if (i < n_out) {
            const uint32_t d = cd_doc[i];
            const double sqd = probs ? topic_dot(p.prior, probs, p.k_topics, d) : 0.0;
            double title, body, fin;
            final_rank(accT[i], accB[i], p.mag2[2 * (size_t)d], p.mag2[2 * (size_t)d + 1], qmag, sqd, title, body, fin);
            h.doc = d; h.title = title; h.body = body; h.pagerank = sqd; h.final = fin;
        }
        p.hits[(size_t)q * k + i] = h;
    }
    if (tid == 0) p.n_hits[q] = (int32_t)n_out;
}

__global__ void k_pack_mag(const double* __restrict__ mt, const double* __restrict__ mb, uint64_t n, double* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[2 * i] = mt[i]; out[2 * i + 1] = mb[i]; }
}
// rank [K][N] topic-major -> prior [N][K] node-major
__global__ void k_transpose_prior(const double* __restrict__ in, uint64_t n, int K, double* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * (uint64_t)K) return;
    const uint64_t doc = i / K;
    const int t = (int)(i % K);
    out[i] = in[(uint64_t)t * n + doc];
}

}  // namespace

struct ss_scorer {
    ss_ctx* ctx = nullptr;
    ss_index* title = nullptr;
    ss_index* body = nullptr;
    uint64_t n_docs = 0, n_terms = 0;
    ss::DevBuf<double> mag2;
    ss::DevBuf<double> prior;
    int k_topics = 0;
    // per-call workspaces, grow-only (no hipMalloc/hipFree on the steady-state query path)
    ss::DevBuf<uint32_t> d_qoff, d_dterm, d_dmult, d_sbase, d_so_doc, d_so_cnt;
    ss::DevBuf<double> d_qmag, d_probs;
    ss::DevBuf<SliceDesc> d_slices;
    ss::DevBuf<uint64_t> d_so_key;
    ss::DevBuf<ss_hit> d_hits;
    ss::DevBuf<int32_t> d_nhits;
};

namespace {
template <typename T>
hipError_t ensure(ss::DevBuf<T>& b, size_t n) {
    if (b.p && b.n >= n) return hipSuccess;
    return b.alloc(n + n / 2 + 16);
}
}  // namespace

extern "C" {

int32_t ss_scorer_create(ss_ctx* ctx, ss_index* title, ss_index* body, ss_scorer** out) {
    if (!ctx) return SS_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!out) return ctx->fail(SS_ERR_INVALID, "ss_scorer_create: out is NULL");
    *out = nullptr;
    if (!title || !body) return ctx->fail(SS_ERR_INVALID, "ss_scorer_create: NULL index");
    if (title->ctx != ctx || body->ctx != ctx) return ctx->fail(SS_ERR_INVALID, "ss_scorer_create: index from another context");
    if (title->n_docs != body->n_docs || title->n_terms != body->n_terms)
        return ctx->fail(SS_ERR_INVALID, "ss_scorer_create: title/body tables disagree on n_docs or n_terms");
    if (!title->weighted || !body->weighted)
        return ctx->fail(SS_ERR_STATE, "ss_scorer_create: run ss_tfidf_build (or ss_index_set_weighted) on both tables first");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<ss_scorer> s(new (std::nothrow) ss_scorer());
    if (!s) return ctx->fail(SS_ERR_OOM, "ss_scorer_create: host OOM");
    s->ctx = ctx;
    s->title = title;
    s->body = body;
    s->n_docs = title->n_docs;
    s->n_terms = title->n_terms;
    SS_HIP(ctx, s->mag2.alloc(2 * s->n_docs));
    hipLaunchKernelGGL(k_pack_mag, dim3(ss::div_up(s->n_docs, TPB)), dim3(TPB), 0, ctx->stream, (const double*)title->mag.p,
                       (const double*)body->mag.p, s->n_docs, s->mag2.p);
    SS_HIP(ctx, hipGetLastError());
    SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_score_slices), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)SCORE_LDS));
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    title->users++;
    body->users++;
    *out = s.release();
    return SS_OK;
}

int32_t ss_scorer_destroy(ss_scorer* s) {
    if (!s) return SS_ERR_INVALID;
    ss_ctx* ctx = s->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    s->title->users--;
    s->body->users--;
    delete s;
    return SS_OK;
}

int32_t ss_scorer_set_prior(ss_scorer* s, int32_t k_topics, const double* rank) {
    if (!s) return SS_ERR_INVALID;
    ss_ctx* ctx = s->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    if (k_topics < 0 || k_topics > SS_MAX_TOPICS) return ctx->fail(SS_ERR_INVALID, "ss_scorer_set_prior: bad k_topics");
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (k_topics == 0 || !rank) {
        s->prior.release();
        s->k_topics = 0;
        return SS_OK;
    }
    const uint64_t n = s->n_docs * (uint64_t)k_topics;
    ss::DevBuf<double> tmp;
    SS_HIP(ctx, tmp.alloc(n));
    SS_HIP(ctx, hipMemcpyAsync(tmp.p, rank, n * sizeof(double), hipMemcpyDefault, ctx->stream));
    SS_HIP(ctx, s->prior.alloc(n));
    hipLaunchKernelGGL(k_transpose_prior, dim3(ss::div_up(n, TPB)), dim3(TPB), 0, ctx->stream, (const double*)tmp.p, s->n_docs,
                       k_topics, s->prior.p);
    SS_HIP(ctx, hipGetLastError());
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->k_topics = k_topics;
    return SS_OK;
}

int32_t ss_score_topk(ss_scorer* s, int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms, const int32_t* query_len,
                      const double* topic_probs, int32_t k, ss_hit* hits_out, int32_t* n_hits_out) {
    if (!s) return SS_ERR_INVALID;
    ss_ctx* ctx = s->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (n_q < 0 || !q_ptr || !hits_out || !n_hits_out) return ctx->fail(SS_ERR_INVALID, "ss_score_topk: NULL argument or n_q < 0");
    if (k < 1) return ctx->fail(SS_ERR_INVALID, "ss_score_topk: k < 1");
    if (k > SS_MAX_TOPK) return ctx->fail(SS_ERR_UNSUPPORTED, "ss_score_topk: k %d > SS_MAX_TOPK %d", k, SS_MAX_TOPK);
    if (topic_probs && s->k_topics == 0) return ctx->fail(SS_ERR_STATE, "ss_score_topk: topic_probs given but no prior set (ss_scorer_set_prior)");
    if (n_q == 0) return SS_OK;

    // ---- host-side plan (the host keeps df per term; queries are tiny) -------------
    std::vector<uint32_t> h_qptr(n_q + 1);
    SS_HIP(ctx, hipMemcpy(h_qptr.data(), q_ptr, (n_q + 1) * sizeof(uint32_t), hipMemcpyDefault));
    const uint32_t n_tok = h_qptr[n_q];
    for (int q = 0; q < n_q; q++)
        if (h_qptr[q + 1] < h_qptr[q]) return ctx->fail(SS_ERR_INVALID, "ss_score_topk: q_ptr not non-decreasing");
    if (n_tok && !q_terms) return ctx->fail(SS_ERR_INVALID, "ss_score_topk: q_terms is NULL");
    std::vector<uint32_t> h_terms(n_tok);
    if (n_tok) SS_HIP(ctx, hipMemcpy(h_terms.data(), q_terms, n_tok * sizeof(uint32_t), hipMemcpyDefault));
    std::vector<int32_t> h_qlen(n_q);
    if (query_len) SS_HIP(ctx, hipMemcpy(h_qlen.data(), query_len, n_q * sizeof(int32_t), hipMemcpyDefault));
    else for (int q = 0; q < n_q; q++) h_qlen[q] = (int32_t)(h_qptr[q + 1] - h_qptr[q]);

    const std::vector<uint64_t>& tp = s->title->h_term_ptr;
    const std::vector<uint64_t>& bp = s->body->h_term_ptr;
    std::vector<uint32_t> h_qoff(n_q + 1, 0), h_dterm, h_dmult, h_sbase(n_q + 1, 0);
    std::vector<double> h_qmag(n_q);
    std::vector<SliceDesc> h_slices;
    h_dterm.reserve(n_tok);
    h_dmult.reserve(n_tok);
    for (int q = 0; q < n_q; q++) {
        const size_t d0 = h_dterm.size();
        uint64_t tot = 0;
        for (uint32_t i = h_qptr[q]; i < h_qptr[q + 1]; i++) {
            const uint32_t t = h_terms[i];
            if ((uint64_t)t >= s->n_terms) continue;            // unknown word: ErrKeyNotFound -> no postings (main_retrieve.go:193,218)
            size_t j = d0;
            while (j < h_dterm.size() && h_dterm[j] != t) j++;
            if (j < h_dterm.size()) { h_dmult[j]++; continue; } // duplicate token: counted again (Q8)
            h_dterm.push_back(t);
            h_dmult.push_back(1);
            tot += (tp[t + 1] - tp[t]) + (bp[t + 1] - bp[t]);
        }
        if (h_dterm.size() - d0 > SS_MAX_QUERY_TERMS)
            return ctx->fail(SS_ERR_UNSUPPORTED, "ss_score_topk: query %d has more than %d distinct terms", q, SS_MAX_QUERY_TERMS);
        h_qoff[q + 1] = (uint32_t)h_dterm.size();
        h_qmag[q] = std::sqrt((double)h_qlen[q]);               // get_metadata.go:53
        uint64_t ns = std::max<uint64_t>(1, (tot + SLICE_TARGET - 1) / SLICE_TARGET);
        ns = std::min<uint64_t>(ns, std::min<uint64_t>(MAX_SLICES_PER_Q, s->n_docs));
        for (uint64_t j = 0; j < ns; j++) {
            SliceDesc sd;
            sd.q = (uint32_t)q;
            sd.dlo = (uint32_t)(s->n_docs * j / ns);
            sd.dhi = j + 1 == ns ? 0xFFFFFFFFu : (uint32_t)(s->n_docs * (j + 1) / ns);
            sd.pad = 0;
            h_slices.push_back(sd);
        }
        h_sbase[q + 1] = (uint32_t)h_slices.size();
    }
    const size_t n_slices = h_slices.size();
    const size_t n_d = h_dterm.size();

    // ---- device buffers for this call ------------------------------------------------
    auto& d_qoff = s->d_qoff; auto& d_dterm = s->d_dterm; auto& d_dmult = s->d_dmult; auto& d_sbase = s->d_sbase;
    auto& d_so_doc = s->d_so_doc; auto& d_so_cnt = s->d_so_cnt; auto& d_qmag = s->d_qmag; auto& d_probs = s->d_probs;
    auto& d_slices = s->d_slices; auto& d_so_key = s->d_so_key; auto& d_hits = s->d_hits; auto& d_nhits = s->d_nhits;
    SS_HIP(ctx, ensure(d_qoff, n_q + 1));
    SS_HIP(ctx, ensure(d_dterm, n_d));
    SS_HIP(ctx, ensure(d_dmult, n_d));
    SS_HIP(ctx, ensure(d_sbase, n_q + 1));
    SS_HIP(ctx, ensure(d_qmag, n_q));
    SS_HIP(ctx, ensure(d_slices, n_slices));
    SS_HIP(ctx, ensure(d_so_key, n_slices * k));
    SS_HIP(ctx, ensure(d_so_doc, n_slices * k));
    SS_HIP(ctx, ensure(d_so_cnt, n_slices));
    SS_HIP(ctx, ensure(d_hits, (size_t)n_q * k));
    SS_HIP(ctx, ensure(d_nhits, n_q));
    SS_HIP(ctx, hipMemcpyAsync(d_qoff.p, h_qoff.data(), (n_q + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if (n_d) {
        SS_HIP(ctx, hipMemcpyAsync(d_dterm.p, h_dterm.data(), n_d * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        SS_HIP(ctx, hipMemcpyAsync(d_dmult.p, h_dmult.data(), n_d * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    }
    SS_HIP(ctx, hipMemcpyAsync(d_sbase.p, h_sbase.data(), (n_q + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    SS_HIP(ctx, hipMemcpyAsync(d_qmag.p, h_qmag.data(), n_q * sizeof(double), hipMemcpyHostToDevice, st));
    SS_HIP(ctx, hipMemcpyAsync(d_slices.p, h_slices.data(), n_slices * sizeof(SliceDesc), hipMemcpyHostToDevice, st));
    if (topic_probs) {
        SS_HIP(ctx, ensure(d_probs, (size_t)n_q * s->k_topics));
        SS_HIP(ctx, hipMemcpyAsync(d_probs.p, topic_probs, (size_t)n_q * s->k_topics * sizeof(double), hipMemcpyDefault, st));
    }

    ScoreParams p{};
    p.t_ptr = s->title->term_ptr.p; p.t_doc = s->title->post_doc.p; p.t_w = s->title->post_w.p;
    p.b_ptr = s->body->term_ptr.p; p.b_doc = s->body->post_doc.p; p.b_w = s->body->post_w.p;
    p.mag2 = s->mag2.p;
    p.prior = s->k_topics ? s->prior.p : nullptr;
    p.k_topics = s->k_topics;
    p.q_off = d_qoff.p; p.dterm = d_dterm.p; p.dmult = d_dmult.p; p.qmag = d_qmag.p;
    p.probs = topic_probs ? d_probs.p : nullptr;
    p.slice_base = d_sbase.p; p.slices = d_slices.p;
    p.k = k;
    p.so_key = d_so_key.p; p.so_doc = d_so_doc.p; p.so_cnt = d_so_cnt.p;
    p.hits = d_hits.p; p.n_hits = d_nhits.p;

    SS_HIP(ctx, hipEventRecord(ctx->ev[1][0], st));
    hipLaunchKernelGGL(k_score_slices, dim3((unsigned)n_slices), dim3(TPB), SCORE_LDS, st, p);
    hipLaunchKernelGGL(k_merge_topk, dim3((unsigned)n_q), dim3(TPB), 0, st, p);
    SS_HIP(ctx, hipEventRecord(ctx->ev[1][1], st));
    ctx->ev_valid[1] = true;
    SS_HIP(ctx, hipGetLastError());
    SS_HIP(ctx, hipMemcpyAsync(hits_out, d_hits.p, (size_t)n_q * k * sizeof(ss_hit), hipMemcpyDefault, st));
    SS_HIP(ctx, hipMemcpyAsync(n_hits_out, d_nhits.p, n_q * sizeof(int32_t), hipMemcpyDefault, st));
    SS_HIP(ctx, hipStreamSynchronize(st));   // the host plan vectors above are released on return
    return SS_OK;
}

}  // extern "C"
