// score_small.hip — k_score_small: one workgroup per SMALL query (retrieval/main_retrieve.go:50-103,
// get_metadata.go:31-69, util.go:48-54).
//
// A typical web query hits tail terms: a few hundred to a few thousand postings in all.  The two big kernels are built for
// millions — k_score_slices marches 512 threads through a slice set-up, a window plan, a filter and an exact stage and then
// searches the winners' postings again to explain them (64 us per slice for 3.8 windows of work, DESIGN K4b); k_score_wave wants
// lists long enough for a threshold floor.  Below SMALL_CAP postings none of that pays: this kernel reads EVERY posting of the
// query once, adds its float32 weight (times the token's multiplicity, Q8) to the document's title or body sum in an LDS hash
// table — float64, order-free exact like the other kernels' exact stages —, runs the reference's float64 arithmetic literally
// for every candidate document (final_rank), keeps the k best in the running top-k the other kernels use (same total order:
// FinalRank descending, ties by ascending doc id, NaN last) and writes the ss_hit rows itself: no filter, no slices, no merge
// launch, no second look at the lists.  No assumption about the inputs either (negative or non-finite weights, zero magnitudes,
// hostile priors): what the filter of the other kernels needs "clean" inputs for does not exist here.
// Routed per query by the host (score.hip: option "score.small", default on): no phrase part, <= SMALL_CAP postings, k <= SMALL_MAX_K.
#include "score_common.hpp"

namespace {

constexpr int ST = 256;                    // threads
// Two table sizes, 20 bytes per slot (doc, title sum, body sum): 1024 slots = 20 KB (+ 9 KB of top-k buffer and list table: five
// workgroups per CU) for queries of up to 768 postings — the typical tail query —, 3072 slots = 60 KB (two per CU) up to 2304.
// A query's postings bound its distinct documents, so a table is never more than 3/4 full.
constexpr int S_A = 1024, S_B = 3072;
constexpr uint32_t CAP_A = 768, CAP_B = 2304;
constexpr int SCB = 512;                   // candidate buffer of the running top-k (>= 2k)
constexpr int SMALL_MAX_K = SCB / 2;
constexpr int SL = 2 * SS_MAX_QUERY_TERMS; // (term, field) lists
constexpr int UNR = 4;                     // postings of a thread in flight

#ifdef SSS_PHASES
// variant build (-DSSS_PHASES): cycles thread 0 of every workgroup spends per phase, summed over the grid; printed by ss::score_small_report()
__device__ unsigned long long g_sss[10];
#define SSS_PH(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&g_sss[i], t_ - ph_t); ph_t = t_; } } while (0)
#else
#define SSS_PH(i) do { } while (0)
#endif

template <int NS>
__device__ __forceinline__ uint32_t slot_of(uint32_t doc) { return (uint32_t)(((uint64_t)(doc * 0x9E3779B1u) * (uint64_t)NS) >> 32); }

template <int NS>
constexpr size_t small_lds_bytes() { return (size_t)NS * 20 + (size_t)SCB * 12 + (size_t)SL * 24 + 64; }

template <int NS>
__global__ __launch_bounds__(ST) void k_score_small(ScoreParams p, uint32_t first) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* accT = reinterpret_cast<double*>(smem);                       // [NS]
    double* accB = accT + NS;                                             // [NS]
    uint64_t* cd_key = reinterpret_cast<uint64_t*>(accB + NS);            // [SCB]
    uint64_t* l_rec = cd_key + SCB;                                       // [SL] address of the list's first scoring record
    uint64_t* l_w = l_rec + SL;                                           // [SL] ... and of its first float32 weight
    uint64_t* sc64 = l_w + SL;                                            // [2]
    uint32_t* hkey = reinterpret_cast<uint32_t*>(sc64 + 2);               // [NS]
    uint32_t* cd_doc = hkey + NS;                                         // [SCB]
    uint32_t* l_end = cd_doc + SCB;                                       // [SL] postings of lists 0 .. l (inclusive prefix)
    uint32_t* l_mf = l_end + SL;                                          // [SL] multiplicity << 1 | field (1 = title)
    uint32_t* sc32 = l_mf + SL;                                           // [8]
    const int tid = threadIdx.x;
#ifdef SSS_PHASES
    unsigned long long ph_t = __builtin_readcyclecounter();
    if (threadIdx.x == 0) atomicAdd(&g_sss[9], 1ull);
#endif
    const uint32_t q = p.small_q[first + blockIdx.x];
    const uint32_t t0 = p.q_off[q], nd = p.q_off[q + 1] - t0;
    const uint32_t L = 2 * nd;
    TopK tk{cd_key, cd_doc, &sc32[0], &sc64[0], reinterpret_cast<float*>(&sc32[2]), 0ull, -INFINITY, (uint32_t)SCB};
    uint32_t* overflow = &sc32[1];

    // the query's lists (title and body of every distinct known term), while the table is cleared
    if ((uint32_t)tid < L) {
        const uint32_t term = p.dterm[t0 + ((uint32_t)tid >> 1)];
        const int field = tid & 1;
        const uint64_t* ptr = field ? p.t_ptr : p.b_ptr;
        const uint64_t b = ptr[term], e = ptr[term + 1];
        l_rec[tid] = (uint64_t)((field ? p.t_rec : p.b_rec) + b);
        l_w[tid] = (uint64_t)((field ? p.t_w : p.b_w) + b);
        l_end[tid] = (uint32_t)(e - b);
        l_mf[tid] = p.dmult[t0 + ((uint32_t)tid >> 1)] << 1 | (uint32_t)field;
    }
    for (int i = tid; i < NS; i += ST) { hkey[i] = EMPTY; accT[i] = 0.0; accB[i] = 0.0; }
    if (tid == 0) { sc32[0] = 0u; sc32[1] = 0u; sc64[0] = 0ull; *tk.thr_f = -INFINITY; }
    __syncthreads();
    SSS_PH(0);
    if (tid == 0) {                        // inclusive prefix of the lengths (<= 128 lists: the host admitted <= CAP postings in all)
        uint32_t run = 0;
        for (uint32_t l = 0; l < L; l++) { run += l_end[l]; l_end[l] = run; }
        sc32[4] = run;
    }
    __syncthreads();
    const uint32_t tot = sc32[4];
    SSS_PH(1);

    // ---- every posting once: {doc, weight} -> the document's slot (linear probing), weight * multiplicity into its field's sum
    for (uint32_t i0 = 0; i0 < tot; i0 += ST * UNR) {
        uint32_t doc[UNR], mf[UNR];
        float w[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const uint32_t i = i0 + (uint32_t)u * ST + (uint32_t)tid;
            mf[u] = EMPTY;
            if (i < tot) {
                uint32_t lo = 0, hi = L;                             // the list of posting i: first l with l_end[l] > i
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (l_end[mid] <= i) lo = mid + 1; else hi = mid;
                }
                const uint32_t j = i - (lo ? l_end[lo - 1] : 0u);
                doc[u] = load_doc(l_rec[lo], j);
                w[u] = load_w(l_w[lo], j);
                mf[u] = l_mf[lo];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            if (mf[u] == EMPTY) continue;
            uint32_t s = slot_of<NS>(doc[u]);
            for (;;) {
                const uint32_t old = atomicCAS(&hkey[s], EMPTY, doc[u]);
                if (old == EMPTY || old == doc[u]) break;
                s = s + 1 == (uint32_t)NS ? 0u : s + 1;
            }
            const double v = (double)w[u] * (double)(mf[u] >> 1);    // main_retrieve.go:61-78: a duplicate token counts again (Q8)
            if (mf[u] & 1u) atomicAdd(&accT[s], v); else atomicAdd(&accB[s], v);
        }
    }
    __syncthreads();
    SSS_PH(2);

    // ---- every candidate document: get_metadata.go:31-69, then the running top-k.  A thread owns the slots tid, tid + 256, ...: the
    //      magnitudes of ALL of them are requested before any is used (one memory latency for the table, not one per slot)
    constexpr int PT = NS / ST;
    const double qmag = p.qmag[q];
    const double* probs = (p.probs && p.prior) ? p.probs + (size_t)q * p.k_topics : nullptr;
    uint32_t e_doc[PT];
    uint64_t e_key[PT];
    {
        double T[PT], B[PT], mt[PT], mb[PT];
#pragma unroll
        for (int j = 0; j < PT; j++) {
            const int s = j * ST + tid;
            e_doc[j] = hkey[s];
            T[j] = accT[s];
            B[j] = accB[s];
        }
#pragma unroll
        for (int j = 0; j < PT; j++) {
            // a field without a posting has sum 0, and 0 / (m * q) is 0 (or NaN -> 0) whatever m is: only a non-zero sum needs its
            // magnitude (no branch around the loads: an empty slot or a zero sum reads doc 0's and drops it)
            const uint32_t d = e_doc[j] != EMPTY ? e_doc[j] : 0u;
            mt[j] = p.t_mag[T[j] != 0.0 ? d : 0u];
            mb[j] = p.b_mag[B[j] != 0.0 ? d : 0u];
        }
#pragma unroll
        for (int j = 0; j < PT; j++) {
            e_key[j] = 0ull;
            if (e_doc[j] != EMPTY) {
                const double sqd = probs ? topic_dot(p.prior, probs, p.k_topics, e_doc[j]) : 0.0;
                double title, body, fin;
                final_rank(T[j], B[j], T[j] != 0.0 ? mt[j] : 1.0, B[j] != 0.0 ? mb[j] : 1.0, qmag, sqd, title, body, fin);
                e_key[j] = fkey(fin);
            }
        }
    }
    SSS_PH(3);
    for (;;) {                              // threshold filter into the candidate buffer; overflow -> compact and retry what is left
        const uint64_t thr = *tk.thr;
#pragma unroll
        for (int j = 0; j < PT; j++) {
            if (e_doc[j] != EMPTY) {
                if (e_key[j] >= thr) {
                    const uint32_t i = atomicAdd(tk.count, 1u);
                    if (i < tk.cb) { tk.key[i] = e_key[j]; tk.doc[i] = e_doc[j]; e_doc[j] = EMPTY; }
                    else *overflow = 1;
                } else {
                    e_doc[j] = EMPTY;
                }
            }
        }
        lds_barrier();
        if (!*overflow) break;
        topk_compact_inl(tk, p.k);
        if (tid == 0) *overflow = 0;
        lds_barrier();
    }
    SSS_PH(4);
    topk_compact_inl(tk, p.k);
    const uint32_t n_out = sc32[0];
    SSS_PH(5);

    // ---- the hits: the winners' sums are still in the table
    for (uint32_t i = tid; i < (uint32_t)p.k; i += ST) {
        ss_hit h;
        h.doc = 0; h._pad = 0; h.title = 0.0; h.body = 0.0; h.pagerank = 0.0; h.final = 0.0;
        if (i < n_out) {
            const uint32_t d = cd_doc[i];
            uint32_t s = slot_of<NS>(d);
            while (hkey[s] != d) s = s + 1 == (uint32_t)NS ? 0u : s + 1;
            const double T = accT[s], B = accB[s];
            const double mt = T != 0.0 ? p.t_mag[d] : 1.0, mb = B != 0.0 ? p.b_mag[d] : 1.0;
            const double sqd = probs ? topic_dot(p.prior, probs, p.k_topics, d) : 0.0;
            double title, body, fin;
            final_rank(T, B, mt, mb, qmag, sqd, title, body, fin);
            h.doc = d; h.title = title; h.body = body; h.pagerank = sqd; h.final = fin;
        }
        p.hits[(size_t)q * p.k + i] = h;
    }
    if (tid == 0) p.n_hits[q] = (int32_t)n_out;
    SSS_PH(6);
}

}  // namespace

namespace ss {

uint32_t score_small_cap() { return CAP_B; }
uint32_t score_small_cap_a() { return CAP_A; }
int score_small_max_k() { return SMALL_MAX_K; }
// the queries small_q[0 .. n_a) take the 1024-slot table (<= score_small_cap_a() postings each), small_q[n_a .. n_a + n_b) the 3072-slot one
int32_t launch_score_small(const void* params, unsigned n_a, unsigned n_b, hipStream_t st) {
    static bool attr_set[64] = {};         // per device: the attribute belongs to the device's copy of the kernel
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_score_small<S_B>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds_bytes<S_B>());
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_score_small<S_A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds_bytes<S_A>());
        if (e != hipSuccess) return (int32_t)e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const ScoreParams& p = *static_cast<const ScoreParams*>(params);
    if (n_b) hipLaunchKernelGGL(k_score_small<S_B>, dim3(n_b), dim3(ST), small_lds_bytes<S_B>(), st, p, (uint32_t)n_a);    // the larger queries first
    if (n_a) hipLaunchKernelGGL(k_score_small<S_A>, dim3(n_a), dim3(ST), small_lds_bytes<S_A>(), st, p, 0u);
    return 0;
}

void score_small_report() {
#ifdef SSS_PHASES
    unsigned long long h[10];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sss), sizeof(h)) == hipSuccess && h[9]) {
        const char* nm[7] = {"lists+clear", "prefix", "postings->table", "magnitudes+scores", "admit", "final sort", "hits"};
        fprintf(stderr, "[k_score_small phases, %llu workgroups] cycles per workgroup:", h[9]);
        for (int i = 0; i < 7; i++) fprintf(stderr, " %s %.0f", nm[i], (double)h[i] / (double)h[9]);
        fprintf(stderr, "\n");
    }
#endif
}

}  // namespace ss
