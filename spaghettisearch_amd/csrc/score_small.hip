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
// Two table sizes, 20 bytes per slot (doc, title sum, body sum): 1024 slots = 20 KB (+ 5 KB of candidate buffer, histogram and list
// table: six workgroups per CU) for queries of up to 768 postings, 2048 slots = 40 KB (three per CU) up to 1664.
// A query's postings bound its distinct documents, so a table is never more than 13/16 full.
constexpr int S_A = 1024, S_B = 2048;
constexpr uint32_t CAP_A = 768, CAP_B = 1664;
constexpr int SCB = 256;                   // candidate buffer: what the selection admits (<= max(64, 2^ceil(log2 k)))
constexpr int SMALL_MAX_K = SCB;
constexpr int SL = SS_SMALL_MAX_LISTS;     // (term, field) lists with postings
constexpr int UNR = 8;                     // postings of a thread in flight (2048 per pass: every admitted query in one round of loads)

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
constexpr size_t small_lds_bytes() { return (size_t)NS * 20 + (size_t)SCB * 12 + (size_t)SL * 24 + 256 * 4 + 96; }

// eight bits of the 96-bit selection key {fkey(FinalRank), ~doc} from bit `sh` (0 .. 88) up: larger = better (order.hpp: FinalRank
// descending, equal finals by ascending doc id)
__device__ __forceinline__ uint32_t sel_digit(uint64_t key, uint32_t doc, int sh) {
    const uint32_t lo = ~doc;
    if (sh >= 32) return (uint32_t)(key >> (sh - 32)) & 255u;
    // (sh < 32: the byte may straddle the key's low bits and the doc word)
    const uint64_t w = ((key & 0xFFFFFFFFull) << 32 | (uint64_t)lo) >> sh;
    return (uint32_t)w & 255u;
}

template <int NS>
__global__ __launch_bounds__(ST) void k_score_small(ScoreParams p, uint32_t first) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* accT = reinterpret_cast<double*>(smem);                       // [NS]
    double* accB = accT + NS;                                             // [NS]
    uint64_t* cd_key = reinterpret_cast<uint64_t*>(accB + NS);            // [SCB]
    uint64_t* l_rec = cd_key + SCB;                                       // [SL] address of the list's first scoring record
    uint64_t* l_w = l_rec + SL;                                           // [SL] ... and of its first float32 weight
    uint64_t* sc64 = l_w + SL;                                            // [4]: running threshold (unused here), OR and AND of the candidates' keys
    uint32_t* hkey = reinterpret_cast<uint32_t*>(sc64 + 4);               // [NS]
    uint32_t* cd_doc = hkey + NS;                                         // [SCB]
    uint32_t* l_end = cd_doc + SCB;                                       // [SL] postings of lists 0 .. l (inclusive prefix)
    uint32_t* l_mf = l_end + SL;                                          // [SL] multiplicity << 1 | field (1 = title)
    uint32_t* hist = l_mf + SL;                                           // [256] selection histogram
    uint32_t* sc32 = hist + 256;                                          // [12]
    const int tid = threadIdx.x;
#ifdef SSS_PHASES
    unsigned long long ph_t = __builtin_readcyclecounter();
    if (threadIdx.x == 0) atomicAdd(&g_sss[9], 1ull);
#endif
    // the query and its lists as the host resolved them (it holds term_ptr): ONE load latency where the first version walked
    // small_q -> q_off -> dterm -> term_ptr
    const unsigned char* const ent = p.small_tab + (size_t)(first + blockIdx.x) * p.small_stride;
    const SmallHdr hd = *reinterpret_cast<const SmallHdr*>(ent);
    const uint32_t q = hd.q, L = hd.n_lists, tot = hd.tot;
    TopK tk{cd_key, cd_doc, &sc32[0], &sc64[0], reinterpret_cast<float*>(&sc32[2]), 0ull, -INFINITY, (uint32_t)SCB};
    if ((uint32_t)tid < L) {
        const SmallList sl = reinterpret_cast<const SmallList*>(ent + sizeof(SmallHdr))[tid];
        const int field = (int)(sl.mf & 1u);
        l_rec[tid] = (uint64_t)((field ? p.t_rec : p.b_rec) + sl.start);
        l_w[tid] = (uint64_t)((field ? p.t_w : p.b_w) + sl.start);
        l_end[tid] = sl.end;                                              // inclusive prefix of the lengths (host)
        l_mf[tid] = sl.mf;
    }
    for (int i = tid; i < NS; i += ST) { hkey[i] = EMPTY; accT[i] = 0.0; accB[i] = 0.0; }
    if (tid == 0) { sc32[0] = 0u; sc32[1] = 0u; sc32[5] = 0u; sc64[0] = 0ull; sc64[1] = 0ull; sc64[2] = ~0ull; *tk.thr_f = -INFINITY; }
    __syncthreads();
    SSS_PH(0);

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
    SSS_PH(1);

    // ---- every candidate document: get_metadata.go:31-69.  A thread owns the slots tid, tid + 256, ...: the magnitudes of ALL of
    //      them are requested before any is used (one memory latency for the table, not one per slot)
    constexpr int PT = NS / ST;
    const double qmag = hd.qmag;
    const double* probs = (p.probs && p.prior) ? p.probs + (size_t)q * p.k_topics : nullptr;
    uint32_t e_doc[PT];
    uint64_t e_key[PT];
    uint32_t alive = 0;                     // bit j: candidate j is still in the running for the k-th place
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
                alive |= 1u << j;
            }
        }
    }
    if (alive) atomicAdd(&sc32[5], (uint32_t)__popc(alive));
    __syncthreads();
    SSS_PH(2);

    // ---- the k best: radix selection on {fkey(FinalRank), ~doc}, most significant byte first.  A pass counts the candidates still
    //      in the running by their next byte, finds the byte value d that holds the k-th best, admits everything above d, drops
    //      everything below; it stops as soon as what is admitted plus what is still running fits the (small) buffer that is sorted
    //      at the end.  FinalRanks of one query share their sign and exponent bytes and spread over the mantissa's: three or four
    //      passes of three barriers each, where the first version compacted a 512-entry buffer by bitonic sort once per overflow.
    {
        uint32_t limit = 64;
        while (limit < (uint32_t)p.k) limit <<= 1;                 // <= SCB (host: k <= SMALL_MAX_K)
        uint32_t k_rem = (uint32_t)p.k, n_alive = sc32[5];
        auto admit = [&](int j) __attribute__((always_inline)) {
            const uint32_t i = atomicAdd(tk.count, 1u);
            if (i < (uint32_t)SCB) { tk.key[i] = e_key[j]; tk.doc[i] = e_doc[j]; }      // (always: see the stop rule below)
        };
        // where the candidates' keys first differ (FinalRanks of one query share sign and exponent and mostly the first mantissa bits):
        // the passes start there instead of walking the constant bytes — OR and AND of all keys, two LDS atomics per wave
        int sh = 88;
        if (n_alive > limit) {
            uint64_t ko = 0ull, ka = ~0ull;
#pragma unroll
            for (int j = 0; j < PT; j++) if ((alive >> j) & 1u) { ko |= e_key[j]; ka &= e_key[j]; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { ko |= __shfl_xor(ko, o, 64); ka &= __shfl_xor(ka, o, 64); }
            if ((tid & 63) == 0) { atomicOr(reinterpret_cast<unsigned long long*>(&sc64[1]), ko); atomicAnd(reinterpret_cast<unsigned long long*>(&sc64[2]), ka); }
            __syncthreads();
            const uint64_t diff = sc64[1] ^ sc64[2];
            // highest differing key bit hb -> first byte = key bits [hb-7, hb]; equal keys throughout: straight to the doc id's bytes
            sh = diff ? max(0, 32 + (63 - __clzll((long long)diff)) - 7) : 24;
        }
        for (;; sh = sh >= 8 ? sh - 8 : (sh > 0 ? 0 : -1)) {
            // (admitted so far = k - k_rem: everything above the k-th candidate's bits so far)
            if ((uint32_t)p.k - k_rem + n_alive <= limit || sh < 0) {
#pragma unroll
                for (int j = 0; j < PT; j++) if ((alive >> j) & 1u) admit(j);
                break;
            }
            hist[tid] = 0u;                                         // ST == 256 bins
            __syncthreads();
#pragma unroll
            for (int j = 0; j < PT; j++) if ((alive >> j) & 1u) atomicAdd(&hist[sel_digit(e_key[j], e_doc[j], sh)], 1u);
            __syncthreads();
            if (tid < 64) {
                // lane l owns bins 4l .. 4l+3; `above` = candidates in the bins of the lanes above it
                const uint4 c = reinterpret_cast<const uint4*>(hist)[tid];
                const uint32_t s4 = c.x + c.y + c.z + c.w;
                uint32_t incl = s4;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t y = (uint32_t)__shfl_down((int)incl, o, 64);
                    if (tid + o < 64) incl += y;
                }
                uint32_t a = incl - s4;
                if (a < k_rem && k_rem <= a + s4) {                 // exactly one lane (n_alive >= k_rem here)
                    const uint32_t cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                    for (int bin = 3; bin >= 0; bin--) {
                        if (k_rem <= a + cc[bin]) { sc32[6] = 4u * (uint32_t)tid + (uint32_t)bin; sc32[7] = k_rem - a; sc32[8] = cc[bin]; break; }
                        a += cc[bin];
                    }
                }
            }
            __syncthreads();
            const uint32_t d = sc32[6];
            k_rem = sc32[7];
            n_alive = sc32[8];
#pragma unroll
            for (int j = 0; j < PT; j++) {
                if ((alive >> j) & 1u) {
                    const uint32_t dg = sel_digit(e_key[j], e_doc[j], sh);
                    if (dg > d) admit(j);
                    if (dg != d) alive &= ~(1u << j);
                }
            }
        }
    }
    SSS_PH(3);
    // ---- order what was admitted: every entry counts the entries that precede it ({key, doc} pairs are distinct, so the counts are
    //      the places).  All 256 threads take part: the entries are a power of two n2 <= 256, thread t counts for entry t % n2 over
    //      the (t / n2)-th part of the list, the parts' counts meet in LDS — three barriers and <= 256 / (256 / n2) broadcast reads per
    //      thread, eight in flight at a time, where a bitonic network over the same entries is 21 to 36 barrier-separated steps
    //      (16.6k cycles for 128 entries, measured; one thread per entry over the whole list: 8.5k)
    __syncthreads();
    const uint32_t n_adm = min(sc32[0], (uint32_t)SCB);
    uint32_t n2 = 64;
    while (n2 < n_adm) n2 <<= 1;
    hist[tid] = 0u;                             // (the selection is over: its histogram holds the places now)
    __syncthreads();
    uint64_t my_k = 0ull;
    uint32_t my_d = EMPTY;
    {
        const uint32_t i = (uint32_t)tid & (n2 - 1u), part = (uint32_t)tid / n2, parts = (uint32_t)ST / n2;
        if (i < n_adm) {
            my_k = cd_key[i];
            my_d = cd_doc[i];
            const uint32_t per = (n_adm + parts - 1u) / parts, j0 = part * per, j1 = min(n_adm, j0 + per);
            uint32_t cnt = 0;
            for (uint32_t j = j0; j < j1; j += 8) {
                uint64_t kk[8];
                uint32_t dd[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const uint32_t jj = min(j + (uint32_t)u, j1 - 1u); kk[u] = cd_key[jj]; dd[u] = cd_doc[jj]; }
#pragma unroll
                for (int u = 0; u < 8; u++) cnt += (j + (uint32_t)u < j1 && better(kk[u], dd[u], my_k, my_d)) ? 1u : 0u;
            }
            if (cnt) atomicAdd(&hist[i], cnt);
        }
    }
    __syncthreads();
    if ((uint32_t)tid < n_adm) {                // (tid < n_adm <= n2: this thread's entry is entry tid)
        const uint32_t place = hist[tid];
        if (place < (uint32_t)p.k) { cd_key[place] = my_k; cd_doc[place] = my_d; }
    }
    __syncthreads();
    const uint32_t n_out = min(n_adm, (uint32_t)p.k);
    SSS_PH(4);

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
        // (a batch that is pipelined over internal streams: the rows go to a block of their own, and k_small_copy moves them into the
        //  caller's buffer on the caller's stream — the hits must be complete in THAT stream's order)
        if (p.small_stage) p.small_stage[(size_t)(first + blockIdx.x) * p.k + i] = h;
        else p.hits[(size_t)q * p.k + i] = h;
    }
    if (tid == 0) {
        if (p.small_stage) p.small_stage_n[first + blockIdx.x] = (int32_t)n_out;
        else p.n_hits[q] = (int32_t)n_out;
    }
    SSS_PH(5);
}

// the staged rows of k_score_small's queries -> the caller's buffers (one workgroup per query; ss_hit rows are 40 bytes = 5 words of 8)
__global__ __launch_bounds__(128) void k_small_copy(ScoreParams p) {
    const uint32_t q = reinterpret_cast<const SmallHdr*>(p.small_tab + (size_t)blockIdx.x * p.small_stride)->q;
    const uint64_t* __restrict__ src = reinterpret_cast<const uint64_t*>(p.small_stage + (size_t)blockIdx.x * p.k);
    uint64_t* __restrict__ dst = reinterpret_cast<uint64_t*>(p.hits + (size_t)q * p.k);
    static_assert(sizeof(ss_hit) == 40, "ss_hit rows are copied as 8-byte words");
    for (uint32_t i = threadIdx.x; i < (uint32_t)p.k * 5u; i += 128) dst[i] = src[i];
    if (threadIdx.x == 0) p.n_hits[q] = p.small_stage_n[blockIdx.x];
}

}  // namespace

namespace ss {

uint32_t score_small_cap() { return CAP_B; }
uint32_t score_small_cap_a() { return CAP_A; }
int score_small_max_k() { return SMALL_MAX_K; }
int score_small_max_lists() { return SL; }
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

void launch_small_copy(const void* params, unsigned n_small, hipStream_t st) {
    const ScoreParams& p = *static_cast<const ScoreParams*>(params);
    if (n_small) hipLaunchKernelGGL(k_small_copy, dim3(n_small), dim3(128), 0, st, p);
}

void score_small_report() {
#ifdef SSS_PHASES
    unsigned long long h[10];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sss), sizeof(h)) == hipSuccess && h[9]) {
        const char* nm[6] = {"lists+clear", "postings->table", "magnitudes+scores", "select", "sort", "hits"};
        fprintf(stderr, "[k_score_small phases, %llu workgroups] cycles per workgroup:", h[9]);
        for (int i = 0; i < 6; i++) fprintf(stderr, " %s %.0f", nm[i], (double)h[i] / (double)h[9]);
        fprintf(stderr, "\n");
    }
#endif
}

}  // namespace ss
