// score_common.hpp — declarations shared by the scoring kernels (score.hip: workgroup-per-slice kernel, phrase search, merge,
// host side; score_wave.hip: wave-per-slice kernel).  Everything sits in an anonymous namespace: each translation unit
// gets its own copy and its own register allocation.
#pragma once
#include "index.hpp"
#include "order.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <new>

namespace {

#ifndef SS_TPB
#define SS_TPB 512
#endif
constexpr int TPB = SS_TPB;        // k_score_slices workgroup
#ifndef SS_TPB_M
#define SS_TPB_M 256
#endif
constexpr int TPB_M = SS_TPB_M;    // k_merge_topk workgroup
#ifndef SS_CAP
#define SS_CAP 1024
#endif
constexpr int CAP = SS_CAP;        // records per window (capacity)
#ifndef SS_TARGET_64THS
#define SS_TARGET_64THS 59
#endif
constexpr int TARGET = CAP * SS_TARGET_64THS / 64;   // planned records per window
constexpr int PPT = CAP / TPB;     // records per thread and window
#ifndef SS_SK_BITS
#define SS_SK_BITS 11
#endif
constexpr int SK = 1 << SS_SK_BITS;                  // slots of one filter table (three rotate)
#ifndef SS_PC
#define SS_PC 1024
#endif
constexpr int PC = SS_PC;          // pending survivor records = capacity of the exact stage (>= CAP)
constexpr int PPX = (PC + TPB - 1) / TPB;            // pending records per thread in a flush
#ifndef SS_HT
#define SS_HT 2048
#endif
constexpr int HT = SS_HT;          // exact-stage hash slots (load <= PC/HT)
constexpr int MAXL = 2 * SS_MAX_QUERY_TERMS + 4;     // (term, field) lists per query + 4 phrase result lists
#ifndef SS_TBL_CAP
#define SS_TBL_CAP 2048
#endif
constexpr int TBL_CAP = SS_TBL_CAP;                  // window-cursor table entries: (n_win+1) * L <= TBL_CAP
constexpr int OFF_CAP = TBL_CAP + TBL_CAP / 4 - (TBL_CAP + TBL_CAP / 4) % 8;   // window-offset table entries: n_win * OS <= OFF_CAP
constexpr int MAX_WIN = 1023;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint32_t NOREC = 0xFFFFu;                  // "no first record" in a packed ht_rec half
#ifndef SS_CB_MIN
#define SS_CB_MIN 256
#endif
#ifndef SS_SLICE_TARGET
#define SS_SLICE_TARGET 262144
#endif
constexpr uint64_t SLICE_TARGET = SS_SLICE_TARGET;
constexpr uint32_t MAX_SLICES_PER_Q = 256;
#ifndef SS_SLICE_MIN
#define SS_SLICE_MIN 8192
#endif
constexpr uint64_t SLICE_MIN = SS_SLICE_MIN;         // smallest adaptive slice (postings).  8192 since round 5 — it only binds for one or two queries at a time: one head
                                                     // query (1M postings, k = 50, host in / host out) 0.120 / 0.119 / 0.126 / 0.142 / 0.163 ms at 4k / 8k / 16k / 32k / 64k
constexpr int MAXCH = PC / 64;                       // chunked windows: most 64-record chunks per window (their records fit the exact stage)
constexpr int WAVES = TPB / 64;
constexpr int CPW = (MAXCH + WAVES - 1) / WAVES;     // chunk slots per wave and window
constexpr int LCH = 12;                              // queries with at most this many lists take the chunked window loop
constexpr int OSC = 16;                              // bytes per window row: LCH + 1 cumulative chunk counts, then the window's records / 8
constexpr int DEPTH = 3;                             // windows whose records are in flight or in registers (= number of filter tables)
// The filter sums in FIXED POINT: ds_add_u32 costs ~1/30 of ds_add_f32 on gfx950 (tools/micro/lds_ops.hip: 26 vs 880 ticks
// per wave-instruction).  A record's share is scaled so that the largest coefficient maps to FX_ONE units, rounded up, and
// clamped to FX_CLAMP; a slot that reaches FX_CLAMP counts as "unbounded" (its records survive).  PC records of
// FX_CLAMP each stay below 2^32: the sums never wrap.
constexpr uint32_t FX_ONE = 1u << 18;
constexpr uint32_t FX_CLAMP = SS_PC > 1024 ? 1u << 20 : 1u << 21;
static_assert((uint64_t)FX_CLAMP * SS_PC < (1ull << 32), "filter sums must not wrap");
constexpr int KTH_N = 11;                            // k'-th largest impact per term for k' = 2^0 .. 2^10
static_assert(DEPTH == 3, "ring slots, filter tables and survivor counters rotate together");
static_assert(PC >= CAP && PC < 0xFFFF, "the exact stage must hold one whole window; record indices are 16-bit");
static_assert(CAP % TPB == 0 && HT % 256 == 0, "sizes");

struct __attribute__((aligned(8))) Rec {   // scoring layout: 8 bytes per posting
    uint32_t doc;
    float imp;                               // float32 upper bound of w / magnitude(doc, field); 0 where the weight is 0
};

struct SliceDesc {
    uint32_t q;
    uint32_t dlo, dhi;   // doc range [dlo, dhi); dhi = 0xFFFFFFFF: to the end
    uint32_t pad;
};

struct ScoreParams {
    // per table: term_ptr, scoring records, the index's float32 weights and float64 magnitudes, k'-th largest impacts
    const uint64_t* t_ptr; const Rec* t_rec; const float* t_w; const double* t_mag; const float* t_kth;
    const uint64_t* b_ptr; const Rec* b_rec; const float* b_w; const double* b_mag; const float* b_kth;
    // combined lists of k_score_wave: per term the title and body postings merged by doc (field in bit 31 of the doc word),
    // c_w = the posting's float32 weight (a copy, in the combined order), c_skip[g] = doc of record 64*g (one entry per 512-byte block)
    const uint64_t* c_ptr; const Rec* c_rec; const float* c_w; const uint32_t* c_skip;
    uint32_t c_pad_block;     // blocks of real records (the block behind them is all padding)
    // positional postings (phrase search, retrieval/phrase.go): pos_ptr[P+1] into pos[] per table, or null
    const uint64_t* t_pos_ptr; const float* t_pos;
    const uint64_t* b_pos_ptr; const float* b_pos;
    // phrase part of the batch: ph_off[n_q+1] into ph_terms (all quoted phrases of a query concatenated,
    // main_retrieve.go:26), driver = index of the phrase's rarest term; outputs of k_phrase_match:
    // four doc-sorted record lists per query (body/title sums found via the driver's body/title postings)
    const uint32_t* ph_off; const uint32_t* ph_terms; const uint32_t* ph_drv;
    const uint32_t* x_off;     // [n_q+1] capacity offsets of the phrase result lists
    Rec* x_rec[4];             // 0: body sums (driver body pass), 1: title sums (driver body pass), 2: body (title pass), 3: title (title pass)
    float* x_w[4];             // the float32 weight sums themselves (phrase.go:59,69,73,83,90)
    uint32_t* x_cnt;           // [n_q][4]
    // k_phrase_match runs one workgroup per PART (<= PH_PART candidates of the driver's body or title list): parts of a query
    // write their matches compactly from the part's own offset, k_phrase_close then closes the gaps per query
    const uint4* ph_parts;     // [n_parts] {query, pass, first candidate (index inside the driver's list), candidates}
    const uint32_t* ph_pbase;  // [n_q+1] parts of query q: ph_parts[ph_pbase[q] .. ph_pbase[q+1])
    uint32_t* ph_pcnt;         // [n_parts][2] matches found by the part: body sums, title sums
    const double* prior;       // [n_docs][k_topics] or null
    int32_t k_topics;
    const uint32_t* q_off;     // [n_q+1] into dterm/dmult
    const uint32_t* dterm;     // distinct known terms per query, first-occurrence order
    const uint32_t* dmult;     // multiplicity of each
    const double* qmag;        // [n_q] sqrt(queryLength)
    const double* probs;       // [n_q][k_topics] or null
    const double* sqd_ub;      // [n_q] upper bound of sqd over all docs (when probs)
    const uint32_t* slice_base;// [n_q+1] (query order)
    const SliceDesc* slices;   // query order
    const uint32_t* order;     // launch order -> slice index (longest first)
    int32_t k;
    int32_t cb;                // candidate buffer entries (power of two >= 2k)
    int32_t cb_flat;           // k_merge_flat's own (>= cb)
    int32_t kth_j;             // smallest j with 2^j >= k
    int32_t exact_all;         // 1: the filter's assumptions do not hold for this call: every record goes to the exact stage
    uint64_t* so_key; uint32_t* so_doc; uint32_t* so_cnt;   // per slice top-k
    uint32_t* q_ticket;       // per query: slices that have handed in their list (fused merge); null = k_merge_topk runs as its own launch
    const uint32_t* merge_q;  // k_merge_flat: query of workgroup b
    const uint8_t* q_fast;    // [n_q] 1: the query is scored by k_score_wave
    uint32_t* qc_cnt;         // [n_q] candidates the wave slices of query q have appended (from entry slice_base[q] * k of so_key / so_doc)
    ss_hit* hits; int32_t* n_hits;
    const uint32_t* small_q;  // k_score_small's queries (they have no slices; q_fast[q] & 2)
    const unsigned char* small_tab;   // k_score_small: entry b (small_stride bytes) = SmallHdr + the query's SmallList rows
    uint32_t small_stride;
    const float* q_floor;             // experiment ("score.debug_floor"): a per-query lower bound of the k-th best FinalRank, or null
    ss_hit* small_stage;              // null: k_score_small writes row q of `hits`; else row b of this block (k_small_copy moves it on the caller's stream)
    int32_t* small_stage_n;
};

// k_score_small's view of a query, resolved by the host (which holds term_ptr)
#define SS_SMALL_MAX_LISTS 16
struct __attribute__((aligned(16))) SmallHdr { uint32_t q, n_lists, tot, pad; double qmag, pad2; };
struct __attribute__((aligned(16))) SmallList { uint64_t start; uint32_t end, mf; };    // first posting in its table; postings of lists 0..this one; multiplicity << 1 | field (1 = title)
static_assert(sizeof(SmallHdr) == 32 && sizeof(SmallList) == 16, "small-query table layout");

using ss::fkey;
using ss::funkey;
using ss::better;

// kernel arguments stay in scalar registers only while they are never indexed with a run-time value
__device__ __forceinline__ Rec* x_rec_of(const ScoreParams& p, int x) { return x == 0 ? p.x_rec[0] : x == 1 ? p.x_rec[1] : x == 2 ? p.x_rec[2] : p.x_rec[3]; }
__device__ __forceinline__ float* x_w_of(const ScoreParams& p, int x) { return x == 0 ? p.x_w[0] : x == 1 ? p.x_w[1] : x == 2 ? p.x_w[2] : p.x_w[3]; }

// Lists are addressed by the absolute address of their first record (regular posting lists and the
// per-query phrase result lists alike); explicit global address space keeps the loads global_load_*.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));   // native vector: loads as one global_load_dwordx2
typedef const u32x2 __attribute__((address_space(1)))* gptr_u2;
typedef const uint32_t __attribute__((address_space(1)))* gptr_u32;
typedef const float __attribute__((address_space(1)))* gptr_f32;
typedef const double __attribute__((address_space(1)))* gptr_f64;
__device__ __forceinline__ u32x2 load_rec(uint64_t list_addr, uint64_t idx) { return *(gptr_u2)(list_addr + idx * sizeof(Rec)); }
__device__ __forceinline__ uint32_t load_doc(uint64_t list_addr, uint64_t idx) { return *(gptr_u32)(list_addr + idx * sizeof(Rec)); }
__device__ __forceinline__ float load_w(uint64_t w_addr, uint64_t idx) { return *(gptr_f32)(w_addr + idx * sizeof(float)); }
__device__ __forceinline__ uint64_t lower_bound_rec(const Rec* __restrict__ a, uint64_t lo, uint64_t hi, uint32_t v) {
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid].doc < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint32_t lower_bound_addr(uint64_t list_addr, uint32_t lo, uint32_t hi, uint32_t v) {
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (load_doc(list_addr, mid) < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// Same result as lower_bound_addr with fewer DEPENDENT loads: every probe of a search in HBM costs a full
// memory latency (~1 us under load) and the slice set-up is nothing but such chains.  Doc ids of a list are
// close to uniform, so the position of v is guessed by interpolation and bracketed by two independent probes
// at guess -/+ sqrt(range) (one latency): the bracket shrinks n -> 2*sqrt(n) per step (262144 -> 1024 -> 64 ->
// 16), then bisection.  Any distribution stays correct: a probe on the wrong side still halves nothing but
// keeps the invariant, and after 4 steps plain bisection finishes.
__device__ __forceinline__ uint32_t lower_bound_interp(uint64_t list_addr, uint32_t lo, uint32_t hi, uint32_t v) {
    if (lo >= hi) return lo;
    uint32_t L = lo, H = hi - 1;
    uint32_t dl = load_doc(list_addr, L), dh = load_doc(list_addr, H);      // independent: one latency
    if (dl >= v) return lo;
    if (dh < v) return hi;
    // invariant: doc[L] = dl < v <= dh = doc[H]; the answer is in (L, H]
    for (int it = 0; it < 4 && H - L > 32; it++) {
        const uint32_t n = H - L;
        const float frac = (float)(v - dl) / (float)(dh - dl);             // dl < dh
        uint32_t g = L + (uint32_t)(frac * (float)n);
        const uint32_t dlt = (uint32_t)__fsqrt_rn((float)n) + 2;
        uint32_t a = g > L + dlt ? g - dlt : L + 1;                        // a in [L+1, H-1]
        a = min(a, H - 1);
        uint32_t b = min(a + 2 * dlt, H - 1);                              // b in [a, H-1]
        const uint32_t da = load_doc(list_addr, a), db = load_doc(list_addr, b);
        if (da >= v) { H = a; dh = da; }
        else if (db < v) { L = b; dl = db; }
        else { L = a; dl = da; H = b; dh = db; }
    }
    // finish 8-ary: seven independent probes per step (one latency) instead of three dependent ones
    while (H - L > 1) {
        const uint32_t step = (H - L + 7) >> 3;                             // >= 1
        uint32_t d[7];
#pragma unroll
        for (int i = 0; i < 7; i++) d[i] = load_doc(list_addr, min(L + step * (uint32_t)(i + 1), H - 1));
        uint32_t nl = L, nh = H;
#pragma unroll
        for (int i = 6; i >= 0; i--) {
            const uint32_t pos = min(L + step * (uint32_t)(i + 1), H - 1);
            if (d[i] >= v) nh = pos;                                        // doc[pos] >= v: the answer is at or before pos
        }
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const uint32_t pos = min(L + step * (uint32_t)(i + 1), H - 1);
            if (d[i] < v) nl = max(nl, pos);                                // doc[pos] < v: the answer is after pos
        }
        L = nl;                                                             // every probe is on one side or the other:
        H = nh;                                                             // the bracket shrinks to <= step
    }
    return H;
}

// lower_bound over records a[lo .. hi) (one posting list: shorter than 2^32) with the interpolation search
__device__ __forceinline__ uint64_t lower_bound_rec_interp(const Rec* __restrict__ a, uint64_t lo, uint64_t hi, uint32_t v) {
    return lo + lower_bound_interp((uint64_t)(a + lo), 0u, (uint32_t)(hi - lo), v);
}

// number of entries of the global u32 array a[lo, hi) (ascending) that are < v, as an offset from lo: interpolation
// steps with two independent probes each, then an 8-ary finish (lower_bound_interp of score_common.hpp for 4-byte entries)
// WIDE = false finishes with a plain binary search: fewer loads in all (k_merge_flat, which is bound by the number of scattered
// loads it issues); WIDE = true with fewer DEPENDENT ones (the set-up of a slice, which waits for every step)
template <bool WIDE = true>
__device__ __forceinline__ uint32_t skip_lower_bound(const uint32_t* __restrict__ a, uint32_t lo, uint32_t hi, uint32_t v) {
    if (lo >= hi) return 0;
    uint32_t L = lo, H = hi - 1;
    uint32_t dl = a[L], dh = a[H];
    if (dl >= v) return 0;
    if (dh < v) return hi - lo;
    for (int it = 0; it < 4 && H - L > 32; it++) {
        const uint32_t n = H - L;
        const float frac = (float)(v - dl) / (float)(dh - dl);
        uint32_t g = L + (uint32_t)(frac * (float)n);
        const uint32_t dlt = (uint32_t)__fsqrt_rn((float)n) + 2;
        uint32_t x = g > L + dlt ? g - dlt : L + 1;
        x = min(x, H - 1);
        const uint32_t y = min(x + 2 * dlt, H - 1);
        const uint32_t dx = a[x], dy = a[y];
        if (dx >= v) { H = x; dh = dx; }
        else if (dy < v) { L = y; dl = dy; }
        else { L = x; dl = dx; H = y; dh = dy; }
    }
    if (!WIDE) {
        while (H - L > 1) {
            const uint32_t mid = (L + H) >> 1;
            if (a[mid] < v) L = mid; else H = mid;
        }
        return H - lo;
    }
    while (H - L > 1) {
        const uint32_t step = (H - L + 7) >> 3;
        uint32_t d[7];
#pragma unroll
        for (int i = 0; i < 7; i++) d[i] = a[min(L + step * (uint32_t)(i + 1), H - 1)];
        uint32_t nl = L, nh = H;
#pragma unroll
        for (int i = 6; i >= 0; i--) {
            const uint32_t pos = min(L + step * (uint32_t)(i + 1), H - 1);
            if (d[i] >= v) nh = pos;
        }
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const uint32_t pos = min(L + step * (uint32_t)(i + 1), H - 1);
            if (d[i] < v) nl = max(nl, pos);
        }
        L = nl;
        H = nh;
    }
    return H - lo;
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

// Workgroup barrier that does NOT drain the vector-memory counter: the next window's records stay
// in flight across it (a __syncthreads() would emit s_waitcnt vmcnt(0), cdna_hip_programming.md §5
// "Pipelining across barriers").  LDS traffic is complete after lgkmcnt(0).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef SS_DIAG
// Diagnostic build only (make DIAG=1): event counts and s_memtime sums of wave 0 of every slice, printed by ss_scorer_destroy.
__device__ unsigned long long g_diag[24];
__device__ unsigned long long g_slice[4096][4];      // per launch index: {start (realtime 100 MHz), end, windows, records}
#define DIAG_ADD(i, v) do { if ((threadIdx.x) == 0) atomicAdd(&g_diag[i], (unsigned long long)(v)); } while (0)
#define DIAG_NOW(var) unsigned long long var; do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DIAG_ADD(i, v) do { } while (0)
#define DIAG_NOW(var) do { } while (0)
#endif
#ifdef SS_DIAG
#define DIAG_NOWX(var) DIAG_NOW(var)
#else
#define DIAG_NOWX(var) do { } while (0)
#endif
#if defined(SS_DIAG) && defined(SS_DIAG_LOOP)      // stamps INSIDE the window loop: they cost more than what they measure
#define DIAG_NOWL(var) DIAG_NOW(var)
#define DIAG_ADDL(i, v) DIAG_ADD(i, v)
#else
#define DIAG_NOWL(var) do { } while (0)
#define DIAG_ADDL(i, v) do { } while (0)
#endif

// ---- running top-k in LDS ------------------------------------------------------
struct TopK {
    uint64_t* key;    // [cb]
    uint32_t* doc;    // [cb]
    uint32_t* count;  // shared scalar (may run past cb while an overflow is pending)
    uint64_t* thr;    // shared scalar: admit keys >= thr
    float* thr_f;     // shared scalar: float lower bound of the threshold score (-inf: no threshold)
    uint64_t thr0;    // floor of the threshold known before any posting was read (0 = none)
    float thr0_f;
    uint32_t cb;
};

// Sort the candidate buffer best-first and keep the k best. All threads call.
// (An enumeration sort — every entry counts the entries that precede it, 2 barriers instead of 38 — measured
// 25 % slower end to end: 65k broadcast LDS reads cost more than the bitonic network's barriers.)
// (by value: a reference would force the struct into scratch memory for the out-of-line call)
#ifndef SS_COMPACT_PLACE
#define SS_COMPACT_PLACE 1      // 0: variant build, the network everywhere (A/B)
#endif
template <bool PLACE = true>       // PLACE = false: the network only (k_score_wave's one-wave buffer: the other path is dead code there, and it cost the kernel 3.5 %)
__device__ __forceinline__ void topk_compact_inl(const TopK& tk, int k) {
    DIAG_ADD(4, 1);
    lds_barrier();
    const uint32_t nthr = blockDim.x;
    const uint32_t n = min(*tk.count, tk.cb);
    uint32_t n2 = 64;
    while (n2 < n) n2 <<= 1;
    if (PLACE && n2 <= 128u && nthr >= 2u * n2) {
        // Up to 128 entries and at least two threads per entry (round 5): PLACEMENT BY COUNTING instead of the network — every entry counts
        // the entries that precede it ({key, doc} pairs are distinct: the counts are the places), `parts` neighbouring lanes share an
        // entry's comparisons and add their counts by shuffles, eight broadcast reads in flight per thread; two barriers
        // (tools/micro/compact_sorts.hip, 256 threads: 64 entries 2.7k cycles against 12k, 100 entries 7.5k against 15.5k; at 256
        // entries the quadratic count loses, 31k against 21k: the network stays there).
        uint32_t parts = 2;
        while (parts < 8u && nthr >= 2u * parts * n2) parts <<= 1;
        const uint32_t i = threadIdx.x / parts, part = threadIdx.x & (parts - 1u);
        const bool in = i < n;                                    // (threads beyond parts * n2 have i >= n2 >= n)
        const uint64_t my_k = in ? tk.key[i] : 0ull;
        const uint32_t my_d = in ? tk.doc[i] : EMPTY;
        uint32_t cnt = 0;
        if (in) {
            const uint32_t per = (n + parts - 1u) / parts, j0 = part * per, j1 = min(n, j0 + per);
            for (uint32_t j = j0; j < j1; j += 4) {                // (four in flight: eight cost k_merge_flat its eighth wave per SIMD)
                uint64_t kk[4];
                uint32_t dd[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const uint32_t jj = min(j + (uint32_t)u, j1 - 1u); kk[u] = tk.key[jj]; dd[u] = tk.doc[jj]; }
#pragma unroll
                for (int u = 0; u < 4; u++) cnt += (j + (uint32_t)u < j1 && better(kk[u], dd[u], my_k, my_d)) ? 1u : 0u;
            }
        }
        for (uint32_t o = 1; o < parts; o <<= 1) cnt += (uint32_t)__shfl_xor((int)cnt, (int)o, 64);
        lds_barrier();                                            // every entry has been read
        if (in && part == 0u) { tk.key[cnt] = my_k; tk.doc[cnt] = my_d; }
        lds_barrier();
    } else {
    for (uint32_t i = n + threadIdx.x; i < n2; i += nthr) { tk.key[i] = 0ull; tk.doc[i] = EMPTY; }   // worst sentinels
    lds_barrier();
    for (uint32_t size = 2; size <= n2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t i = threadIdx.x; i < (n2 >> 1); i += nthr) {
                const uint32_t lo = 2 * i - (i & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool desc = ((lo & size) == 0);     // this run sorted best-first
                const uint64_t ka = tk.key[lo], kb = tk.key[hi];
                const uint32_t da = tk.doc[lo], db = tk.doc[hi];
                const bool swap = desc ? better(kb, db, ka, da) : better(ka, da, kb, db);
                if (swap) { tk.key[lo] = kb; tk.key[hi] = ka; tk.doc[lo] = db; tk.doc[hi] = da; }
            }
            lds_barrier();
        }
    }
    }
    if (threadIdx.x == 0) {
        const uint32_t keep = min(n, (uint32_t)k);
        *tk.count = keep;
        const bool full = keep == (uint32_t)k;
        uint64_t t = full ? tk.key[k - 1] : 0ull;
        float tf = -INFINITY;
        if (full && t != 0ull) tf = __double2float_rd(funkey(t));
        if (tk.thr0 > t) { t = tk.thr0; tf = tk.thr0_f; }
        else if (tk.thr0_f > tf) tf = tk.thr0_f;
        *tk.thr = t;
        *tk.thr_f = tf;
    }
    lds_barrier();
}
// Cut the candidate buffer to its k best WITHOUT ordering them (round 5): what a compaction is for is the new threshold — the k-th best
// key — and room in the buffer; the order only matters where hits are written, and that is the merge's own sort.  Radix selection on the
// 96-bit key {key, ~doc} (the order of `better`): OR and AND of the keys give the first bit in which they differ, a pass counts the
// entries still running by the next eight bits (256-bin histogram, one wave scans it from the top), keeps what lies above the bin of the
// k-th, drops what lies below, and the passes end when the entries still running are exactly the ones still needed; the kept entries
// move to the front, the smallest kept key is the new threshold.  One entry per thread (cb <= blockDim.x): ~13 barriers, where the
// bitonic network over 256 entries is 36 barrier-separated steps of ~585 cycles (tools/micro/compact_sorts.hip: 21k cycles; 2.5
// compactions per slice of a tail batch = a third of the slice).  Same survivors and same threshold as topk_compact_inl.
__device__ __forceinline__ uint32_t sel_digit96(uint64_t key, uint32_t doc, int sh) {
    if (sh >= 32) return (uint32_t)(key >> (sh - 32)) & 255u;
    const uint64_t w = ((key & 0xFFFFFFFFull) << 32 | (uint64_t)(~doc)) >> sh;
    return (uint32_t)w & 255u;
}
// sel: [256 + 8] words, sel64: [4] — LDS scratch of the caller (kept out of TopK: the struct lives in k_score_wave's registers)
__device__ __forceinline__ void topk_select_inl(const TopK& tk, int k, uint32_t* sel, uint64_t* sel64) {
    uint32_t* const hist = sel;             // [256]
    uint32_t* const sw = sel + 256;         // [8]: bin, k_rem, n_alive, kept counter
    uint64_t* const s64 = sel64;            // [4]: OR, AND, min kept key
    lds_barrier();
    const uint32_t tid = threadIdx.x;
    const uint32_t n = min(*tk.count, tk.cb);
    const bool mine = tid < n;              // (cb <= blockDim.x: entry tid is this thread's)
    const uint64_t e_key = mine ? tk.key[tid] : 0ull;
    const uint32_t e_doc = mine ? tk.doc[tid] : EMPTY;
    if (tid == 0) { s64[0] = 0ull; s64[1] = ~0ull; s64[2] = ~0ull; sw[3] = 0u; }
    lds_barrier();
    bool keep = mine;                       // n <= k: everything stays
    if (n > (uint32_t)k) {
        uint64_t ko = e_key, ka = mine ? e_key : ~0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ko |= __shfl_xor(ko, o, 64); ka &= __shfl_xor(ka, o, 64); }
        if ((tid & 63u) == 0u) { atomicOr(reinterpret_cast<unsigned long long*>(&s64[0]), ko); atomicAnd(reinterpret_cast<unsigned long long*>(&s64[1]), ka); }
        lds_barrier();
        const uint64_t diff = s64[0] ^ s64[1];
        int sh = diff ? max(0, 32 + (63 - __clzll((long long)diff)) - 7) : 24;
        uint32_t k_rem = (uint32_t)k, n_alive = n;
        bool alive = mine;
        keep = false;
        for (;; sh = sh >= 8 ? sh - 8 : (sh > 0 ? 0 : -1)) {
            if (n_alive == k_rem || sh < 0) { keep = keep || alive; break; }      // (sh < 0: equal {key, doc} pairs — cannot happen, docs are distinct)
            if (tid < 256u) hist[tid] = 0u;
            lds_barrier();
            const uint32_t dg = sel_digit96(e_key, e_doc, sh);
            if (alive) atomicAdd(&hist[dg], 1u);
            lds_barrier();
            if (tid < 64u) {
                const uint4 c = reinterpret_cast<const uint4*>(hist)[tid];
                const uint32_t s4 = c.x + c.y + c.z + c.w;
                uint32_t incl = s4;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t y = (uint32_t)__shfl_down((int)incl, o, 64);
                    if (tid + (uint32_t)o < 64u) incl += y;
                }
                uint32_t a = incl - s4;
                if (a < k_rem && k_rem <= a + s4) {
                    const uint32_t cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                    for (int bin = 3; bin >= 0; bin--) {
                        if (k_rem <= a + cc[bin]) { sw[0] = 4u * tid + (uint32_t)bin; sw[1] = k_rem - a; sw[2] = cc[bin]; break; }
                        a += cc[bin];
                    }
                }
            }
            lds_barrier();
            const uint32_t d = sw[0];
            k_rem = sw[1];
            n_alive = sw[2];
            if (alive) {
                if (dg > d) keep = true;
                if (dg != d) alive = false;
            }
        }
    }
    // the kept entries to the front (every thread holds its entry in registers), their smallest key = the new threshold
    lds_barrier();
    if (keep) {
        const uint32_t pos = atomicAdd(&sw[3], 1u);
        tk.key[pos] = e_key;
        tk.doc[pos] = e_doc;
        atomicMin(reinterpret_cast<unsigned long long*>(&s64[2]), e_key);
    }
    lds_barrier();
    if (tid == 0) {
        const uint32_t keepn = min(n, (uint32_t)k);
        *tk.count = keepn;
        const bool full = keepn == (uint32_t)k;
        uint64_t t = full ? s64[2] : 0ull;
        float tf = -INFINITY;
        if (full && t != 0ull) tf = __double2float_rd(funkey(t));
        if (tk.thr0 > t) { t = tk.thr0; tf = tk.thr0_f; }
        else if (tk.thr0_f > tf) tf = tk.thr0_f;
        *tk.thr = t;
        *tk.thr_f = tf;
    }
    lds_barrier();
}
// the called form: the buffer is in LDS in every caller, and the function is told so (pointer arguments are generic otherwise:
// flat_load / flat_store with vmcnt(0) lgkmcnt(0) behind every step of the sort)
__device__ void topk_compact(const TopK tk, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_assume(__builtin_amdgcn_is_shared(tk.key));
    __builtin_assume(__builtin_amdgcn_is_shared(tk.doc));
    __builtin_assume(__builtin_amdgcn_is_shared(tk.count));
    __builtin_assume(__builtin_amdgcn_is_shared(tk.thr));
    __builtin_assume(__builtin_amdgcn_is_shared(tk.thr_f));
#endif
    topk_compact_inl<SS_COMPACT_PLACE != 0>(tk, k);
}
// the network only: k_merge_flat, whose 40 registers are what lets four of its waves sit beside two k_score_wave waves on a SIMD (with the
// placement path it needs 46 -> 48: three, and config 3's pipelined period went from 0.342 to 0.361 ms)
__device__ void topk_compact_net(const TopK tk, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_assume(__builtin_amdgcn_is_shared(tk.key));
    __builtin_assume(__builtin_amdgcn_is_shared(tk.doc));
    __builtin_assume(__builtin_amdgcn_is_shared(tk.count));
    __builtin_assume(__builtin_amdgcn_is_shared(tk.thr));
    __builtin_assume(__builtin_amdgcn_is_shared(tk.thr_f));
#endif
    topk_compact_inl<false>(tk, k);
}
// cut to the k best, unordered (radix selection) where one entry per thread fits; else the bitonic network
__device__ void topk_cut(const TopK tk, int k, uint32_t* sel, uint64_t* sel64) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_assume(__builtin_amdgcn_is_shared(tk.key));
    __builtin_assume(__builtin_amdgcn_is_shared(tk.doc));
    __builtin_assume(__builtin_amdgcn_is_shared(tk.count));
    __builtin_assume(__builtin_amdgcn_is_shared(tk.thr));
    __builtin_assume(__builtin_amdgcn_is_shared(tk.thr_f));
    __builtin_assume(__builtin_amdgcn_is_shared(sel));
    __builtin_assume(__builtin_amdgcn_is_shared(sel64));
#endif
    if (tk.cb <= blockDim.x) topk_select_inl(tk, k, sel, sel64);
    else topk_compact_inl(tk, k);
}

struct SliceQuery {      // per-query constants of the exact stage
    double qmag, sqd_ub;
    float qmag_f, sqd_ub_f;
    const double* probs;
};

// the running threshold in the filter's units, rounded down: a slot below it cannot hold a doc of the top-k.
// thr_f = -inf (no threshold yet) -> 0: everything survives; never above FX_CLAMP: clamped shares always survive.
__device__ __forceinline__ uint32_t fx_threshold(float thr_f, float r_ub, float fx_scale) {
    const float t = (thr_f - r_ub) * fx_scale * (1.0f - 0x1p-20f);
    return t > 0.0f ? min((uint32_t)t, FX_CLAMP) : 0u;
}

// a record's share of FinalRank in fixed-point units, rounded up, clamped (NaN -> 1: only reachable with the filter off)
__device__ __forceinline__ uint32_t fx_share(float imp, float coef_fx) {
    return min((uint32_t)(imp * coef_fx), FX_CLAMP - 1u) + 1u;
}

}  // namespace
