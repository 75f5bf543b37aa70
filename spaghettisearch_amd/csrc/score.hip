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
// Device design (HBM-bound streaming of posting records):
//   * scoring layout: one 16-byte record per posting {doc u32, w f32, mag f64}; the document's
//     field magnitude travels with the posting, so a candidate needs NO random gather (a separate
//     magnitude table costs a 64-byte line per candidate: 8x the posting bytes at config 3).
//     A magnitude only matters when the doc has a posting of that field among the query terms
//     (0/(m*q) is 0 or NaN->0 for every m), so nothing else is needed.
//   * the host plans (it keeps df per term): duplicates -> multiplicities, unknown terms dropped,
//     each query's doc range cut into slices (size follows the batch: ~1.5x the batch's postings per
//     resident workgroup slot, 16k..256k postings), longest first; one workgroup per (query, slice).
//   * k_score_slices first builds the slice's whole window plan in LDS (window j = docs between the
//     j-th and (j+1)-th cut of the longest list; every other list's cursor at each cut by an
//     interpolating search), then walks the windows (<= CAP postings, every posting of a doc in one
//     window); window j+1's records are loaded (one coalesced 16-byte load per lane) while window j
//     is processed.  Per window: every record parks {float64(w)*multiplicity, magnitude} at its own
//     index (stride-1 store), claims its doc's slot in an LDS hash table (32-bit CAS, linear
//     probing) and the slot's per-field "first record" word (32-bit CAS); later records of the
//     same (doc, field) add into the first one's addend (float32 addends in float64 are exact, so
//     order is irrelevant) — single-record docs need no float64 atomic.  After one barrier the table
//     is scanned stride-1: a float pre-filter drops almost every doc against the running
//     threshold, survivors are scored exactly in float64 and go to the running top-k (threshold
//     filter + bitonic compaction).  With a PageRank blend the 128-byte prior row of a doc is only
//     fetched if an upper bound of its score can still enter the top-k.
//   * k_merge_topk: one workgroup per query merges its slices' top-k lists, re-derives
//     title/body/pagerank of the k winners by binary search and writes ss_hit rows.
//   Ties: ascending doc id (Q10); NaN finals last.
#include "index.hpp"
#include "order.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>

namespace {

#ifndef SS_TPB
#define SS_TPB 512
#endif
constexpr int TPB = SS_TPB;        // k_score_slices workgroup
constexpr int TPB_M = 256;         // k_merge_topk workgroup
#ifndef SS_CAP
#define SS_CAP 1024
#endif
constexpr int CAP = SS_CAP;        // postings per window (capacity)
#ifndef SS_TARGET_64THS
#define SS_TARGET_64THS 59
#endif
constexpr int TARGET = CAP * SS_TARGET_64THS / 64;   // planned postings per window (944 of 1024; measured best of 832..1008: beyond 960 oversize windows start to cost more than the fewer windows save)
constexpr int PPT = CAP / TPB;     // records per thread
#ifndef SS_HT
#define SS_HT 2048
#endif
constexpr int HT = SS_HT;           // hash slots (multiple of 256; 2048: load factor <= 0.5, measured best of 1536..3072)
constexpr int EPT = (HT + TPB - 1) / TPB;   // hash entries per thread in the scan
constexpr int MAXL = 2 * SS_MAX_QUERY_TERMS + 4;   // (term, field) lists per query + 4 phrase result lists
#ifndef SS_TBL_CAP
#define SS_TBL_CAP 4096
#endif
constexpr int TBL_CAP = SS_TBL_CAP;      // window-cursor table entries: (n_win+1) * L <= TBL_CAP
constexpr int OFF_CAP = TBL_CAP + TBL_CAP / 4 - (TBL_CAP + TBL_CAP / 4) % 8;   // window-offset table entries: n_win * OS <= OFF_CAP
constexpr int MAX_WIN = 1023;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
#ifndef SS_CB_MIN
#define SS_CB_MIN 512
#endif
#ifndef SS_SLICE_TARGET
#define SS_SLICE_TARGET 262144
#endif
constexpr uint64_t SLICE_TARGET = SS_SLICE_TARGET;
constexpr uint32_t MAX_SLICES_PER_Q = 256;
#ifndef SS_SLICE_MIN
#define SS_SLICE_MIN 16384
#endif
constexpr uint64_t SLICE_MIN = SS_SLICE_MIN;     // smallest adaptive slice (postings)

struct __attribute__((aligned(16))) Post {
    uint32_t doc;
    float w;
    double mag;
};

struct SliceDesc {
    uint32_t q;
    uint32_t dlo, dhi;   // doc range [dlo, dhi); dhi = 0xFFFFFFFF: to the end
    uint32_t pad;
};

struct ScoreParams {
    const uint64_t* t_ptr; const Post* t_post;
    const uint64_t* b_ptr; const Post* b_post;
    // positional postings (phrase search, retrieval/phrase.go): pos_ptr[P+1] into pos[] per table, or null
    const uint64_t* t_pos_ptr; const float* t_pos;
    const uint64_t* b_pos_ptr; const float* b_pos;
    // phrase part of the batch: ph_off[n_q+1] into ph_terms (all quoted phrases of a query concatenated,
    // main_retrieve.go:26), driver = index of the phrase's rarest term; outputs of k_phrase_match:
    // four doc-sorted record lists per query (body/title sums found via the driver's body/title postings)
    const uint32_t* ph_off; const uint32_t* ph_terms; const uint32_t* ph_drv;
    const uint32_t* x_off;     // [n_q+1] capacity offsets of the phrase result lists
    Post* x_list[4];           // 0: body sums (driver body pass), 1: title sums (driver body pass), 2: body (title pass), 3: title (title pass)
    uint32_t* x_cnt;           // [n_q][4]
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
    uint64_t* so_key; uint32_t* so_doc; uint32_t* so_cnt;   // per slice top-k
    ss_hit* hits; int32_t* n_hits;
};

using ss::fkey;
using ss::funkey;
using ss::better;

__device__ __forceinline__ uint64_t lower_bound_post(const Post* __restrict__ a, uint64_t lo, uint64_t hi, uint32_t v) {
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid].doc < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// Lists are addressed by the absolute address of their first record (regular posting lists and the
// per-query phrase result lists alike); explicit global address space keeps the loads global_load_*.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // native vector: loads as one global_load_dwordx4
typedef const u32x4 __attribute__((address_space(1)))* gptr_u4;
typedef const uint32_t __attribute__((address_space(1)))* gptr_u32;
__device__ __forceinline__ u32x4 load_rec(uint64_t list_addr, uint64_t idx) { return *(gptr_u4)(list_addr + idx * sizeof(Post)); }
__device__ __forceinline__ uint32_t load_doc(uint64_t list_addr, uint64_t idx) { return *(gptr_u32)(list_addr + idx * sizeof(Post)); }
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
    while (H - L > 1) {
        const uint32_t mid = L + ((H - L) >> 1);
        if (load_doc(list_addr, mid) < v) L = mid; else H = mid;
    }
    return H;
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

#ifdef SS_DIAG
// Diagnostic build only (make DIAG=1): per-phase s_memtime sums of one wave per block, printed by ss_scorer_destroy.
__device__ unsigned long long g_stamps[16];
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(var) do { } while (0)
#endif

// Workgroup barrier that does NOT drain the vector-memory counter: the next window's records stay
// in flight across it (a __syncthreads() would emit s_waitcnt vmcnt(0), cdna_hip_programming.md §5
// "Pipelining across barriers").  LDS traffic is complete after lgkmcnt(0).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- running top-k in LDS ------------------------------------------------------
struct TopK {
    uint64_t* key;    // [cb]
    uint32_t* doc;    // [cb]
    uint32_t* count;  // shared scalar (may run past cb while an overflow is pending)
    uint64_t* thr;    // shared scalar: admit keys >= thr
    float* thr_f;     // shared scalar: float lower bound of the threshold score (-inf while fewer than k)
    uint32_t cb;
};

// Sort the candidate buffer best-first and keep the k best. All threads call.
// (An enumeration sort — every entry counts the entries that precede it, 2 barriers instead of 38 — measured
// 25 % slower end to end: 65k broadcast LDS reads cost more than the bitonic network's barriers.)
__device__ void topk_compact(const TopK& tk, int k) {
    lds_barrier();
    const uint32_t nthr = blockDim.x;
    const uint32_t n = min(*tk.count, tk.cb);
    uint32_t n2 = 64;
    while (n2 < n) n2 <<= 1;
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
    if (threadIdx.x == 0) {
        const uint32_t keep = min(n, (uint32_t)k);
        *tk.count = keep;
        const bool full = keep == (uint32_t)k;
        *tk.thr = full ? tk.key[k - 1] : 0ull;
        float tf = -INFINITY;
        if (full && tk.key[k - 1] != 0ull) tf = __double2float_rd(funkey(tk.key[k - 1]));
        *tk.thr_f = tf;
    }
    lds_barrier();
}

// ---- K4: score one (query, doc-range slice) --------------------------------------
#ifndef SS_WGS_PER_CU
#define SS_WGS_PER_CU 2
#endif
__global__ __launch_bounds__(TPB, (TPB / 256) * SS_WGS_PER_CU) void k_score_slices(ScoreParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // Every record of the window parks {its addend float64(w)*multiplicity, its doc's field magnitude} at its own index
    // (a stride-1 store).  A table slot names, per field, the FIRST record of that doc and field (one 32-bit CAS each);
    // the rare later records of the same (doc, field) add their addend into the first one's.  So the 95 % of docs with
    // one record per field cost no float64 atomic (the dearest LDS operation) and there is no per-slot
    // accumulator array.
    double2* s_rec = reinterpret_cast<double2*>(smem);                 // [CAP] {addend (summed: BodyRank/TitleRank of the doc), magnitude}
    uint64_t* l_base = reinterpret_cast<uint64_t*>(s_rec + CAP);       // [MAXL] address of the list's first record
    double* l_mult = reinterpret_cast<double*>(l_base + MAXL);         // [MAXL]
    uint64_t* sc64 = reinterpret_cast<uint64_t*>(l_mult + MAXL);       // [2]: thr
    uint64_t* cd_key = sc64 + 2;                                       // [cb]
    uint32_t* cd_doc = reinterpret_cast<uint32_t*>(cd_key + p.cb);     // [cb]
    uint32_t* ht_key = cd_doc + p.cb;                                  // [HT]
    uint32_t* tbl = ht_key + HT;                                       // [TBL_CAP] cursor of list l at the start of window j: tbl[j*L+l]
    uint32_t* l_field = tbl + TBL_CAP;                                 // [MAXL]
    uint32_t* f_cur = l_field + MAXL;                                  // [MAXL] oversize fallback: sub-window start
    uint32_t* f_nxt = f_cur + MAXL;                                    // [MAXL] oversize fallback: sub-window end
    uint32_t* sc32 = f_nxt + MAXL;                                     // [8] scalars
    uint32_t* ht_rec = sc32 + 8;                                       // [HT][2] first body record, first title record of the slot's doc (EMPTY = none)
    uint16_t* off = reinterpret_cast<uint16_t*>(ht_rec + 2 * HT);      // [OFF_CAP] offset of list l inside window j: off[j*OS+l], count at [j*OS+OS-1]

    uint32_t& cand_count = sc32[0];
    uint32_t& overflow = sc32[1];
    float* thr_f_s = reinterpret_cast<float*>(&sc32[2]);
    TopK tk{cd_key, cd_doc, &sc32[0], &sc64[0], thr_f_s, (uint32_t)p.cb};

    const int tid = threadIdx.x;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)ts; (void)acc;
    STAMP(ts[7]);
    const uint32_t slice_id = p.order[blockIdx.x];
    const SliceDesc sd = p.slices[slice_id];
    const uint32_t q = sd.q;
    const uint32_t t0 = p.q_off[q], nd = p.q_off[q + 1] - t0;
    const bool has_phrase = p.ph_off && p.ph_off[q + 1] > p.ph_off[q];
    const int L = (int)(2 * nd) + (has_phrase ? 4 : 0);
    const double qmag = p.qmag[q];
    const double* probs = (p.probs && p.prior) ? p.probs + (size_t)q * p.k_topics : nullptr;
    const double sqd_ub = probs ? p.sqd_ub[q] : 0.0;
    const float sqd_ub_f = probs ? __double2float_ru(sqd_ub) : 0.0f;
    const float qmag_f = (float)qmag;

    for (int i = tid; i < HT; i += TPB) { ht_key[i] = EMPTY; ht_rec[2 * i] = EMPTY; ht_rec[2 * i + 1] = EMPTY; }
    if (tid == 0) { cand_count = 0; overflow = 0; sc64[0] = 0ull; *thr_f_s = -INFINITY; }

    // ---- slice set-up: where every list enters and leaves the slice's doc range ----
    uint32_t* t_lo = tbl;                   // row 0 of the table
    uint32_t* t_hi = f_nxt;                 // parked here until n_win is known
    if (tid < L) {
        const int field = tid & 1;                     // 0 = body, 1 = title (also for the phrase lists)
        uint64_t addr;
        uint32_t len;
        double mult;
        if (tid < (int)(2 * nd)) {
            const uint32_t term = p.dterm[t0 + (tid >> 1)];
            const uint64_t* ptr = field ? p.t_ptr : p.b_ptr;
            const uint64_t p0 = ptr[term];
            addr = (uint64_t)((field ? p.t_post : p.b_post) + p0);
            len = (uint32_t)(ptr[term + 1] - p0);
            mult = (double)p.dmult[t0 + (tid >> 1)];
        } else {
            // phrase contributions are appended once, after the terms (main_retrieve.go:73-78)
            const int x = tid - (int)(2 * nd);
            addr = (uint64_t)(p.x_list[x] + p.x_off[q]);
            len = p.x_cnt[(size_t)q * 4 + x];
            mult = 1.0;
        }
        l_base[tid] = addr;
        l_mult[tid] = mult;
        l_field[tid] = field;
        f_cur[tid] = len;
    }
    __syncthreads();
    if (tid < 2 * L) {                      // the two bounds of a list by two threads: independent search chains
        const int l = tid >> 1;
        if (tid & 1) t_hi[l] = sd.dhi == 0xFFFFFFFFu ? f_cur[l] : lower_bound_interp(l_base[l], 0, f_cur[l], sd.dhi);
        else t_lo[l] = sd.dlo == 0 ? 0u : lower_bound_interp(l_base[l], 0, f_cur[l], sd.dlo);
    }
    __syncthreads();
    // every thread: total, driver (longest list in the slice), number of windows
    uint32_t tot = 0, drv = 0, drv_len = 0;
    for (int l = 0; l < L; l++) {
        const uint32_t len = t_hi[l] - t_lo[l];
        tot += len;
        if (len > drv_len) { drv_len = len; drv = l; }
    }
    int n_win = 0;
    if (tot) {
        n_win = (int)((tot + TARGET - 1) / TARGET);
        n_win = min(n_win, min(MAX_WIN, min(TBL_CAP / L - 1, OFF_CAP / (L <= 6 ? 8 : L + 1))));
        n_win = max(n_win, 1);
    }
    const int OS = L <= 6 ? 8 : L + 1;      // row stride of `off`; the window's record count sits in the row's last entry
    // window j covers docs [b_j, b_{j+1}), b_j = doc of the driver's record at j/n_win of its run:
    // all cursors are known up front (no per-window serial planning; windows fill evenly because
    // the other lists simply contribute whatever falls into the driver's doc range)
    const uint64_t drv_addr = l_base[drv];
    for (int idx = tid; idx < (n_win - 1) * L; idx += TPB) {
        const int j = idx / L + 1, l = idx - (j - 1) * L;
        const uint32_t b = load_doc(drv_addr, t_lo[drv] + (uint64_t)j * drv_len / n_win);
        tbl[j * L + l] = lower_bound_interp(l_base[l], t_lo[l], t_hi[l], b);
    }
    __syncthreads();
    if (tid < L && n_win) tbl[n_win * L + tid] = t_hi[tid];
    __syncthreads();
    for (int j = tid; j < n_win; j += TPB) {
        uint32_t run = 0;
        for (int l = 0; l < OS - 1; l++) {
            off[j * OS + l] = l < L ? (uint16_t)min(run, 0xFFFFu) : (uint16_t)0xFFFFu;
            if (l < L) run += tbl[(j + 1) * L + l] - tbl[j * L + l];
        }
        off[j * OS + OS - 1] = (uint16_t)min(run, 0xFFFFu);      // > CAP: oversize window (fallback below)
    }
    __syncthreads();

    // records of one window: raw 16-byte vectors (one global_load_dwordx4 per posting)
    u32x4 rec[PPT];
    uint32_t rl[PPT];
    auto load_window = [&](int j) {
        const uint32_t* tbl_j = tbl + j * L;
        if (L <= 6) {
            // the whole offset row in one 16-byte LDS read; list of record i by comparisons in registers
            const uint4 o = *reinterpret_cast<const uint4*>(off + j * 8);
            const uint32_t o1 = o.x >> 16, o2 = o.y & 0xFFFFu, o3 = o.y >> 16, o4 = o.z & 0xFFFFu, o5 = o.z >> 16, n = o.w >> 16;
#pragma unroll
            for (int r = 0; r < PPT; r++) {
                const uint32_t i = tid + r * TPB;
                rl[r] = EMPTY;
                if (i < n) {
                    uint32_t l = 0, ol = 0;     // padding entries are 0xFFFF: never <= i
                    if (i >= o1) { l = 1; ol = o1; }
                    if (i >= o2) { l = 2; ol = o2; }
                    if (i >= o3) { l = 3; ol = o3; }
                    if (i >= o4) { l = 4; ol = o4; }
                    if (i >= o5) { l = 5; ol = o5; }
                    rec[r] = load_rec(l_base[l], (uint64_t)tbl_j[l] + (i - ol));
                    rl[r] = l;
                }
            }
        } else {
            const uint16_t* off_j = off + j * OS;
            const uint32_t n = off_j[OS - 1];
#pragma unroll
            for (int r = 0; r < PPT; r++) {
                const uint32_t i = tid + r * TPB;
                rl[r] = EMPTY;
                if (i < n) {
                    int lo = 0, hi = L;            // largest l with off[l] <= i
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (off_j[mid] <= i) lo = mid; else hi = mid;
                    }
                    rec[r] = load_rec(l_base[lo], (uint64_t)tbl_j[lo] + (i - off_j[lo]));
                    rl[r] = lo;
                }
            }
        }
    };
    // accumulate the loaded records per doc (main_retrieve.go:61-69,170-187); float32 addends in float64: exact.
    // (probing a thread's records one after the other measured 27 % faster than issuing their CAS together)
    auto insert_records = [&]() {
        uint32_t h[PPT];
        bool pend[PPT];
#pragma unroll
        for (int r = 0; r < PPT; r++) {
            pend[r] = rl[r] != EMPTY;
            h[r] = (((rec[r].x * 2654435761u) >> 20) * (uint32_t)(HT / 256)) >> 4;      // 12 hash bits -> [0, HT)
        }
#pragma unroll
        for (int r = 0; r < PPT; r++) {
            if (pend[r]) {
                for (;;) {
                    const uint32_t prev = atomicCAS(&ht_key[h[r]], EMPTY, rec[r].x);
                    if (prev == EMPTY || prev == rec[r].x) break;
                    h[r] = h[r] + 1 == HT ? 0 : h[r] + 1;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < PPT; r++) {
            const uint32_t l = rl[r];
            if (l != EMPTY) {
                const uint32_t i = tid + r * TPB;
                const uint32_t field = l & 1;          // 0 = body, 1 = title
                const double v = (double)__uint_as_float(rec[r].y) * l_mult[l];
                s_rec[i] = make_double2(v, __hiloint2double((int)rec[r].w, (int)rec[r].z));
                asm volatile("" ::: "memory");         // program order: parked before it can be named (LDS runs a wave's operations in order)
                const uint32_t first = atomicCAS(&ht_rec[2 * h[r] + field], EMPTY, i);
                // float32 addends in float64: exact, so the order of these rare adds does not matter
                if (first != EMPTY) atomicAdd(&s_rec[first].x, v);
            }
        }
    };
    // score every touched doc (get_metadata.go:31-69), reset the table, filter into the running top-k.
    // All LDS reads of a thread's entries are issued together; empty slots read as zero sums and drop out.
    auto scan_table = [&]() {
        uint64_t e_key[EPT];
        uint32_t e_doc[EPT];
        double2 tb[EPT];
        uint32_t ib[EPT], it[EPT];
        const uint64_t thr0 = *tk.thr;
        const float thr_f = *tk.thr_f;
#pragma unroll
        for (int r = 0; r < EPT; r++) {
            const int hh = tid + r * TPB;
            e_doc[r] = EMPTY;
            ib[r] = EMPTY;
            it[r] = EMPTY;
            if (HT % TPB == 0 || hh < HT) {
                e_doc[r] = ht_key[hh];
                const uint2 fr = *reinterpret_cast<const uint2*>(ht_rec + 2 * hh);
                ib[r] = fr.x;
                it[r] = fr.y;
            }
        }
        double mt[EPT], mb[EPT];
#pragma unroll
        for (int r = 0; r < EPT; r++) {
            const int hh = tid + r * TPB;
            // no posting of a field => its sum is 0 and 0/(m*q) is 0 (or NaN -> 0) for every m
            double2 rb = make_double2(0.0, 1.0), rt = make_double2(0.0, 1.0);
            if (ib[r] != EMPTY) rb = s_rec[ib[r]];
            if (it[r] != EMPTY) rt = s_rec[it[r]];
            tb[r] = make_double2(rb.x, rt.x);
            mb[r] = rb.y;
            mt[r] = rt.y;
            if (HT % TPB == 0 || hh < HT) {
                ht_key[hh] = EMPTY;
                *reinterpret_cast<uint2*>(ht_rec + 2 * hh) = make_uint2(EMPTY, EMPTY);
            }
        }
#pragma unroll
        for (int r = 0; r < EPT; r++) {
            e_key[r] = 0;
            if (e_doc[r] != EMPTY) {
                const double B = tb[r].x, T = tb[r].y;
                // cheap float estimate first: almost every doc is far below the running threshold.
                // Skipped only if the estimate, with a 1e-4 relative margin, is clearly below; anything
                // non-finite falls through to the exact path.
                // v_rcp_f32 (1 ulp) instead of an IEEE division: the 1e-4 margin below covers it; 0*inf = NaN falls through
                const float ea = 38.0f * ((float)T * __builtin_amdgcn_rcpf((float)mt[r] * qmag_f)), eb = 29.0f * ((float)B * __builtin_amdgcn_rcpf((float)mb[r] * qmag_f)),
                            ec = 33.0f * sqd_ub_f;
                if ((ea + eb + ec) + (fabsf(ea) + fabsf(eb) + fabsf(ec)) * 1e-4f + 1e-30f < thr_f) { e_doc[r] = EMPTY; continue; }
                double title, body, fin;
                if (probs) {
                    // the prior row (128 B) is only fetched if the doc can still make the top-k:
                    // every operation of final_rank is monotone in sqd, so sqd_ub bounds the score
                    final_rank(T, B, mt[r], mb[r], qmag, sqd_ub, title, body, fin);
                    if (fkey(fin) >= thr0 || fin != fin) final_rank(T, B, mt[r], mb[r], qmag, topic_dot(p.prior, probs, p.k_topics, e_doc[r]), title, body, fin);
                    else e_doc[r] = EMPTY;
                } else {
                    final_rank(T, B, mt[r], mb[r], qmag, 0.0, title, body, fin);
                }
                e_key[r] = fkey(fin);
            }
        }
        // threshold filter into the candidate buffer; overflow -> compact and retry
        for (;;) {
            const uint64_t thr = *tk.thr;
#pragma unroll
            for (int r = 0; r < EPT; r++) {
                if (e_doc[r] != EMPTY) {
                    if (e_key[r] >= thr) {
                        const uint32_t i = atomicAdd(tk.count, 1u);
                        if (i < tk.cb) { tk.key[i] = e_key[r]; tk.doc[i] = e_doc[r]; e_doc[r] = EMPTY; }
                        else overflow = 1;
                    } else {
                        e_doc[r] = EMPTY;
                    }
                }
            }
            lds_barrier();
            if (!overflow) break;
            topk_compact(tk, p.k);         // raises thr; count back to <= k
            if (tid == 0) overflow = 0;
            lds_barrier();
        }
    };

    if (n_win > 0 && off[OS - 1] <= CAP) load_window(0);
    STAMP(ts[0]);
    acc[6] += ts[0] - ts[7];
    for (int j = 0; j < n_win; j++) {
        const uint32_t n = off[j * OS + OS - 1];
        if (n <= CAP) {
            STAMP(ts[0]);
            insert_records();                                   // waits for window j's records
            STAMP(ts[1]);
            lds_barrier();
            STAMP(ts[2]);
            if (j + 1 < n_win && off[(j + 1) * OS + OS - 1] <= CAP) load_window(j + 1);   // in flight during the scan
            STAMP(ts[3]);
            scan_table();
            STAMP(ts[4]);
            acc[0] += ts[1] - ts[0]; acc[1] += ts[2] - ts[1]; acc[2] += ts[3] - ts[2]; acc[3] += ts[4] - ts[3]; acc[7] += 1;
        } else {
            // ---- oversize window (a list is locally much denser than planned): bisect its doc range
            //      until a piece fits, process the piece, continue.  Rare; no prefetch here. ----
            if (tid < L) f_cur[tid] = tbl[j * L + tid];
            uint32_t flo = j == 0 ? sd.dlo : load_doc(drv_addr, tbl[j * L + drv]);
            const uint32_t fend = j + 1 == n_win ? sd.dhi : load_doc(drv_addr, tbl[(j + 1) * L + drv]);
            __syncthreads();
            for (;;) {
                uint32_t fhi = fend;
                uint32_t cnt;
                for (;;) {
                    if (tid < L) {
                        f_nxt[tid] = fhi == fend ? tbl[(j + 1) * L + tid]
                                                 : lower_bound_addr(l_base[tid], f_cur[tid], tbl[(j + 1) * L + tid], fhi);
                    }
                    __syncthreads();
                    cnt = 0;
                    for (int l = 0; l < L; l++) cnt += f_nxt[l] - f_cur[l];
                    if (cnt <= (uint32_t)CAP) break;
                    // a single doc has at most L <= 128 postings, so the bisection ends
                    fhi = flo + (uint32_t)(((uint64_t)fhi - flo) >> 1);
                    __syncthreads();
                }
#pragma unroll
                for (int r = 0; r < PPT; r++) {
                    const uint32_t i = tid + r * TPB;
                    rl[r] = EMPTY;
                    if (i < cnt) {
                        uint32_t run = 0;
                        int l = 0;
                        for (; l < L; l++) {
                            const uint32_t len = f_nxt[l] - f_cur[l];
                            if (i < run + len) break;
                            run += len;
                        }
                        rec[r] = load_rec(l_base[l], (uint64_t)f_cur[l] + (i - run));
                        rl[r] = l;
                    }
                }
                insert_records();
                lds_barrier();
                scan_table();
                bool done = true;
                for (int l = 0; l < L; l++) done = done && f_nxt[l] == tbl[(j + 1) * L + l];
                __syncthreads();
                if (done) break;
                if (tid < L) f_cur[tid] = f_nxt[tid];
                flo = fhi;
                __syncthreads();
            }
            if (j + 1 < n_win && off[(j + 1) * OS + OS - 1] <= CAP) load_window(j + 1);
        }
    }

    topk_compact(tk, p.k);
    const uint32_t n_out = cand_count;
    for (uint32_t i = tid; i < n_out; i += TPB) {
        p.so_key[(size_t)slice_id * p.k + i] = cd_key[i];
        p.so_doc[(size_t)slice_id * p.k + i] = cd_doc[i];
    }
    if (tid == 0) p.so_cnt[slice_id] = n_out;
#ifdef SS_DIAG
    STAMP(ts[1]);
    if ((tid & 63) == 0 && (tid >> 6) == 3) {
        for (int i = 0; i < 8; i++) atomicAdd(&g_stamps[i], acc[i]);
        atomicAdd(&g_stamps[8], ts[1] - ts[7]);
        atomicAdd(&g_stamps[9], 1ull);
    }
#endif
}

// ---- K6: quoted-phrase matching (retrieval/phrase.go:11-170, util.go:162-203) --------------------
// One workgroup per query with a phrase.  Candidates = the docs of the phrase's rarest term (its body
// postings, then its title-only postings); a thread takes one candidate doc and
//   * looks the doc up in the body and title lists of every phrase term (binary search): the doc must have
//     an entry for EVERY term position, in body or title (phrase.go:63);
//   * per field: every term must be present in that field, the float32 weights are summed in phrase
//     order (phrase.go:59,69,73,83,90) and the position lists, shifted by the term's index
//     (getPosTerm :145,:157: listPos[i] -= float32(pos)), must have a common value (intersect,
//     util.go:179-203; bit-exact float32 equality as in the reference);
//   * matches leave as 16-byte records {doc, float32 sum, field magnitude} in doc order (ordered
//     compaction), i.e. as ordinary doc-sorted lists that k_score_slices merges like any term list
//     (main_retrieve.go:73-78).
constexpr int PH_TPB = 256;
constexpr int PH_MAX = 16;       // phrase terms (SS_MAX_PHRASE_TERMS)

__device__ __forceinline__ bool positions_chain(const uint64_t* const* pos_ptr2, const float* const* pos2, int field,
                                                const int32_t* my_post, int m) {
    // S = A_0; S = S ∩ (A_i - i) for i = 1..m-1; non-empty?
    const uint64_t* pp = pos_ptr2[field];
    const float* ps = pos2[field];
    const uint64_t a_beg = pp[my_post[0]], a_end = pp[my_post[0] + 1];
    for (uint64_t a = a_beg; a < a_end; a++) {
        const float v = ps[a] - 0.0f;
        bool alive = true;
        for (int i = 1; i < m && alive; i++) {
            const uint64_t b_beg = pp[my_post[i * PH_TPB]], b_end = pp[my_post[i * PH_TPB] + 1];
            bool found = false;
            for (uint64_t b = b_beg; b < b_end; b++)
                if (ps[b] - (float)i == v) { found = true; break; }
            alive = found;
        }
        if (alive) return true;
    }
    return false;
}

__global__ __launch_bounds__(PH_TPB) void k_phrase_match(ScoreParams p) {
    __shared__ int32_t s_pb[PH_MAX * PH_TPB];     // body posting index of term i for this thread's doc, -1 = none
    __shared__ int32_t s_pt[PH_MAX * PH_TPB];
    __shared__ uint32_t s_wave[2][PH_TPB / 64];
    __shared__ uint32_t s_base[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t q = blockIdx.x;
    const uint32_t f0 = p.ph_off[q], m = p.ph_off[q + 1] - f0;
    uint32_t* cnt = p.x_cnt + (size_t)q * 4;
    if (tid < 4) cnt[tid] = 0;
    if (m == 0 || p.ph_drv[q] == 0xFFFFFFFFu) return;       // no phrase / a phrase word unknown to the index
    const uint32_t drv_term = p.ph_terms[f0 + p.ph_drv[q]];
    const uint64_t* pos_ptr2[2] = {p.b_pos_ptr, p.t_pos_ptr};
    const float* pos2[2] = {p.b_pos, p.t_pos};
    int32_t* my_pb = s_pb + tid;
    int32_t* my_pt = s_pt + tid;
    if (tid < 2) s_base[tid] = 0;
    __syncthreads();

    for (int pass = 0; pass < 2; pass++) {
        // pass 0: driver's body postings; pass 1: driver's title postings whose doc has no driver body posting
        const uint64_t c0 = pass == 0 ? p.b_ptr[drv_term] : p.t_ptr[drv_term];
        const uint64_t c1 = pass == 0 ? p.b_ptr[drv_term + 1] : p.t_ptr[drv_term + 1];
        const Post* cpost = pass == 0 ? p.b_post : p.t_post;
        if (tid < 2) s_base[tid] = 0;
        __syncthreads();
        for (uint64_t cb = c0; cb < c1; cb += PH_TPB) {
            const uint64_t ci = cb + tid;
            bool body_ok = false, title_ok = false;
            float sum_b = 0.0f, sum_t = 0.0f;
            uint32_t d = 0;
            if (ci < c1) {
                d = cpost[ci].doc;
                bool all = true, body_all = true, title_all = true;
                if (pass == 1) {
                    const uint64_t b0 = p.b_ptr[drv_term], b1 = p.b_ptr[drv_term + 1];
                    const uint64_t pos = lower_bound_post(p.b_post, b0, b1, d);
                    if (pos < b1 && p.b_post[pos].doc == d) all = false;       // already handled in pass 0
                }
                for (uint32_t i = 0; i < m && all; i++) {
                    const uint32_t term = p.ph_terms[f0 + i];
                    const uint64_t b0 = p.b_ptr[term], b1 = p.b_ptr[term + 1];
                    const uint64_t t0 = p.t_ptr[term], t1 = p.t_ptr[term + 1];
                    const uint64_t pb = lower_bound_post(p.b_post, b0, b1, d);
                    const uint64_t pt = lower_bound_post(p.t_post, t0, t1, d);
                    const bool hb = pb < b1 && p.b_post[pb].doc == d, ht = pt < t1 && p.t_post[pt].doc == d;
                    my_pb[i * PH_TPB] = hb ? (int32_t)pb : -1;
                    my_pt[i * PH_TPB] = ht ? (int32_t)pt : -1;
                    if (!hb && !ht) all = false;                      // phrase.go:63
                    if (hb) sum_b += p.b_post[pb].w; else body_all = false;     // phrase.go:69,80-84
                    if (ht) sum_t += p.t_post[pt].w; else title_all = false;    // phrase.go:73,87-91
                }
                if (all) {
                    if (body_all) body_ok = positions_chain(pos_ptr2, pos2, 0, my_pb, (int)m);
                    if (title_all) title_ok = positions_chain(pos_ptr2, pos2, 1, my_pt, (int)m);
                }
            }
            // ordered compaction of the matches of this chunk (keeps the lists doc-sorted)
            const unsigned long long mb = __ballot(body_ok), mt = __ballot(title_ok);
            if (lane == 0) { s_wave[0][wave] = (uint32_t)__popcll(mb); s_wave[1][wave] = (uint32_t)__popcll(mt); }
            __syncthreads();
            uint32_t ob = s_base[0], ot = s_base[1];
            for (int w2 = 0; w2 < wave; w2++) { ob += s_wave[0][w2]; ot += s_wave[1][w2]; }
            const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
            if (body_ok) {
                Post r;
                r.doc = d; r.w = sum_b; r.mag = p.b_post[my_pb[0]].mag;
                p.x_list[pass == 0 ? 0 : 2][p.x_off[q] + ob + (uint32_t)__popcll(mb & below)] = r;
            }
            if (title_ok) {
                Post r;
                r.doc = d; r.w = sum_t; r.mag = p.t_post[my_pt[0]].mag;
                p.x_list[pass == 0 ? 1 : 3][p.x_off[q] + ot + (uint32_t)__popcll(mt & below)] = r;
            }
            __syncthreads();
            if (tid == 0) {
                uint32_t tb = 0, tt = 0;
                for (int w2 = 0; w2 < PH_TPB / 64; w2++) { tb += s_wave[0][w2]; tt += s_wave[1][w2]; }
                s_base[0] += tb;
                s_base[1] += tt;
            }
            __syncthreads();
        }
        if (tid == 0) { cnt[pass == 0 ? 0 : 2] = s_base[0]; cnt[pass == 0 ? 1 : 3] = s_base[1]; }
        __syncthreads();
    }
}

size_t score_lds_bytes(int cb) {
    return (size_t)CAP * 16 + (size_t)MAXL * 8 * 2 + 2 * 8 + (size_t)cb * 12 +
           ((size_t)3 * HT + TBL_CAP + 3 * MAXL + 8) * 4 + (size_t)OFF_CAP * 2 + 16;
}

// ---- K5: merge a query's slices, explain the winners ------------------------------
__global__ __launch_bounds__(TPB_M) void k_merge_topk(ScoreParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* accT = reinterpret_cast<double*>(smem);                    // [k]
    double* accB = accT + p.k;                                         // [k]
    double* mgT = accB + p.k;                                          // [k]
    double* mgB = mgT + p.k;                                           // [k]
    uint64_t* cd_key = reinterpret_cast<uint64_t*>(mgB + p.k);         // [cb]
    uint64_t* sc64 = cd_key + p.cb;                                    // [1]
    uint32_t* cd_doc = reinterpret_cast<uint32_t*>(sc64 + 1);          // [cb]
    uint32_t* sc32 = cd_doc + p.cb;                                    // [4]
    TopK tk{cd_key, cd_doc, &sc32[0], &sc64[0], reinterpret_cast<float*>(&sc32[2]), (uint32_t)p.cb};
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const int k = p.k;
    if (tid == 0) { sc32[0] = 0; sc64[0] = 0ull; }
    __syncthreads();
    for (uint32_t s = p.slice_base[q]; s < p.slice_base[q + 1]; s++) {
        const uint32_t n = p.so_cnt[s];                  // <= k, and cb >= 2k: room after a compaction
        if (sc32[0] + n > tk.cb) topk_compact(tk, k);
        __syncthreads();
        const uint64_t thr = *tk.thr;
        for (uint32_t i = tid; i < n; i += TPB_M) {
            const uint64_t key = p.so_key[(size_t)s * k + i];
            if (key >= thr) {
                const uint32_t j = atomicAdd(tk.count, 1u);
                tk.key[j] = key;
                tk.doc[j] = p.so_doc[(size_t)s * k + i];
            }
        }
        __syncthreads();
    }
    topk_compact(tk, k);
    const uint32_t n_out = sc32[0];

    // explain: TitleRank/BodyRank of the winners, re-derived from the posting lists
    for (uint32_t i = tid; i < n_out; i += TPB_M) { accT[i] = 0.0; accB[i] = 0.0; mgT[i] = 1.0; mgB[i] = 1.0; }
    __syncthreads();
    const uint32_t t0 = p.q_off[q], nd = p.q_off[q + 1] - t0;
    const bool has_phrase = p.ph_off && p.ph_off[q + 1] > p.ph_off[q];
    const uint32_t L = 2 * nd + (has_phrase ? 4u : 0u);
    for (uint32_t task = tid; task < n_out * L; task += TPB_M) {
        const uint32_t i = task / L, l = task % L;
        const int field = l & 1;
        uint64_t addr;
        uint32_t len;
        double mult;
        if (l < 2 * nd) {
            const uint32_t term = p.dterm[t0 + (l >> 1)];
            const uint64_t* ptr = field ? p.t_ptr : p.b_ptr;
            addr = (uint64_t)((field ? p.t_post : p.b_post) + ptr[term]);
            len = (uint32_t)(ptr[term + 1] - ptr[term]);
            mult = (double)p.dmult[t0 + (l >> 1)];
        } else {
            addr = (uint64_t)(p.x_list[l - 2 * nd] + p.x_off[q]);
            len = p.x_cnt[(size_t)q * 4 + (l - 2 * nd)];
            mult = 1.0;
        }
        const uint32_t d = cd_doc[i];
        const uint32_t pos = lower_bound_interp(addr, 0, len, d);
        if (pos < len) {
            const u32x4 raw = load_rec(addr, pos);
            if (raw.x == d) {
                const double v = (double)__uint_as_float(raw.y) * mult;
                const double mag = __hiloint2double((int)raw.w, (int)raw.z);
                if (field) { atomicAdd(&accT[i], v); mgT[i] = mag; }
                else { atomicAdd(&accB[i], v); mgB[i] = mag; }
            }
        }
    }
    __syncthreads();
    const double qmag = p.qmag[q];
    const double* probs = (p.probs && p.prior) ? p.probs + (size_t)q * p.k_topics : nullptr;
    for (uint32_t i = tid; i < (uint32_t)k; i += TPB_M) {
        ss_hit h;
        h.doc = 0; h._pad = 0; h.title = 0.0; h.body = 0.0; h.pagerank = 0.0; h.final = 0.0;
        if (i < n_out) {
            const uint32_t d = cd_doc[i];
            const double sqd = probs ? topic_dot(p.prior, probs, p.k_topics, d) : 0.0;
            double title, body, fin;
            final_rank(accT[i], accB[i], mgT[i], mgB[i], qmag, sqd, title, body, fin);
            h.doc = d; h.title = title; h.body = body; h.pagerank = sqd; h.final = fin;
        }
        p.hits[(size_t)q * k + i] = h;
    }
    if (tid == 0) p.n_hits[q] = (int32_t)n_out;
}

size_t merge_lds_bytes(int k, int cb) { return (size_t)k * 32 + (size_t)cb * 12 + 8 + 16 + 16; }

// scoring layout: {doc, w, mag[doc]} per posting
__global__ void k_pack_posts(const uint32_t* __restrict__ doc, const float* __restrict__ w, const double* __restrict__ mag,
                             uint64_t n, Post* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        Post r;
        r.doc = doc[i];
        r.w = w[i];
        r.mag = mag[r.doc];
        out[i] = r;
    }
}
// rank [K][N] topic-major -> prior [N][K] node-major
__global__ void k_transpose_prior(const double* __restrict__ in, uint64_t n, int K, double* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * (uint64_t)K) return;
    const uint64_t doc = i / K;
    const int t = (int)(i % K);
    out[i] = in[(uint64_t)t * n + doc];
}
// per-topic max / min of the prior (for the score upper bound), read from the topic-major input: one contiguous,
// coalesced row per topic (blockIdx.y); a wave reduces with DPP shuffles, one atomic pair per wave
__global__ void k_prior_extrema(const double* __restrict__ rank_topic_major, uint64_t n, unsigned long long* __restrict__ mx,
                                unsigned long long* __restrict__ mn) {
    const int t = blockIdx.y;
    const double* row = rank_topic_major + (uint64_t)t * n;
    double a = -INFINITY, b = INFINITY;
    bool nan = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const double v = row[i];
        if (v != v) nan = true;
        a = fmax(a, v);
        b = fmin(b, v);
    }
    if (nan) { a = INFINITY; b = -INFINITY; }
    unsigned long long ka = fkey(a), kb = fkey(b);
    for (int m = 32; m > 0; m >>= 1) {
        const unsigned long long oa = __shfl_xor(ka, m), ob = __shfl_xor(kb, m);
        ka = oa > ka ? oa : ka;
        kb = ob < kb ? ob : kb;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&mx[t], ka);
        atomicMin(&mn[t], kb);
    }
}

double unkey(uint64_t k) {
    const uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    double d;
    std::memcpy(&d, &b, 8);
    return d;
}

}  // namespace

struct ss_scorer {
    ss_ctx* ctx = nullptr;
    ss_index* title = nullptr;
    ss_index* body = nullptr;
    uint64_t n_docs = 0, n_terms = 0;
    ss::DevBuf<Post> t_post, b_post;
    ss::DevBuf<double> prior;
    std::vector<double> prior_max, prior_min;   // per topic
    int k_topics = 0;
    int lds_attr = 0;
    // per-call workspaces, grow-only (no hipMalloc/hipFree on the steady-state query path)
    ss::DevBuf<unsigned char> d_plan;
    // pinned staging for the plan, double-buffered: a call that returns results in device memory does not wait
    // for the GPU, so the next call plans (and fills the other buffer) while this one's copy and kernels run
    unsigned char* h_plan[2] = {nullptr, nullptr};
    size_t h_plan_cap[2] = {0, 0};
    hipEvent_t plan_ev[2] = {nullptr, nullptr}; // recorded after the H2D copy of the buffer
    bool plan_ev_pending[2] = {false, false};
    int plan_turn = 0;
    ss::DevBuf<double> d_probs;
    ss::DevBuf<Post> d_x[4];                    // phrase result lists
    ss::DevBuf<uint32_t> d_xcnt;
    ss::DevBuf<uint64_t> d_so_key;
    ss::DevBuf<uint32_t> d_so_doc, d_so_cnt;
    ss::DevBuf<ss_hit> d_hits;
    ss::DevBuf<int32_t> d_nhits;
    ~ss_scorer() {
        for (int i = 0; i < 2; i++) {
            if (h_plan[i]) (void)hipHostFree(h_plan[i]);
            if (plan_ev[i]) (void)hipEventDestroy(plan_ev[i]);
        }
    }
};

namespace {
template <typename T>
hipError_t ensure(ss::DevBuf<T>& b, size_t n) {
    if (b.p && b.n >= n) return hipSuccess;
    return b.alloc(n + n / 2 + 16);
}
size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
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
    SS_HIP(ctx, s->t_post.alloc(title->n_post));
    SS_HIP(ctx, s->b_post.alloc(body->n_post));
    if (title->n_post)
        hipLaunchKernelGGL(k_pack_posts, dim3(std::min<unsigned>(ss::div_up(title->n_post, 256), 16384u)), dim3(256), 0, ctx->stream,
                           (const uint32_t*)title->post_doc.p, (const float*)title->post_w.p, (const double*)title->mag.p,
                           title->n_post, s->t_post.p);
    if (body->n_post)
        hipLaunchKernelGGL(k_pack_posts, dim3(std::min<unsigned>(ss::div_up(body->n_post, 256), 16384u)), dim3(256), 0, ctx->stream,
                           (const uint32_t*)body->post_doc.p, (const float*)body->post_w.p, (const double*)body->mag.p,
                           body->n_post, s->b_post.p);
    SS_HIP(ctx, hipGetLastError());
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
#ifdef SS_DIAG
    {
        unsigned long long h[16];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h)) == hipSuccess) {
            const char* names[10] = {"insert(+wait)", "barrier", "issue_loads", "scan+admit", "-", "-", "slice_init", "windows", "block_total", "blocks"};
            fprintf(stderr, "[ss diag] k_score_slices wave 3 cycles:");
            for (int i = 0; i < 10; i++) fprintf(stderr, " %s=%llu", names[i], h[i]);
            fprintf(stderr, "\n");
        }
    }
#endif
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
    ss::DevBuf<unsigned long long> ext;
    SS_HIP(ctx, tmp.alloc(n));
    SS_HIP(ctx, ext.alloc(2 * (size_t)k_topics));
    SS_HIP(ctx, hipMemcpyAsync(tmp.p, rank, n * sizeof(double), hipMemcpyDefault, ctx->stream));
    SS_HIP(ctx, s->prior.alloc(n));
    hipLaunchKernelGGL(k_transpose_prior, dim3(ss::div_up(n, 256)), dim3(256), 0, ctx->stream, (const double*)tmp.p, s->n_docs,
                       k_topics, s->prior.p);
    SS_HIP(ctx, hipMemsetAsync(ext.p, 0x00, k_topics * sizeof(unsigned long long), ctx->stream));                 // max keys
    SS_HIP(ctx, hipMemsetAsync(ext.p + k_topics, 0xFF, k_topics * sizeof(unsigned long long), ctx->stream));      // min keys
    hipLaunchKernelGGL(k_prior_extrema, dim3(std::min<unsigned>(ss::div_up(s->n_docs, 256), 1024u), k_topics), dim3(256), 0,
                       ctx->stream, (const double*)tmp.p, s->n_docs, ext.p, ext.p + k_topics);
    SS_HIP(ctx, hipGetLastError());
    std::vector<unsigned long long> h_ext(2 * (size_t)k_topics);
    SS_HIP(ctx, hipMemcpyAsync(h_ext.data(), ext.p, h_ext.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->prior_max.resize(k_topics);
    s->prior_min.resize(k_topics);
    for (int t = 0; t < k_topics; t++) {
        s->prior_max[t] = unkey(h_ext[t]);
        s->prior_min[t] = unkey(h_ext[k_topics + t]);
    }
    s->k_topics = k_topics;
    return SS_OK;
}

static int32_t score_impl(ss_scorer* s, int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms, const uint32_t* p_ptr,
                          const uint32_t* p_terms, const int32_t* query_len, const double* topic_probs, int32_t k,
                          ss_hit* hits_out, int32_t* n_hits_out);

int32_t ss_score_topk(ss_scorer* s, int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms, const int32_t* query_len,
                      const double* topic_probs, int32_t k, ss_hit* hits_out, int32_t* n_hits_out) {
    return score_impl(s, n_q, q_ptr, q_terms, nullptr, nullptr, query_len, topic_probs, k, hits_out, n_hits_out);
}

int32_t ss_score_topk_phrase(ss_scorer* s, int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms, const uint32_t* p_ptr,
                             const uint32_t* p_terms, const int32_t* query_len, const double* topic_probs, int32_t k,
                             ss_hit* hits_out, int32_t* n_hits_out) {
    if (s && !p_ptr) return s->ctx->fail(SS_ERR_INVALID, "ss_score_topk_phrase: p_ptr is NULL");
    return score_impl(s, n_q, q_ptr, q_terms, p_ptr, p_terms, query_len, topic_probs, k, hits_out, n_hits_out);
}

static int32_t score_impl(ss_scorer* s, int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms, const uint32_t* p_ptr,
                          const uint32_t* p_terms, const int32_t* query_len, const double* topic_probs, int32_t k,
                          ss_hit* hits_out, int32_t* n_hits_out) {
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
    // phrase part (retrieval/phrase.go): tokens of all quoted phrases of a query, concatenated
    std::vector<uint32_t> h_pptr(n_q + 1, 0), h_pterms, h_pdrv(n_q, 0xFFFFFFFFu), h_xoff(n_q + 1, 0);
    if (p_ptr) {
        if (!s->title->pos_ptr.p || !s->body->pos_ptr.p)
            return ctx->fail(SS_ERR_STATE, "ss_score_topk_phrase: positional postings not loaded (ss_index_set_positions on both tables)");
        SS_HIP(ctx, hipMemcpy(h_pptr.data(), p_ptr, (n_q + 1) * sizeof(uint32_t), hipMemcpyDefault));
        for (int q = 0; q < n_q; q++)
            if (h_pptr[q + 1] < h_pptr[q]) return ctx->fail(SS_ERR_INVALID, "ss_score_topk_phrase: p_ptr not non-decreasing");
        h_pterms.resize(h_pptr[n_q]);
        if (h_pptr[n_q]) {
            if (!p_terms) return ctx->fail(SS_ERR_INVALID, "ss_score_topk_phrase: p_terms is NULL");
            SS_HIP(ctx, hipMemcpy(h_pterms.data(), p_terms, h_pterms.size() * sizeof(uint32_t), hipMemcpyDefault));
        }
    }
    std::vector<int32_t> h_qlen(n_q);
    if (query_len) SS_HIP(ctx, hipMemcpy(h_qlen.data(), query_len, n_q * sizeof(int32_t), hipMemcpyDefault));
    else for (int q = 0; q < n_q; q++)     // len(queryTokenised)+len(phraseTokenised), main_retrieve.go:90
        h_qlen[q] = (int32_t)(h_qptr[q + 1] - h_qptr[q]) + (int32_t)(h_pptr[q + 1] - h_pptr[q]);
    std::vector<double> h_probs;
    const int K = s->k_topics;
    if (topic_probs) {
        h_probs.resize((size_t)n_q * K);
        SS_HIP(ctx, hipMemcpy(h_probs.data(), topic_probs, h_probs.size() * sizeof(double), hipMemcpyDefault));
    }

    const std::vector<uint64_t>& tp = s->title->h_term_ptr;
    const std::vector<uint64_t>& bp = s->body->h_term_ptr;
    bool any_phrase = false;
    for (int q = 0; q < n_q && p_ptr; q++) {
        const uint32_t m = h_pptr[q + 1] - h_pptr[q];
        h_xoff[q + 1] = h_xoff[q];
        if (m == 0) continue;
        any_phrase = true;
        if (m > 16) return ctx->fail(SS_ERR_UNSUPPORTED, "ss_score_topk_phrase: query %d has a phrase of %u terms (max 16)", q, m);
        uint64_t best = ~0ull;
        bool known = true;
        for (uint32_t i = 0; i < m; i++) {
            const uint32_t t = h_pterms[h_pptr[q] + i];
            if ((uint64_t)t >= s->n_terms) { known = false; break; }    // a doc must contain EVERY phrase term (phrase.go:63)
            const uint64_t df = (tp[t + 1] - tp[t]) + (bp[t + 1] - bp[t]);
            if (df < best) { best = df; h_pdrv[q] = i; }
        }
        if (!known) { h_pdrv[q] = 0xFFFFFFFFu; continue; }
        h_xoff[q + 1] = h_xoff[q] + (uint32_t)std::min<uint64_t>(best, 0x7FFFFFFFull);   // matches <= docs of the rarest term
    }
    // Slice size: SLICE_TARGET postings when the batch fills the chip several times over; smaller (down to
    // SLICE_MIN) for small batches, so that one query's lists are spread over many CUs instead of being
    // walked by a single workgroup (latency of a lone query: 0.72 ms -> see DESIGN.md K4).
    uint64_t slice_target = SLICE_TARGET;
    {
        uint64_t batch_tot = 0;
        for (int q = 0; q < n_q; q++)
            for (uint32_t i = h_qptr[q]; i < h_qptr[q + 1]; i++) {
                const uint32_t t = h_terms[i];
                if ((uint64_t)t < s->n_terms) batch_tot += (tp[t + 1] - tp[t]) + (bp[t + 1] - bp[t]);
            }
        const uint64_t slots = (uint64_t)std::max(ctx->cu_count, 1) * SS_WGS_PER_CU;
        // measured (10M docs, 3-term head queries, batches of 1..4096): one partial wave of slices is best — about
        // 1.5x the batch's postings per resident workgroup slot, never below SLICE_MIN (a slice costs ~45 us of
        // threshold warm-up whatever its size) nor above SLICE_TARGET
        slice_target = std::min<uint64_t>(SLICE_TARGET, std::max<uint64_t>(SLICE_MIN, batch_tot * 3 / (2 * slots)));
        static const char* env = std::getenv("SS_SLICE_TARGET");       // experiments only
        if (env && *env) slice_target = std::max<uint64_t>(1024, std::strtoull(env, nullptr, 10));
    }
    std::vector<uint32_t> h_qoff(n_q + 1, 0), h_dterm, h_dmult, h_sbase(n_q + 1, 0);
    std::vector<double> h_qmag(n_q), h_ub(n_q, 0.0);
    std::vector<SliceDesc> h_slices;
    std::vector<uint64_t> h_slice_cost;
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
        if (topic_probs) {
            // upper bound of sqd over all docs, same operation order as topic_dot (monotone)
            double ub = 0.0;
            for (int t = 0; t < K; t++) {
                const double pt = h_probs[(size_t)q * K + t];
                ub += pt * (pt >= 0.0 ? s->prior_max[t] : s->prior_min[t]);
            }
            h_ub[q] = ub != ub ? INFINITY : ub;
        }
        uint64_t ns = std::max<uint64_t>(1, (tot + slice_target - 1) / slice_target);
        ns = std::min<uint64_t>(ns, std::min<uint64_t>(MAX_SLICES_PER_Q, s->n_docs));
        for (uint64_t j = 0; j < ns; j++) {
            SliceDesc sd;
            sd.q = (uint32_t)q;
            sd.dlo = (uint32_t)(s->n_docs * j / ns);
            sd.dhi = j + 1 == ns ? 0xFFFFFFFFu : (uint32_t)(s->n_docs * (j + 1) / ns);
            sd.pad = 0;
            h_slices.push_back(sd);
            h_slice_cost.push_back(tot / ns);
        }
        h_sbase[q + 1] = (uint32_t)h_slices.size();
    }
    const size_t n_slices = h_slices.size();
    const size_t n_d = h_dterm.size();
    std::vector<uint32_t> h_order(n_slices);
    for (size_t i = 0; i < n_slices; i++) h_order[i] = (uint32_t)i;
    std::stable_sort(h_order.begin(), h_order.end(), [&](uint32_t a, uint32_t b) { return h_slice_cost[a] > h_slice_cost[b]; });

    int cb = SS_CB_MIN;
    while (cb < 2 * k) cb <<= 1;

    // ---- one pinned staging buffer, one H2D copy -------------------------------------
    size_t o = 0;
    const size_t o_qoff = o;   o = align16(o + (n_q + 1) * sizeof(uint32_t));
    const size_t o_dterm = o;  o = align16(o + n_d * sizeof(uint32_t));
    const size_t o_dmult = o;  o = align16(o + n_d * sizeof(uint32_t));
    const size_t o_sbase = o;  o = align16(o + (n_q + 1) * sizeof(uint32_t));
    const size_t o_order = o;  o = align16(o + n_slices * sizeof(uint32_t));
    const size_t o_qmag = o;   o = align16(o + n_q * sizeof(double));
    const size_t o_ub = o;     o = align16(o + n_q * sizeof(double));
    const size_t o_slices = o; o = align16(o + n_slices * sizeof(SliceDesc));
    const size_t o_pptr = o;   o = align16(o + (n_q + 1) * sizeof(uint32_t));
    const size_t o_pterms = o; o = align16(o + h_pterms.size() * sizeof(uint32_t));
    const size_t o_pdrv = o;   o = align16(o + n_q * sizeof(uint32_t));
    const size_t o_xoff = o;   o = align16(o + (n_q + 1) * sizeof(uint32_t));
    const size_t plan_bytes = o;
    const int pb = s->plan_turn;
    s->plan_turn ^= 1;
    if (!s->plan_ev[pb]) SS_HIP(ctx, hipEventCreateWithFlags(&s->plan_ev[pb], hipEventDisableTiming));
    if (s->plan_ev_pending[pb]) {                // the copy that last read this buffer (two calls ago) must be over
        SS_HIP(ctx, hipEventSynchronize(s->plan_ev[pb]));
        s->plan_ev_pending[pb] = false;
    }
    if (s->h_plan_cap[pb] < plan_bytes) {
        if (s->h_plan[pb]) (void)hipHostFree(s->h_plan[pb]);
        s->h_plan[pb] = nullptr;
        s->h_plan_cap[pb] = 0;
        SS_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&s->h_plan[pb]), plan_bytes * 2, hipHostMallocDefault));
        s->h_plan_cap[pb] = plan_bytes * 2;
    }
    SS_HIP(ctx, ensure(s->d_plan, plan_bytes));
    unsigned char* hp = s->h_plan[pb];
    std::memcpy(hp + o_qoff, h_qoff.data(), (n_q + 1) * sizeof(uint32_t));
    if (n_d) {
        std::memcpy(hp + o_dterm, h_dterm.data(), n_d * sizeof(uint32_t));
        std::memcpy(hp + o_dmult, h_dmult.data(), n_d * sizeof(uint32_t));
    }
    std::memcpy(hp + o_sbase, h_sbase.data(), (n_q + 1) * sizeof(uint32_t));
    std::memcpy(hp + o_order, h_order.data(), n_slices * sizeof(uint32_t));
    std::memcpy(hp + o_qmag, h_qmag.data(), n_q * sizeof(double));
    std::memcpy(hp + o_ub, h_ub.data(), n_q * sizeof(double));
    std::memcpy(hp + o_slices, h_slices.data(), n_slices * sizeof(SliceDesc));
    std::memcpy(hp + o_pptr, h_pptr.data(), (n_q + 1) * sizeof(uint32_t));
    if (!h_pterms.empty()) std::memcpy(hp + o_pterms, h_pterms.data(), h_pterms.size() * sizeof(uint32_t));
    std::memcpy(hp + o_pdrv, h_pdrv.data(), n_q * sizeof(uint32_t));
    std::memcpy(hp + o_xoff, h_xoff.data(), (n_q + 1) * sizeof(uint32_t));
    SS_HIP(ctx, hipMemcpyAsync(s->d_plan.p, hp, plan_bytes, hipMemcpyHostToDevice, st));
    SS_HIP(ctx, hipEventRecord(s->plan_ev[pb], st));
    s->plan_ev_pending[pb] = true;
    if (any_phrase) {
        for (int x = 0; x < 4; x++) SS_HIP(ctx, ensure(s->d_x[x], (size_t)h_xoff[n_q]));
        SS_HIP(ctx, ensure(s->d_xcnt, (size_t)n_q * 4));
    }
    SS_HIP(ctx, ensure(s->d_so_key, n_slices * k));
    SS_HIP(ctx, ensure(s->d_so_doc, n_slices * k));
    SS_HIP(ctx, ensure(s->d_so_cnt, n_slices));
    SS_HIP(ctx, ensure(s->d_hits, (size_t)n_q * k));
    SS_HIP(ctx, ensure(s->d_nhits, n_q));
    if (topic_probs) {
        SS_HIP(ctx, ensure(s->d_probs, (size_t)n_q * K));
        SS_HIP(ctx, hipMemcpyAsync(s->d_probs.p, topic_probs, (size_t)n_q * K * sizeof(double), hipMemcpyDefault, st));
    }

    const unsigned char* dp = s->d_plan.p;
    ScoreParams p{};
    p.t_ptr = s->title->term_ptr.p; p.t_post = s->t_post.p;
    p.b_ptr = s->body->term_ptr.p; p.b_post = s->b_post.p;
    p.t_pos_ptr = s->title->pos_ptr.p; p.t_pos = s->title->pos.p;
    p.b_pos_ptr = s->body->pos_ptr.p; p.b_pos = s->body->pos.p;
    if (any_phrase) {
        p.ph_off = reinterpret_cast<const uint32_t*>(dp + o_pptr);
        p.ph_terms = reinterpret_cast<const uint32_t*>(dp + o_pterms);
        p.ph_drv = reinterpret_cast<const uint32_t*>(dp + o_pdrv);
        p.x_off = reinterpret_cast<const uint32_t*>(dp + o_xoff);
        for (int x = 0; x < 4; x++) p.x_list[x] = s->d_x[x].p;
        p.x_cnt = s->d_xcnt.p;
    }
    p.prior = K ? s->prior.p : nullptr;
    p.k_topics = K;
    p.q_off = reinterpret_cast<const uint32_t*>(dp + o_qoff);
    p.dterm = reinterpret_cast<const uint32_t*>(dp + o_dterm);
    p.dmult = reinterpret_cast<const uint32_t*>(dp + o_dmult);
    p.qmag = reinterpret_cast<const double*>(dp + o_qmag);
    p.probs = topic_probs ? s->d_probs.p : nullptr;
    p.sqd_ub = reinterpret_cast<const double*>(dp + o_ub);
    p.slice_base = reinterpret_cast<const uint32_t*>(dp + o_sbase);
    p.slices = reinterpret_cast<const SliceDesc*>(dp + o_slices);
    p.order = reinterpret_cast<const uint32_t*>(dp + o_order);
    p.k = k;
    p.cb = cb;
    p.so_key = s->d_so_key.p; p.so_doc = s->d_so_doc.p; p.so_cnt = s->d_so_cnt.p;
    // results straight into the caller's buffers when both live in device memory (then the call does not wait either)
    bool dev_out = false;
    {
        hipPointerAttribute_t a1{}, a2{};
        const bool d1 = hipPointerGetAttributes(&a1, hits_out) == hipSuccess && a1.type == hipMemoryTypeDevice;
        const bool d2 = hipPointerGetAttributes(&a2, n_hits_out) == hipSuccess && a2.type == hipMemoryTypeDevice;
        (void)hipGetLastError();                 // plain host memory is reported as an error: not one
        dev_out = d1 && d2;
    }
    p.hits = dev_out ? hits_out : s->d_hits.p;
    p.n_hits = dev_out ? n_hits_out : s->d_nhits.p;

    const size_t lds_score = score_lds_bytes(cb), lds_merge = merge_lds_bytes(k, cb);
    if (s->lds_attr < cb) {
        SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_score_slices), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_score));
        SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_merge_topk), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)merge_lds_bytes(SS_MAX_TOPK, cb)));
        s->lds_attr = cb;
    }
    SS_HIP(ctx, hipEventRecord(ctx->ev[1][0], st));
    if (any_phrase) hipLaunchKernelGGL(k_phrase_match, dim3((unsigned)n_q), dim3(PH_TPB), 0, st, p);
    hipLaunchKernelGGL(k_score_slices, dim3((unsigned)n_slices), dim3(TPB), lds_score, st, p);
    hipLaunchKernelGGL(k_merge_topk, dim3((unsigned)n_q), dim3(TPB_M), lds_merge, st, p);
    SS_HIP(ctx, hipEventRecord(ctx->ev[1][1], st));
    ctx->ev_valid[1] = true;
    SS_HIP(ctx, hipGetLastError());
    if (dev_out) return SS_OK;                   // ordered on the ctx stream; ss_synchronize (or the stream's owner) waits
    SS_HIP(ctx, hipMemcpyAsync(hits_out, s->d_hits.p, (size_t)n_q * k * sizeof(ss_hit), hipMemcpyDefault, st));
    SS_HIP(ctx, hipMemcpyAsync(n_hits_out, s->d_nhits.p, n_q * sizeof(int32_t), hipMemcpyDefault, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    return SS_OK;
}

}  // extern "C"
