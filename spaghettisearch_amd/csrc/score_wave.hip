// score_wave.hip — k_score_wave: the OR-query scorer with ONE WAVE per (query, doc-range slice), for gfx950 (MI355X).
//
// Same job and same results as k_score_slices (score.hip; retrieval/main_retrieve.go:50-103,170-247,
// get_metadata.go:31-69): stream the 8-byte scoring records of the query's posting lists, bound every document's
// FinalRank from above with a fixed-point sketch in LDS, send the few documents that can still enter the top-k through
// the reference's float64 arithmetic, keep the k best.  What differs is who owns what:
//
//   * a slice belongs to ONE wave (a 64-thread workgroup).  The sketch, the window plan, the pending survivors, the
//     exact stage's table and the running top-k are all private to the wave, so the main loop has no workgroup barrier
//     and no cross-wave traffic at all: LDS operations of one wave execute in order, which is the only ordering the
//     filter needs (add -> read back -> clear).
//   * the lists are the COMBINED lists of the scorer (title and body postings of a term merged by doc, field in bit 31 of
//     the doc word): a query of n terms is n lists, not 2n, and the sparse title lists cost no blocks of their own.
//   * records are read as whole 512-byte BLOCKS (64 records, aligned in the record array) found through a skip index
//     (skip[g] = doc of record 64*g, 1/128 of the record bytes).  A window is a doc range [b_lo, b_hi); a block belongs
//     to every window it overlaps and each record decides by ONE compare whether it is in the current window — so
//     window cursors never have to be searched in the posting lists (the set-up of k_score_slices is ~13 % of a slice
//     and consists of nothing but such latency chains).  The skip entries of the slice's doc range are staged in LDS
//     in rounds; the window plan of a round (which blocks of which list every window reads) is computed from them
//     with LDS searches only.
//   * the loop is a ring of WDEPTH groups of WCW block loads: always WCW loads per group (empty slots read a dummy
//     word), so that every wait in front of a group is a counted s_waitcnt vmcnt(N).  Per block and lane: one subtract
//     and compare (window membership), three integer operations for the slot, four for the fixed-point share, ds_add,
//     ds_read, compare, ds_write — about a quarter of the instructions k_score_slices spends per 64 records.
//   * windows that do not fit a group (more than WCW blocks) and windows with more survivors than the pending list
//     holds take a slow path that walks the window's blocks one at a time and hands the survivors to the exact stage by
//     doc sub-range (bisected until a piece fits; a single doc has at most WL postings).  Rare, not tuned, exact.
//
// The host routes a query here when it has at most WL lists, no phrase part, k <= WK_MAX, clean inputs (the filter's
// assumptions hold) and a list long enough for the threshold floor to be meaningful; everything else runs k_score_slices.
// The slices' top-k lists are merged per query by k_merge_topk.
#include "score_common.hpp"

namespace {

constexpr int WCW = 16;                 // block slots per group (= one regular window)
#ifndef SSW_PACK
#define SSW_PACK 16      // blocks a packed window may hold (<= WCW)
#endif
#ifndef SSW_MINW
#define SSW_MINW 3
#endif
#ifndef SSW_DEPTH
#define SSW_DEPTH 2
#endif
constexpr int WDEPTH = SSW_DEPTH;       // groups whose loads are in flight or in registers (2 or 3)
#ifndef SSW_WSK
#define SSW_WSK 1024
#endif
constexpr int WSK = SSW_WSK;            // sketch slots (1024: 2048 measured, see DESIGN K4b); slot WSK is a dummy that always holds 0
// LDS per wave decides the occupancy: 13.6 KB is three waves per SIMD (12 per CU), and the loop is latency-bound below that
#ifndef SSW_SE
#define SSW_SE 224
#endif
#ifndef SSW_SLACK
#define SSW_SLACK 3
#endif
constexpr int WSE = SSW_SE;             // skip entries staged per round
constexpr int WGMAX = 2 * ((WSE + 12 + 15) / 16) + 2;   // group rows per round
constexpr int WPW = 96;                 // pending survivors = capacity of the wave's exact stage
constexpr int WHT = 128;                // exact-stage hash slots (they share their LDS with the skip entries: plan_round refills them)
constexpr int WHT_SHIFT = 25;           // 32 - log2(WHT)
static_assert((1 << (32 - WHT_SHIFT)) == WHT, "hash shift");
constexpr int WL = 12;                  // lists per query
constexpr int WCB = 256;                // candidate buffer (>= 2k)
constexpr uint32_t WINF = 0xFFFFFFFFu;
// block descriptor: list (4 bits) | D_FIRST | D_LAST | block index in the table's record array << 6.  A slot without a block
// names a block the wave reads anyway, as a "list" whose lane bounds are empty: the loop has no "empty" case.
constexpr uint32_t D_FIRST = 0x10u, D_LAST = 0x20u;    // the block is the first / last one of its list: part of it belongs to a neighbouring list
constexpr uint32_t FX_CLAMP_SLOW = 1u << 17;           // oversize windows: (WSE + WL) * 64 records of this much stay below 2^32
static_assert((uint64_t)(WSE + WL) * 64 * FX_CLAMP_SLOW < (1ull << 32), "slow-path sums must not wrap");
static_assert((uint64_t)WCW * 64 * FX_CLAMP < (1ull << 32), "fast-path sums must not wrap");
static_assert(2 * ((WSE + WL + WCW - 1) / WCW) <= WGMAX, "a round's plan must fit its rows");
static_assert(WPW <= 128, "wave_flush takes two pending entries per lane");
enum : uint32_t { M_NORMAL = 1, M_SLOW = 2, M_SKIP = 0 };

#ifdef SSW_CHECK
// checked build (tools/build_diag.sh with EXTRA=-DSSW_CHECK): a block index outside its table is recorded and replaced by block 0
__device__ uint32_t g_wcheck[8];
#endif
#ifdef SSW_PHASES
// -DSSW_PHASES (tools/build_variant.sh phases -DSSW_PHASES): s_memtime intervals of a slice's phases, kept in LDS and added to 256
// rows of global counters at the slice's end — the kernel keeps its time (0.401 against 0.396 ms per batch).  Printed by
// ss_scorer_destroy.  At config 3: set-up 5.8 %, planning 17.1 %, streaming 65.7 %, events 3.7 %, final flush 4.0 %, hand-in 2.9 %.
__device__ unsigned long long g_wrows[256][4];
__device__ unsigned long long g_wphase[256][8];     // spread over 256 rows: 105k atomics per batch on eight words took 0.8 ms (the SS_DIAG build's counters still do that)
__shared__ unsigned long long ph_acc[8];
// interval i opens (SGN = -) and closes (SGN = +): the stamp goes straight into LDS
#define PH_MARK(i, SGN) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); if (threadIdx.x == 0) ph_acc[i] SGN##= t_; } while (0)
#else
#define PH_MARK(i, SGN) do { } while (0)
#endif
#ifdef SSW_PHASES
__shared__ unsigned int ph_rows[4];     // rows, rows with a survivor, survivors, rows with > 4 survivors
#define PH_COUNT(tot) do { if (threadIdx.x == 0) { ph_rows[0]++; ph_rows[1] += (tot) ? 1u : 0u; ph_rows[2] += (tot); ph_rows[3] += (tot) > 4u ? 1u : 0u; } } while (0)
#else
#define PH_COUNT(tot) do { } while (0)
#endif
#ifdef SS_DIAG
__device__ unsigned long long g_wdiag[256][32];     // 256 rows (by block index): atomics of 13k slices on 32 words serialise and cost the kernel more than it takes
#define WDIAG_ADD(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_wdiag[blockIdx.x & 255][i], (unsigned long long)(v)); } while (0)
#else
#define WDIAG_ADD(i, v) do { } while (0)
#endif

__device__ __forceinline__ uint32_t rl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// number of set bits of the wave mask m below this lane (v_mbcnt: no 64-bit per-lane mask to keep in registers)
__device__ __forceinline__ uint32_t bits_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}
// exclusive prefix sum over the lanes of the wave
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, int lane) {
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, o, 64);
        if (lane >= o) x += y;
    }
    return x - v;
}

// number of entries of a[0, n) (ascending, in LDS) that are < v
__device__ __forceinline__ uint32_t lds_lower_bound(const uint32_t* a, uint32_t n, uint32_t v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ uint32_t w_slot(uint32_t doc) { return (doc ^ (doc >> 10)) & (uint32_t)(WSK - 1); }
static_assert((WSK & (WSK - 1)) == 0 && WSK % 256 == 0, "sketch size");

// What a slice needs to know about one of its lists, resolved by k_wave_prep for all slices of the batch at once (the
// chains of dependent loads and the searches in the skip index then run side by side instead of at the head of every slice).
struct __attribute__((aligned(16))) WPrep {
    uint32_t g0, g1;          // first / last block of the (combined) list
    uint32_t cg, ge;          // blocks that hold the slice's doc range
    uint32_t lanes;           // lo_lane | hi_lane << 8 | active << 16
    float kth_b, kth_t;       // k'-th largest impact of the term's body / title list (threshold floor)
    uint32_t mult;
};

typedef const ScoreParams __attribute__((address_space(4)))* kparams_chk_t;
struct RareArgs { const double* t_mag; const double* b_mag; const double* prior; const float* c_w; int32_t k_topics, k; };

// The wave's LDS, at namespace scope: k_score_wave's rare paths (the exact stage, an oversize window) are functions the kernel
// CALLS (inlined they cost the hot loop its registers), and a called function that gets LDS pointers as arguments sees generic
// pointers: every access became a flat_load with vmcnt(0) lgkmcnt(0) behind it, and the argument structs travelled through
// scratch memory.  Named directly, the same accesses are ds_ instructions and the calls pass two scalars.
__shared__ __attribute__((aligned(16))) uint32_t sk[WSK + 4];

__shared__ uint32_t se[(WSE + 32) > 2 * WHT ? (WSE + 32) : 2 * WHT];   // skip entries while a round is planned, then ht_key | ht_rec
__shared__ __attribute__((aligned(16))) uint32_t ghdr[WGMAX + 2 * WDEPTH][4];
__shared__ uint32_t gdesc[WGMAX + 2 * WDEPTH][WCW];
__shared__ __attribute__((aligned(16))) unsigned char pend_raw[WPW * 16];
__shared__ uint64_t cd_key[WCB];
__shared__ uint32_t cd_doc[WCB];
__shared__ uint32_t l_mult[WL], l_adv[WL];
__shared__ uint64_t sc64[2];
__shared__ uint32_t sc32[8];
struct RareState {          // what the called paths need of the slice and the batch, written once by the slice's set-up
    SliceQuery Q;
    RareArgs ra;
    uint64_t thr0_key, tb;
    float thr0_f, r_ub, fx_scale;
};
__shared__ RareState rs;

struct WaveLds {
    double2* s_rec;      // [WPW] exact stage: {addend, magnitude}
    uint4* pend;         // [WPW] the same bytes while pending: {doc, index in the combined arrays, term, field}
    uint32_t* l_mult;    // [WL]
    uint32_t* ht_key;    // [WHT]
    uint32_t* ht_rec;    // [WHT]
    uint32_t* overflow;
};

__device__ __forceinline__ WaveLds wave_lds() {
    return WaveLds{reinterpret_cast<double2*>(pend_raw), reinterpret_cast<uint4*>(pend_raw), l_mult, se, se + WHT, &sc32[1]};
}
__device__ __forceinline__ TopK wave_topk(uint64_t thr0_key, float thr0_f) {
    return TopK{cd_key, cd_doc, &sc32[0], &sc64[0], reinterpret_cast<float*>(&sc32[2]), thr0_key, thr0_f, (uint32_t)WCB};
}
// the candidate buffer sorted and cut to k (topk_compact for one wave), as a call
__device__ __noinline__ void wave_compact(int k) {
    const TopK tk = wave_topk(rs.thr0_key, rs.thr0_f);
    topk_compact_inl<false>(tk, k);
}

// sqd for the benchmark's 16 topics: the row's 16 doubles and the 16 probabilities requested together (the loop of runtime length
// waits for every product's two loads in turn), the products added in topic order as the loop adds them.  A CALL, so that its
// 64 registers of operands are not live beside the exact stage's own.
__device__ __noinline__ double prior_dot16(gptr_f64 pr, gptr_f64 pb) {
    double r[16], q[16];
#pragma unroll
    for (int t = 0; t < 16; t++) { r[t] = pr[t]; q[t] = pb[t]; }
    double sqd = 0.0;
#pragma unroll
    for (int t = 0; t < 16; t++) sqd += q[t] * r[t];
    return sqd;
}

// ---- exact stage of one wave (flush_pending / score_owned of score.hip for 64 threads) ------------------------------
__device__ __forceinline__ void wave_score_owned(const WaveLds& S, const TopK& tk, const SliceQuery& Q, const RareArgs& ra, int lane,
                                                 uint32_t doc, uint32_t slot) {
    const RareArgs* kp = &ra;
    uint32_t e_doc = EMPTY;
    uint64_t e_key = 0;
    if (slot != EMPTY) {
        e_doc = doc;
        const uint32_t fr = S.ht_rec[slot];
        double2 rb = make_double2(0.0, 1.0), rt = make_double2(0.0, 1.0);
        const uint32_t ib = fr & 0xFFFFu, it = fr >> 16;
        if (ib != NOREC) rb = S.s_rec[ib];
        if (it != NOREC) rt = S.s_rec[it];
        S.ht_key[slot] = EMPTY;
        S.ht_rec[slot] = EMPTY;
        const double B = rb.x, T = rt.x, mb = rb.y, mt = rt.y;
        const uint64_t thr0 = *tk.thr;
        double title, body, fin;
        if (Q.probs) {
            // the prior row is only fetched if the doc can still make the top-k (every operation of final_rank is monotone in sqd)
            final_rank(T, B, mt, mb, Q.qmag, Q.sqd_ub, title, body, fin);
            if (fkey(fin) >= thr0 || fin != fin) {
                double sqd = 0.0;                        // topic_dot (get_metadata.go:39-42, topic order) through global pointers
                const gptr_f64 pr = (gptr_f64)kp->prior + (size_t)e_doc * kp->k_topics, pb = (gptr_f64)Q.probs;
                if (kp->k_topics == 16) sqd = prior_dot16(pr, pb);
                else
                for (int t = 0; t < kp->k_topics; t++) sqd += pb[t] * pr[t];
                final_rank(T, B, mt, mb, Q.qmag, sqd, title, body, fin);
            }
            else e_doc = EMPTY;
        } else {
            final_rank(T, B, mt, mb, Q.qmag, 0.0, title, body, fin);
        }
        e_key = fkey(fin);
    }
    for (;;) {
        const uint64_t thr = *tk.thr;
        if (e_doc != EMPTY) {
            if (e_key >= thr) {
                const uint32_t i = atomicAdd(tk.count, 1u);
                if (i < tk.cb) { tk.key[i] = e_key; tk.doc[i] = e_doc; e_doc = EMPTY; }
                else *S.overflow = 1;
            } else {
                e_doc = EMPTY;
            }
        }
        lds_wait();
        if (!*S.overflow) break;
        wave_compact(kp->k);
        if (lane == 0) *S.overflow = 0;
        lds_wait();
    }
}

// NR = pending entries per lane: most flushes (the one at the end of every slice above all) hold fewer than 64 entries, and the
// second entry's share of the code would run for nothing
template <int NR>
__device__ __forceinline__ void flush_body(int lane, uint32_t n) {
    const WaveLds S = wave_lds();
    const SliceQuery Q = rs.Q;
    const RareArgs ra = rs.ra;
    const TopK tk = wave_topk(rs.thr0_key, rs.thr0_f);
    const RareArgs* kp = &ra;
    DIAG_NOW(t_f0);
    uint32_t pdoc[NR], pl[NR], pf[NR], own[NR];
    float pw[NR];
    double pm[NR];
    // no branch around the loads (a load under a branch makes the compiler drain every load in flight where the branch joins:
    // the second half's loads would wait for the first's): lanes without an entry read entry 0 (n >= 1) and drop it
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const uint32_t i = lane + r * 64;
        const bool in = i < n;
        const uint4 e = S.pend[in ? i : 0u];
        pdoc[r] = e.x;
        pl[r] = in ? e.z : EMPTY;
        pf[r] = e.w;
        // (the pointers come out of LDS as generic ones: named global, the loads are global_load and count on vmcnt alone)
        pw[r] = ((gptr_f32)kp->c_w)[e.y];               // the posting's float32 weight, by its position in the combined list
        pm[r] = ((gptr_f64)(e.w ? kp->t_mag : kp->b_mag))[e.x];
    }
    lds_wait();                                     // every pending entry has been read: the bytes may be rewritten
#ifdef SS_DIAG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    DIAG_NOW(t_f1);
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const uint32_t l = pl[r];
        own[r] = EMPTY;
        if (l != EMPTY) {
            const uint32_t i = lane + r * 64;
            uint32_t h = (pdoc[r] * 2654435761u) >> WHT_SHIFT;      // -> [0, WHT)
            for (;;) {
                const uint32_t prev = atomicCAS(&S.ht_key[h], EMPTY, pdoc[r]);
                if (prev == EMPTY) own[r] = h;
                if (prev == EMPTY || prev == pdoc[r]) break;
                h = (h + 1) & (uint32_t)(WHT - 1);
            }
            const uint32_t field = pf[r];
            const double v = (double)pw[r] * (double)S.l_mult[l];    // main_retrieve.go:61-69: a duplicate token counts again
            S.s_rec[i] = make_double2(v, pm[r]);
            asm volatile("" ::: "memory");
            uint32_t old = S.ht_rec[h], first;
            for (;;) {
                first = field ? old >> 16 : old & 0xFFFFu;
                if (first != NOREC) break;
                const uint32_t want = field ? (old & 0xFFFFu) | (i << 16) : (old & 0xFFFF0000u) | i;
                const uint32_t prev = atomicCAS(&S.ht_rec[h], old, want);
                if (prev == old) break;
                old = prev;
            }
            if (first != NOREC) atomicAdd(&S.s_rec[first].x, v);
        }
    }
    lds_wait();
    DIAG_NOW(t_f2);
#pragma unroll
    for (int r = 0; r < NR; r++) wave_score_owned(S, tk, Q, ra, lane, pdoc[r], own[r]);
    DIAG_NOW(t_f3);
    WDIAG_ADD(21, t_f1 - t_f0);
    WDIAG_ADD(22, t_f2 - t_f1);
    WDIAG_ADD(23, t_f3 - t_f2);
}

__device__ __noinline__ void wave_flush(int lane, uint32_t n) {
    if (n > 64u) flush_body<2>(lane, n);
    else flush_body<1>(lane, n);
}

// per-list constants, lane l < L holds list l (= distinct query term l)
struct WList {
    uint32_t g0, g1;            // first / last 64-record block of the list
    uint32_t lo_lane, hi_lane;  // lanes of block g0 below lo_lane and lanes of block g1 from hi_lane on belong to other lists
    uint32_t ge;                // last block that can hold a doc of the slice
    uint32_t cg;                // block that holds the frontier
    uint32_t q, soff;           // this round: skip entries staged, where in `se`
    float coef_b, coef_t;       // filter coefficients of a body / title posting, in fixed-point units
    bool active;
};

// membership of a block's 64 records in the doc range [b_lo, b_lo + span) of list `l`
__device__ __forceinline__ bool block_active(uint32_t d, uint32_t doc, uint32_t b_lo, uint32_t span, const WList& w, int lane) {
    bool act = (doc - b_lo) < span;
    if (d & (D_FIRST | D_LAST)) {                    // scalar: rare
        const int l = (int)(d & 15u);
        const uint32_t lo = (d & D_FIRST) ? rl(w.lo_lane, l) : 0u;
        const uint32_t hi = (d & D_LAST) ? rl(w.hi_lane, l) : 64u;
        act = act && (uint32_t)lane >= lo && (uint32_t)lane < hi;
    }
    return act;
}

// address of the 512-byte block a descriptor names: table base (scalar, by the list's field) + 512 * block index
// address of the 512-byte block a descriptor names
__device__ __forceinline__ uint64_t block_addr(uint32_t d, uint64_t tb) { return tb + ((uint64_t)(d >> 6) << 9); }

struct WaveCtx {
    uint32_t* sk;
    uint32_t (*ghdr)[4];
    uint32_t (*gdesc)[WCW];
    uint64_t tb;             // the combined record array
    WaveLds S;
    TopK tk;
    SliceQuery Q;
    float r_ub, fx_scale;
};

__device__ __forceinline__ uint32_t wave_thr_fx(const WaveCtx& C) { return max(1u, fx_threshold(*C.tk.thr_f, C.r_ub, C.fx_scale)); }

// One window the slow way: its blocks one at a time.  need_add: the window's shares are not in the sketch yet (oversize
// window; they are added with the smaller clamp, and the whole sketch is cleared at the end).  Otherwise the caller has
// added them (and its slots are cleared here).  The pending list must be empty.  Survivors go to the exact stage by doc
// sub-range, bisected until a piece fits.
__device__ __noinline__ void slow_window(const WList w, int lane, uint32_t row, uint32_t n_blocks,
                                         uint32_t b_lo, uint32_t span, bool need_add) {
    WaveCtx C;
    C.sk = sk; C.ghdr = ghdr; C.gdesc = gdesc; C.tb = rs.tb;
    C.S = wave_lds();
    C.tk = wave_topk(rs.thr0_key, rs.thr0_f);
    C.r_ub = rs.r_ub; C.fx_scale = rs.fx_scale;
    const uint32_t clamp = need_add ? FX_CLAMP_SLOW : FX_CLAMP;
    if (need_add) {
        for (uint32_t i = 0; i < n_blocks; i++) {
            const uint32_t d = rfl(C.gdesc[row + i / WCW][i % WCW]);
            const u32x2 rec = *(gptr_u2)(block_addr(d, C.tb) + (uint32_t)lane * 8u);
            const uint32_t doc = rec.x & 0x7FFFFFFFu;
            const float cf = __uint_as_float(rl(__float_as_uint(w.coef_b), (int)(d & 15u)));
            if (block_active(d, doc, b_lo, span, w, lane))
                atomicAdd(&C.sk[w_slot(doc)], min(fx_share(__uint_as_float(rec.y), cf), clamp));
        }
        lds_wait();
    }
    uint32_t cur = 0;                                 // docs [b_lo, b_lo + cur) are done
    while (cur < span) {
        uint32_t sub = span - cur;
        const uint32_t thr = min(wave_thr_fx(C), clamp);
        for (;;) {
            uint32_t cnt = 0;
            for (uint32_t i = 0; i < n_blocks; i++) {
                const uint32_t d = rfl(C.gdesc[row + i / WCW][i % WCW]);
                const u32x2 rec = *(gptr_u2)(block_addr(d, C.tb) + (uint32_t)lane * 8u);
                const uint32_t doc = rec.x & 0x7FFFFFFFu;
                const bool act = block_active(d, doc, b_lo, span, w, lane) && (doc - (b_lo + cur)) < sub;
                const uint32_t u = act ? C.sk[w_slot(doc)] : 0u;
                cnt += (uint32_t)__popcll(__ballot(act && u >= thr));
            }
            if (cnt <= (uint32_t)WPW || sub == 1) break;
            sub = (sub + 1) >> 1;
        }
        uint32_t n = 0;
        for (uint32_t i = 0; i < n_blocks; i++) {
            const uint32_t d = rfl(C.gdesc[row + i / WCW][i % WCW]);
            const u32x2 rec = *(gptr_u2)(block_addr(d, C.tb) + (uint32_t)lane * 8u);
            const uint32_t doc = rec.x & 0x7FFFFFFFu;
            const bool act = block_active(d, doc, b_lo, span, w, lane) && (doc - (b_lo + cur)) < sub;
            const uint32_t u = act ? C.sk[w_slot(doc)] : 0u;
            const bool surv = act && u >= thr;
            const unsigned long long m = __ballot(surv);
            if (m) {
                const int l = (int)(d & 15u);
                const uint32_t pos = n + bits_below(m);
                if (surv && pos < (uint32_t)WPW) C.S.pend[pos] = make_uint4(doc, (d >> 6) * 64u + (uint32_t)lane, (uint32_t)l, rec.x >> 31);
                n += (uint32_t)__popcll(m);
            }
        }
        lds_wait();
        if (n) wave_flush(lane, min(n, (uint32_t)WPW));
        cur += sub;
    }
    if (need_add) {
        for (int i = lane; i < WSK; i += 64) C.sk[i] = 0u;
    } else {
        // the caller's adds: clear exactly the slots of the window's records
        for (uint32_t i = 0; i < n_blocks; i++) {
            const uint32_t d = rfl(C.gdesc[row + i / WCW][i % WCW]);
            const u32x2 rec = *(gptr_u2)(block_addr(d, C.tb) + (uint32_t)lane * 8u);
            if (block_active(d, rec.x & 0x7FFFFFFFu, b_lo, span, w, lane)) C.sk[w_slot(rec.x & 0x7FFFFFFFu)] = 0u;
        }
    }
    lds_wait();
}

// The ring's loads are hidden from the compiler (inline asm): its s_waitcnt bookkeeping drains a 48-deep ring at every
// step (it emitted vmcnt(15..0) where vmcnt(47..32) is right), which exposes one memory latency per row.  So the loads
// are counted by hand: a group is ALWAYS WCW loads, nothing else issues vector-memory operations inside the hot loop,
// and block c of the group being processed has exactly (WCW - 1 - c) + (WDEPTH - 1) * WCW younger loads in flight.
// ring_wait<N> names the destination "+v" so that no consumer is scheduled above it (cdna_hip_programming.md §5.7 (ii)).
#ifndef SSW_PLAIN_RING
#define SSW_ASM_RING 1      // -DSSW_PLAIN_RING: compiler-counted loads (A/B builds; about 1.6x slower)
#endif
#ifdef SSW_ASM_RING
template <int N>
__device__ __forceinline__ void ring_wait(u32x2& r) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N)); }
__device__ __forceinline__ void ring_keep(u32x2& r) { asm volatile("" : "+v"(r)); }
template <int N>
__device__ __forceinline__ void ring_drain_slot() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }   // the slot's data is not used: no operand, no copy
#else
template <int N>
__device__ __forceinline__ void ring_drain_slot() {}
template <int N>
__device__ __forceinline__ void ring_wait(u32x2&) {}
__device__ __forceinline__ void ring_keep(u32x2&) {}
#endif

template <int S_, int C_>
__device__ __forceinline__ void block_issue(uint64_t tb, uint32_t dv, uint32_t lane8, u32x2 (&rec)[WDEPTH][WCW]) {
#ifdef SSW_CHECK
    uint32_t dchk = rl(dv, C_);
    {
        const kparams_chk_t kp = (kparams_chk_t)__builtin_amdgcn_kernarg_segment_ptr();
        const uint32_t lim = kp->c_pad_block;
        if ((dchk >> 6) > lim) {
            if (lane8 == 0) { atomicAdd(&g_wcheck[0], 1u); g_wcheck[1] = dchk; g_wcheck[2] = lim; g_wcheck[3] = blockIdx.x; }
            dchk &= 63u;
        }
    }
    const uint64_t base = block_addr(dchk, tb);
#else
    const uint64_t base = block_addr(rl(dv, C_), tb);               // scalar
#endif
#ifdef SSW_ASM_RING
    const uint64_t addr = base + lane8;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rec[S_][C_]) : "v"(addr));
#else
    rec[S_][C_] = *(gptr_u2)(base + lane8);
#endif
}
template <int S_>
__device__ __forceinline__ void group_issue(const ScoreParams& p, uint32_t (*gdesc)[WCW], uint32_t row, int lane, u32x2 (&rec)[WDEPTH][WCW],
                                            uint32_t (&dvr)[WDEPTH]) {
    // the table base straight from the kernel arguments (a scalar register by construction; as a member of a struct that is
    // also handed to the non-inlined rare paths it ends up in vector registers)
    const uint64_t tb = (uint64_t)p.c_rec;
    const uint32_t dv = gdesc[row][lane & (WCW - 1)];
    const uint32_t lane8 = (uint32_t)lane * 8u;
    dvr[S_] = dv;
    block_issue<S_, 0>(tb, dv, lane8, rec);  block_issue<S_, 1>(tb, dv, lane8, rec);
    block_issue<S_, 2>(tb, dv, lane8, rec);  block_issue<S_, 3>(tb, dv, lane8, rec);
    block_issue<S_, 4>(tb, dv, lane8, rec);  block_issue<S_, 5>(tb, dv, lane8, rec);
    block_issue<S_, 6>(tb, dv, lane8, rec);  block_issue<S_, 7>(tb, dv, lane8, rec);
    block_issue<S_, 8>(tb, dv, lane8, rec);  block_issue<S_, 9>(tb, dv, lane8, rec);
    block_issue<S_, 10>(tb, dv, lane8, rec); block_issue<S_, 11>(tb, dv, lane8, rec);
    block_issue<S_, 12>(tb, dv, lane8, rec); block_issue<S_, 13>(tb, dv, lane8, rec);
    block_issue<S_, 14>(tb, dv, lane8, rec); block_issue<S_, 15>(tb, dv, lane8, rec);
    static_assert(WCW == 16, "group_issue is written out for 16 block slots");
}

// block C_ of ring slot S_: every record of the window adds its share into the sketch; returns the slot's BYTE offset
// (4 * WSK: the lane holds no record of the window)
template <int S_, int C_>
__device__ __forceinline__ uint32_t block_add(uint32_t* sk, const WList& w, uint32_t dv, uint32_t b_lo, uint32_t span, int lane,
                                              u32x2 (&rec)[WDEPTH][WCW]) {
    ring_wait<(WCW - 1 - C_) + (WDEPTH - 1) * WCW>(rec[S_][C_]);
#ifdef SSW_EXP_SKIPADD      // timing experiment only (wrong results): loads and waits, no filter work
    return (rec[S_][C_].x == 0x12345678u) ? 0u : 4u * (uint32_t)WSK;
#endif
    const uint32_t d = rl(dv, C_);
    const float cf = __uint_as_float(rl(__float_as_uint(w.coef_b), (int)(d & 15u)));      // one coefficient per list: title impacts are stored pre-scaled (k_merge_lists)
    const uint32_t doc = rec[S_][C_].x & 0x7FFFFFFFu;
    uint32_t h4 = 4u * (uint32_t)WSK;
    if (block_active(d, doc, b_lo, span, w, lane)) {
        h4 = w_slot(doc) * 4u;
        atomicAdd(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(sk) + h4), fx_share(__uint_as_float(rec[S_][C_].y), cf));
    }
    return h4;
}

// ---- planning of one round (not hot: once per ~WSE blocks) ---------------------------------------------------------
// Stages the skip entries of the blocks after every list's cursor, cuts [F, e) into windows of `s` driver blocks and
// writes the rows (ghdr / gdesc) the streaming loop reads.  Returns {n_rows, e}; adv_out[l] = how many blocks list l's
// cursor advances when the next round starts at e.
struct RoundPlan { uint32_t n_rows, e; };
__device__ __noinline__ RoundPlan plan_round(WList w, const uint32_t* c_skip, uint32_t* se, uint32_t (*ghdr)[4], uint32_t (*gdesc)[WCW],
                                             uint32_t* adv_out, int L, unsigned long long act_mask, uint32_t F, uint32_t dhi, int lane) {
    // a slot without a block re-reads the cursor block of the first active list as "list" 15, whose lane bounds are
    // empty: no record of it is ever in a window
    const int l_first = __ffsll((long long)act_mask) - 1;
    const uint32_t d_empty = 15u | D_FIRST | D_LAST | (rl(w.cg, l_first) << 6);
    const uint32_t rem = w.active && lane < L ? w.ge - w.cg + 1u : 0u;
    const uint32_t tot_rem = wave_sum(rem);
    uint32_t ql = 0;
    // At least TWO entries of every list that has them: a round that ended at a block boundary of this list starts the next
    // one with its first staged entry EQUAL to F; with that entry alone the list's "last staged entry" is F again, the round
    // ends where it began and the wave never returns (seen with slices above 32k postings: a list a hundred times sparser
    // than its neighbours was given one entry per round).  Entries of a list are strictly increasing, so the second is > F.
    if (rem > 1) ql = min(rem - 1u, max(2u, (uint32_t)(((uint64_t)(WSE - WL) * rem) / tot_rem)));
    w.q = ql;
    w.soff = wave_excl_scan(ql, lane);
    // the lists' entries lie back to back in `se` (soff is a running sum): entry j belongs to the last list with entries whose
    // soff <= j.  All of a lane's loads are issued before the first is stored — list by list, every list cost the round one
    // memory latency.  No branch around a load: lanes past the end repeat the last entry.
    const uint32_t tot_q = wave_sum(ql);
    if (tot_q) {
        constexpr int NIT = (WSE + 63) / 64;
        uint32_t sv[NIT], sj[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const uint32_t j = min((uint32_t)lane + 64u * (uint32_t)it, tot_q - 1u);
            uint32_t so = 0, cgl = 0;
            for (int l = 0; l < L; l++) {
                const uint32_t s_l = rl(w.soff, l);
                if (rl(w.q, l) && j >= s_l) { so = s_l; cgl = rl(w.cg, l); }
            }
            sj[it] = j;
            sv[it] = ((gptr_u32)c_skip)[cgl + 1u + (j - so)];
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) se[sj[it]] = sv[it];
    }
    lds_wait();
    // the round ends where the first list runs out of staged entries
    uint32_t last = WINF;
    if (rem && w.cg + w.q < w.ge) last = se[w.soff + w.q - 1u];
    uint32_t e = min(wave_min(last), dhi);
    e = max(e, F + 1u);                                // whatever happens above, a round makes progress (F < dhi: the caller's loop condition)
    // driver = the list with the most staged entries; a window = s of its blocks
    const uint32_t dkey = wave_max((w.q << 6) | (uint32_t)(63 - lane));
    const int drv = 63 - (int)(dkey & 63u);
    const uint32_t q_drv = dkey >> 6;
    const uint32_t so_drv = rl(w.soff, drv);
    const int l_act = __popcll(act_mask);
    // BASE windows of s driver blocks each (lane j: [B_j, B_j+1)), about half a row's worth; then consecutive base windows are
    // packed into ONE window as long as the EXACT block count — known here from the staged skip entries — fits a row of WCW
    // blocks.  (The first version cut windows of a fixed stride aimed SSW_SLACK blocks below WCW, because a window above WCW
    // takes the slow path: rows were 64 % full at config 3, and every slot of a row costs its load, its wait and its filter code
    // whether it holds a block or not.)
    uint32_t s = 1;
    if (q_drv) s = max(1u, (uint32_t)(((uint64_t)(WCW > l_act ? WCW - l_act : 1) * q_drv) / max(tot_q, 1u)) / 2u);
    const uint32_t nd_e = q_drv ? lds_lower_bound(se + so_drv, q_drv, e) : 0u;      // driver blocks that start inside (F, e)
    uint32_t nw = nd_e / s + 1u;
    if (nw > 63u) { nw = 63u; e = se[so_drv + 63u * s - 1u]; }
    // lane j <= nw: boundary B_j (B_0 = F, B_nw = e)
    uint32_t Bj = WINF;
    if ((uint32_t)lane <= nw) Bj = lane == 0 ? F : ((uint32_t)lane == nw ? e : se[so_drv + (uint32_t)lane * s - 1u]);
    // per list: entries < B_j (= first block of base window j, relative to the list's cursor block; also the cursor advance if the
    // round ends at B_j), and the blocks of base window j
    uint32_t fst[WL];
    uint32_t cnt = 0;
#pragma unroll
    for (int l = 0; l < WL; l++) {
        fst[l] = 0;
        if (l < L && ((act_mask >> l) & 1ull)) {
            const uint32_t n = rl(w.q, l), so = rl(w.soff, l);
            // Block i of the list (0 = the cursor block) holds docs from its first doc (the entry before E[i]) up to E[i]
            // INCLUSIVE: in a combined list the body and the title posting of one doc may sit on either side of a block
            // boundary.  So the blocks of [B_a, B_b) are i = (entries < B_a) .. (entries < B_b).
            const uint32_t lb = (uint32_t)lane <= nw ? lds_lower_bound(se + so, n, Bj) : 0u;     // entries < B_j
            const uint32_t lb_next = (uint32_t)__shfl_down((int)lb, 1, 64);
            fst[l] = lb;
            cnt += (uint32_t)lane < nw ? lb_next - lb + 1u : 0u;
        }
    }
    // packing: a run of base windows a..b holds sum(cnt) - (b - a) * l_act blocks (the boundary block of every list is
    // counted by both neighbours).  Greedy from the left; a base window that is oversize on its own stays alone.
    const uint32_t la = (uint32_t)l_act;
    const uint32_t pre = wave_excl_scan((uint32_t)lane < nw ? cnt - la : 0u, lane);                // lane nw: the total
    // run_end (every lane a: where the run that STARTS at a ends): the largest t in [a + 1, nw] with pre[t] <= pre[a] + WCW - l_act,
    // by a lane-parallel binary search over the other lanes' prefix values (six shuffles); the leaders are then reached by
    // following a -> run_end(a) + 1 from window 0 (one readlane per run; a scalar scan over all base windows cost 14k cycles a round)
    uint32_t run_end;
    {
        const uint32_t limit = pre + (uint32_t)SSW_PACK - min(la, (uint32_t)SSW_PACK);
        uint32_t lo = (uint32_t)lane + 1u, hi = max(nw, (uint32_t)lane + 1u);
#pragma unroll
        for (int it = 0; it < 6; it++) {
            const uint32_t mid = (lo + hi + 1u) >> 1;
            const uint32_t v = (uint32_t)__shfl((int)pre, (int)min(mid, 63u), 64);
            if (lo < hi) { if (v <= limit) lo = mid; else hi = mid - 1u; }
        }
        run_end = lo - 1u;
    }
    unsigned long long leaders = 0ull;
    for (uint32_t a = 0; a < nw; a = rl(run_end, (int)a) + 1u) leaders |= 1ull << a;
    bool leader = ((leaders >> lane) & 1ull) != 0ull;
    // a leader's window: [B_a, B_{end+1})
    uint32_t num[WL];
    cnt = 0;
    const int nxt_lane = (int)(leader ? run_end + 1u : (uint32_t)lane);
#pragma unroll
    for (int l = 0; l < WL; l++) {
        num[l] = 0;
        if (l < L && ((act_mask >> l) & 1ull)) {
            const uint32_t lb_end = (uint32_t)__shfl((int)fst[l], nxt_lane, 64);
            num[l] = leader ? lb_end - fst[l] + 1u : 0u;
            cnt += num[l];
        }
    }
    const uint32_t b_hi = (uint32_t)__shfl((int)Bj, nxt_lane, 64);
    // rows: one per regular window; an oversize window takes ceil(cnt / WCW) rows (its block list, walked by the slow path)
    uint32_t rows = leader ? (cnt + WCW - 1) / WCW : 0u;
    const uint32_t roff = wave_excl_scan(rows, lane);
    {
        // the windows that fit the row table are a prefix (roff is monotone; one window alone always fits)
        const unsigned long long bad = __ballot(leader && roff + rows > (uint32_t)WGMAX);
        if (bad) {
            uint32_t cut = (uint32_t)__ffsll((long long)bad) - 1u;                 // base window at which the first run that does not fit starts
            if (cut == 0u) cut = rl(run_end, 0) + 1u;
            nw = cut;
            e = rl(Bj, (int)nw);
            if ((uint32_t)lane >= nw) { leader = false; rows = 0; }
        }
    }
    const uint32_t n_rows = wave_sum(rows);
    if (leader) {
        uint32_t n = 0;
        const uint32_t row0 = roff;
#pragma unroll
        for (int l = 0; l < WL; l++) {
            if (l < L && ((act_mask >> l) & 1ull)) {
                const uint32_t cgl = rl(w.cg, l), g0l = rl(w.g0, l), g1l = rl(w.g1, l);
                for (uint32_t b = 0; b < num[l]; b++) {
                    const uint32_t g = cgl + fst[l] + b;
                    uint32_t d = (uint32_t)l | (g << 6);
                    if (g == g0l) d |= D_FIRST;
                    if (g == g1l) d |= D_LAST;
                    gdesc[row0 + n / WCW][n % WCW] = d;
                    n++;
                }
            }
        }
        for (uint32_t i = n; i < rows * WCW; i++) gdesc[row0 + i / WCW][i % WCW] = d_empty;
        for (uint32_t r = 0; r < rows; r++) {
            ghdr[row0 + r][0] = Bj;
            ghdr[row0 + r][1] = b_hi - Bj;
            ghdr[row0 + r][2] = r ? M_SKIP : (rows == 1 ? M_NORMAL : M_SLOW);
            ghdr[row0 + r][3] = cnt;
        }
    }
    if ((uint32_t)lane == nw) {
#pragma unroll
        for (int l = 0; l < WL; l++) adv_out[l] = fst[l];
    }
    // pad rows: the ring reads WDEPTH rows ahead of the row it processes
    for (int i = lane; i < 2 * WDEPTH * WCW; i += 64) gdesc[n_rows + i / WCW][i % WCW] = d_empty;
    if (lane < 2 * WDEPTH) { ghdr[n_rows + lane][0] = 0; ghdr[n_rows + lane][1] = 0; ghdr[n_rows + lane][2] = M_SKIP; ghdr[n_rows + lane][3] = 0; }
    lds_wait();
    // the exact stage's table lives where the skip entries were: all empty again
    for (int i = lane; i < 2 * WHT; i += 64) se[i] = EMPTY;
    lds_wait();
    return RoundPlan{n_rows, e};
}

}  // namespace

namespace ssw {

// one thread per (wave slice, term): list bounds, the slice's block range in the list (two searches in the skip index)
__global__ __launch_bounds__(256) void k_wave_prep(ScoreParams p, uint32_t n_slices, WPrep* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slices * (uint32_t)WL) return;
    const uint32_t si = i / WL, l = i % WL;
    const uint32_t slice_id = p.order[si];
    const SliceDesc sd = p.slices[slice_id];
    const uint32_t t0 = p.q_off[sd.q], nd = p.q_off[sd.q + 1] - t0;
    WPrep r{};
    if (l < nd) {
        const uint32_t term = p.dterm[t0 + l];
        const uint64_t p0 = p.c_ptr[term], p1 = p.c_ptr[term + 1];
        const bool active = p1 > p0;
        r.g0 = (uint32_t)(p0 >> 6);
        r.g1 = active ? (uint32_t)((p1 - 1) >> 6) : r.g0;
        r.lanes = (uint32_t)(p0 & 63) | ((active ? (uint32_t)((p1 - 1) & 63) + 1u : 0u) << 8) | ((active ? 1u : 0u) << 16);
        r.kth_b = p.b_kth[(size_t)term * KTH_N + p.kth_j];
        r.kth_t = p.t_kth[(size_t)term * KTH_N + p.kth_j];
        r.mult = p.dmult[t0 + l];
        uint32_t lo = 0, hi = 0;
        if (r.g1 > r.g0) {
            // entries c_skip[g0+1 .. g1] are first docs of the list's own blocks
            lo = sd.dlo == 0 ? 0u : skip_lower_bound(p.c_skip, r.g0 + 1, r.g1 + 1, sd.dlo);          // entries < dlo (a block that starts AT dlo may follow one that ends with dlo)
            hi = sd.dhi == WINF ? r.g1 - r.g0 : skip_lower_bound(p.c_skip, r.g0 + 1, r.g1 + 1, sd.dhi);   // entries < dhi
        }
        r.cg = r.g0 + lo;
        r.ge = r.g0 + hi;
        if (r.ge < r.cg) r.lanes &= 0xFFFFu;
    }
    out[i] = r;
}

__global__ __launch_bounds__(64, SSW_MINW) void k_score_wave(ScoreParams p, const WPrep* __restrict__ prep) {
    uint32_t* const ht_key = se;
    uint32_t* const ht_rec = se + WHT;

    const int lane = threadIdx.x;
    DIAG_NOW(t_w0);
#ifdef SSW_PHASES
    if (threadIdx.x < 8) ph_acc[threadIdx.x] = 0ull;
    if (threadIdx.x < 4) ph_rows[threadIdx.x] = 0u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    PH_MARK(0, -); PH_MARK(6, -);
    const uint32_t slice_id = p.order[blockIdx.x];
    const SliceDesc sd = p.slices[slice_id];
    const uint32_t q = sd.q;
    const uint32_t t0 = p.q_off[q], nd = p.q_off[q + 1] - t0;
    const int L = (int)nd;                             // <= WL (host): one combined list per distinct term

    WaveCtx C;
    C.sk = sk;
    C.ghdr = ghdr;
    C.gdesc = gdesc;
    C.tb = (uint64_t)p.c_rec;
    C.S.s_rec = reinterpret_cast<double2*>(pend_raw);
    C.S.pend = reinterpret_cast<uint4*>(pend_raw);
    C.S.l_mult = l_mult;
    C.S.ht_key = ht_key;
    C.S.ht_rec = ht_rec;
    C.S.overflow = &sc32[1];
    C.Q.qmag = p.qmag[q];
    C.Q.probs = (p.probs && p.prior) ? p.probs + (size_t)q * p.k_topics : nullptr;
    C.Q.sqd_ub = C.Q.probs ? p.sqd_ub[q] : 0.0;
    C.Q.sqd_ub_f = C.Q.probs ? __double2float_ru(C.Q.sqd_ub) : 0.0f;
    C.Q.qmag_f = (float)C.Q.qmag;
    C.r_ub = C.Q.probs ? __double2float_ru(33.0 * C.Q.sqd_ub * (1.0 + 0x1p-12)) : 0.0f;

    for (int i = lane; i < WSK + 4; i += 64) sk[i] = 0u;
    if (lane == 0) { sc32[0] = 0; sc32[1] = 0; sc64[0] = 0ull; *reinterpret_cast<float*>(&sc32[2]) = -INFINITY; }

    // ---- the lists (lane l: the combined list of distinct term l), resolved by k_wave_prep ----
    WList w{};
    w.lo_lane = 64;                                    // lanes that hold no list: no lane of a block is theirs (see plan_round: d_empty)
    float coef_raw_b = 0.f, coef_raw_t = 0.f, floor_l = 0.f;
    if (lane < L) {
        const WPrep r = prep[(size_t)blockIdx.x * WL + lane];
        w.active = (r.lanes >> 16) != 0;
        w.g0 = r.g0;
        w.g1 = r.g1;
        w.lo_lane = r.lanes & 0xFFu;
        w.hi_lane = (r.lanes >> 8) & 0xFFu;
        w.cg = r.cg;
        w.ge = r.ge;
        l_mult[lane] = r.mult;
        // filter coefficients and threshold floor exactly as in k_score_slices (get_metadata.go:57-58,69)
        const double share_b = 29.0 * (double)r.mult / C.Q.qmag, share_t = 38.0 * (double)r.mult / C.Q.qmag;
        coef_raw_b = __double2float_ru(share_b * (1.0 + 0x1p-12));
        coef_raw_t = __double2float_ru(share_t * (1.0 + 0x1p-12));
        if (r.kth_b > 0.0f) floor_l = fmaxf(floor_l, __double2float_rd(share_b * (1.0 - 0x1p-12) * (double)r.kth_b));
        if (r.kth_t > 0.0f) floor_l = fmaxf(floor_l, __double2float_rd(share_t * (1.0 - 0x1p-12) * (double)r.kth_t));
    }
    const float coef_raw = fmaxf(coef_raw_b, coef_raw_t);
    const float coef_max = __uint_as_float(wave_max((coef_raw > 0.0f && coef_raw < INFINITY) ? __float_as_uint(coef_raw) : 0u));
    C.fx_scale = coef_max > 0.0f ? (float)FX_ONE / coef_max : 1.0f;
    w.coef_b = coef_raw_b * C.fx_scale * (1.0f + 0x1p-20f);
    w.coef_t = coef_raw_t * C.fx_scale * (1.0f + 0x1p-20f);
    float thr0_f = __uint_as_float(wave_max(__float_as_uint(floor_l)));
#ifdef SS_EXP_FLOOR      // variant build only (tools/floor_exp.py)
    if (p.q_floor) thr0_f = fmaxf(thr0_f, p.q_floor[q]);
#endif
    const uint64_t thr0_key = thr0_f > 0.0f ? fkey((double)thr0_f) : 0ull;
    C.tk = TopK{cd_key, cd_doc, &sc32[0], &sc64[0], reinterpret_cast<float*>(&sc32[2]), thr0_key, thr0_f > 0.0f ? thr0_f : -INFINITY, (uint32_t)WCB};
    if (lane == 0 && thr0_f > 0.0f) { sc64[0] = thr0_key; *reinterpret_cast<float*>(&sc32[2]) = thr0_f; }
    if (lane == 0) {
        rs.Q = C.Q;
        rs.ra = RareArgs{p.t_mag, p.b_mag, p.prior, p.c_w, p.k_topics, p.k};
        rs.thr0_key = thr0_key;
        rs.thr0_f = thr0_f > 0.0f ? thr0_f : -INFINITY;
        rs.tb = C.tb;
        rs.r_ub = C.r_ub;
        rs.fx_scale = C.fx_scale;
    }
    const unsigned long long act_mask = __ballot(lane < L && w.active);
    lds_wait();

    DIAG_NOW(t_w1);
    PH_MARK(0, +);
    WDIAG_ADD(0, 1);
    WDIAG_ADD(10, t_w1 - t_w0);
    uint32_t F = sd.dlo;                               // frontier: docs below it are done
    uint32_t pend_n = 0;                               // pending survivors (wave-uniform)
    uint32_t thr_fx = wave_thr_fx(C);

    u32x2 rec[WDEPTH][WCW];
    uint32_t dvr[WDEPTH];
#ifdef SS_DIAG
    unsigned long long dg[4] = {0, 0, 0, 0};
#endif
    while (act_mask && F < sd.dhi) {
        DIAG_NOW(t_p0);
        PH_MARK(1, -);
        const RoundPlan rp = plan_round(w, p.c_skip, se, ghdr, gdesc, l_adv, L, act_mask, F, sd.dhi, lane);
        const uint32_t n_rows = rp.n_rows;
        DIAG_NOW(t_p1);
        PH_MARK(1, +); PH_MARK(2, -);
        WDIAG_ADD(1, 1);
        WDIAG_ADD(2, n_rows);
        WDIAG_ADD(11, t_p1 - t_p0);
        uint32_t r0 = 0;
        // ---- stream the round's rows.  The hot loop makes no call: a rare event (pending list nearly full, a window with
        //      more survivors than fit, an oversize window) leaves it, is handled below and the ring starts again behind it.
        while (r0 < n_rows) {
            enum : uint32_t { EV_NONE = 0, EV_FLUSH = 1, EV_OVERFLOW = 2, EV_SLOW = 3 };
            uint32_t ev = EV_NONE, ev_row = 0;
#ifdef SSW_ASM_RING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the ring counts from zero
#endif
            group_issue<0>(p, gdesc, r0, lane, rec, dvr);
            group_issue<1>(p, gdesc, r0 + 1, lane, rec, dvr);
#if SSW_DEPTH == 3
            group_issue<2>(p, gdesc, r0 + 2, lane, rec, dvr);
#endif
#if defined(SS_DIAG) && defined(SS_DIAG_LOOP)      // stamps inside the row loop cost more than the rows (2.4 ms per batch instead of 0.44)
#define WSTAMP(var) DIAG_NOW(var)
#define WACC(i, a, b) dg[i] += (b) - (a)
#else
#define WSTAMP(var) do { } while (0)
#define WACC(i, a, b) do { } while (0)
#endif
#define SSW_ADD(S_, c) h[c] = block_add<S_, c>(sk, w, dvr[S_], b_lo, span, lane, rec);
#define SSW_STEP(S_, RR)                                                                                                   \
            {                                                                                                              \
                const uint32_t r_ = (RR);                                                                                  \
                WSTAMP(ts0);                                                                                               \
                const uint4 hv = *reinterpret_cast<const uint4*>(ghdr[r_]);                                                \
                const uint32_t b_lo = rfl(hv.x), span = rfl(hv.y), mode = rfl(hv.z);                                       \
                if (mode == M_NORMAL) {                                                                                    \
                    if (pend_n > (uint32_t)(WPW - WPW / 4)) { ev = EV_FLUSH; ev_row = r_; goto ssw_event; }                 \
                    uint32_t h[WCW], u[WCW];                                                                               \
                    SSW_ADD(S_, 0) SSW_ADD(S_, 1) SSW_ADD(S_, 2) SSW_ADD(S_, 3) SSW_ADD(S_, 4) SSW_ADD(S_, 5) SSW_ADD(S_, 6) SSW_ADD(S_, 7) \
                    SSW_ADD(S_, 8) SSW_ADD(S_, 9) SSW_ADD(S_, 10) SSW_ADD(S_, 11) SSW_ADD(S_, 12) SSW_ADD(S_, 13) SSW_ADD(S_, 14) SSW_ADD(S_, 15) \
                    WSTAMP(ts1);                                                                                           \
                    _Pragma("unroll") for (int c = 0; c < WCW; c++) u[c] = *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(sk) + h[c]); \
                    uint32_t tot = 0;                                                                                      \
                    _Pragma("unroll") for (int c = 0; c < WCW; c++) tot += (uint32_t)__popcll(__ballot(u[c] >= thr_fx));   \
                    WSTAMP(ts2);                                                                                           \
                    WACC(0, ts0, ts1); WACC(1, ts1, ts2);                                                                  \
                    PH_COUNT(tot);                                                                                         \
                    if (tot) {                                                                                             \
                        if (pend_n + tot > (uint32_t)WPW) { ev = EV_OVERFLOW; ev_row = r_; goto ssw_event; }               \
                        _Pragma("unroll") for (int c = 0; c < WCW; c++) {                                                  \
                            const bool surv = u[c] >= thr_fx;                                                              \
                            const unsigned long long m = __ballot(surv);                                                   \
                            if (m) {                                                                                       \
                                const uint32_t d = rl(dvr[S_], c);                                                         \
                                const int l = (int)(d & 15u);                                                              \
                                if (surv) C.S.pend[pend_n + bits_below(m)] = make_uint4(rec[S_][c].x & 0x7FFFFFFFu, (d >> 6) * 64u + (uint32_t)lane, (uint32_t)l, rec[S_][c].x >> 31); \
                                pend_n += (uint32_t)__popcll(m);                                                           \
                            }                                                                                              \
                        }                                                                                                  \
                    }                                                                                                      \
                    /* the whole sketch back to zero: four 16-byte stores per lane (the same 4 KB as one 4-byte store per slot and lane, in 4 instructions instead of 48) */ \
                    { uint4* const s4_ = reinterpret_cast<uint4*>(sk); const uint4 z_ = make_uint4(0u, 0u, 0u, 0u);       \
                      _Pragma("unroll") for (int j_ = 0; j_ < WSK / 256; j_++) s4_[lane + 64 * j_] = z_; }                  \
                    WSTAMP(ts3);                                                                                           \
                    WACC(2, ts2, ts3);                                                                                     \
                } else {                                                                                                   \
                    if (mode == M_SLOW) { ev = EV_SLOW; ev_row = r_; goto ssw_event; }                                     \
                    /* a row without work (pad row, tail of an oversize window): its dummy loads still count */           \
                    ring_drain_slot<(WDEPTH - 1) * WCW>();                                                                 \
                }                                                                                                          \
                WSTAMP(ts4);                                                                                               \
                group_issue<S_>(p, gdesc, r_ + WDEPTH, lane, rec, dvr);                                                 \
                WSTAMP(ts5);                                                                                               \
                WACC(3, ts4, ts5);                                                                                         \
            }
            {
                uint32_t r = r0;
                for (; r < n_rows; r += WDEPTH) {
                    SSW_STEP(0, r)
                    SSW_STEP(1, r + 1)
#if SSW_DEPTH == 3
                    SSW_STEP(2, r + 2)
#endif
                }
            }
#undef SSW_STEP
#undef SSW_ADD
            r0 = n_rows;
        ssw_event:
            // whatever the ring still has in flight lands in registers the compiler must not have reused yet
#ifdef SSW_ASM_RING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#pragma unroll
            for (int s_ = 0; s_ < WDEPTH; s_++)
#pragma unroll
                for (int c = 0; c < WCW; c++) ring_keep(rec[s_][c]);
            if (ev != EV_NONE) {
                WDIAG_ADD(3 + ev, 1);
                WDIAG_ADD(7, pend_n);
                DIAG_NOW(t_e0);
                PH_MARK(3, -);
                const uint4 hv = *reinterpret_cast<const uint4*>(ghdr[ev_row]);
                const uint32_t b_lo = rfl(hv.x), span = rfl(hv.y), n_blk = rfl(hv.w);
#ifndef SSW_EXP_NOFLUSH
                if (pend_n) wave_flush(lane, pend_n);
#endif
                pend_n = 0;
                r0 = ev_row;                            // EV_FLUSH: the row has not been touched: it runs again
                if (ev == EV_OVERFLOW) { slow_window(w, lane, ev_row, n_blk, b_lo, span, false); r0 = ev_row + 1; }
                if (ev == EV_SLOW) { slow_window(w, lane, ev_row, n_blk, b_lo, span, true); r0 = ev_row + 1; }
                thr_fx = wave_thr_fx(C);
                DIAG_NOW(t_e1);
                PH_MARK(3, +);
                WDIAG_ADD(12, t_e1 - t_e0);
            }
        }
        DIAG_NOW(t_p2);
        PH_MARK(2, +);
        WDIAG_ADD(13, t_p2 - t_p1);
        // ---- next round starts at e ----
        if (lane < L && w.active) w.cg += l_adv[lane];
        F = rp.e;
        lds_wait();
    }
    WDIAG_ADD(7, pend_n);
    DIAG_NOW(t_ff0);
    PH_MARK(4, -);
#if !defined(SSW_EXP_NOFINAL) && !defined(SSW_EXP_NOFLUSH)      // (timing experiments only: wrong results)
    if (pend_n) wave_flush(lane, pend_n);
#endif
    DIAG_NOW(t_w2);
    PH_MARK(4, +); PH_MARK(5, -);
    WDIAG_ADD(20, t_w2 - t_ff0);
    WDIAG_ADD(24, pend_n ? 1 : 0);

    // hand the candidates in: appended to the query's list (k_merge_flat sorts; a slice sorts only if it holds more than k)
    if (sc32[0] > (uint32_t)p.k) wave_compact(p.k);
    lds_wait();
    const uint32_t n_out = min(sc32[0], (uint32_t)p.k);
    uint32_t pos = 0;
    if (lane == 0 && n_out) pos = atomicAdd(&p.qc_cnt[q], n_out);
    pos = rfl(pos);
    const size_t base = (size_t)p.slice_base[q] * p.k + pos;
    for (uint32_t i = lane; i < n_out; i += 64) {
        p.so_key[base + i] = cd_key[i];
        p.so_doc[base + i] = cd_doc[i];
    }
    DIAG_NOW(t_w3);
#ifdef SSW_PHASES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PH_MARK(5, +); PH_MARK(6, +);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) {
        for (int i = 0; i < 7; i++) atomicAdd(&g_wphase[blockIdx.x & 255][i], ph_acc[i]);
        atomicAdd(&g_wphase[blockIdx.x & 255][7], 1ull);
        for (int i = 0; i < 4; i++) atomicAdd(&g_wrows[blockIdx.x & 255][i], (unsigned long long)ph_rows[i]);
    }
#endif
    WDIAG_ADD(8, n_out);
#ifdef SS_DIAG
    WDIAG_ADD(16, dg[0]); WDIAG_ADD(17, dg[1]); WDIAG_ADD(18, dg[2]); WDIAG_ADD(19, dg[3]);
#endif
    WDIAG_ADD(14, t_w3 - t_w2);
    WDIAG_ADD(15, t_w3 - t_w0);
}

}  // namespace ssw

namespace ss {
// prep: device workspace of score_wave_prep_bytes(n_slices) bytes
size_t score_wave_prep_bytes(unsigned n_slices) { return (size_t)n_slices * WL * sizeof(WPrep); }
void launch_wave_prep(const void* params, unsigned n_slices, void* prep, hipStream_t st) {
    const ScoreParams& p = *reinterpret_cast<const ScoreParams*>(params);
    hipLaunchKernelGGL(ssw::k_wave_prep, dim3((n_slices * WL + 255) / 256), dim3(256), 0, st, p, n_slices, reinterpret_cast<WPrep*>(prep));
}
void launch_score_wave(const void* params, unsigned n_slices, const void* prep, hipStream_t st) {
    const ScoreParams& p = *reinterpret_cast<const ScoreParams*>(params);
    hipLaunchKernelGGL(ssw::k_score_wave, dim3(n_slices), dim3(64), 0, st, p, reinterpret_cast<const WPrep*>(prep));
}
int score_wave_max_lists() { return WL; }
int score_wave_max_k() { return WCB / 2; }
void score_wave_diag_dump() {
#ifdef SSW_PHASES
    {
        static unsigned long long hh[256][8];
        unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_wphase), sizeof(hh)) == hipSuccess) {
            for (int r = 0; r < 256; r++)
                for (int i = 0; i < 8; i++) h[i] += hh[r][i];
            const char* names[8] = {"setup", "plan", "stream_incl_events", "events", "final_flush", "hand_in", "slice", "slices"};
            static unsigned long long rr[256][4];
            unsigned long long r4[4] = {0, 0, 0, 0};
            if (hipMemcpyFromSymbol(rr, HIP_SYMBOL(g_wrows), sizeof(rr)) == hipSuccess)
                for (int r = 0; r < 256; r++) for (int i = 0; i < 4; i++) r4[i] += rr[r][i];
            fprintf(stderr, "[ss phases] rows=%llu rows_with_survivors=%llu survivors=%llu rows_with_more_than_4=%llu\n", r4[0], r4[1], r4[2], r4[3]);
            fprintf(stderr, "[ss phases] k_score_wave, s_memtime ticks summed over slices:");
            for (int i = 0; i < 8; i++) fprintf(stderr, " %s=%llu", names[i], h[i]);
            fprintf(stderr, "\n");
        }
    }
#endif
#ifdef SSW_CHECK
    {
        uint32_t c[8];
        if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_wcheck), sizeof(c)) == hipSuccess)
            fprintf(stderr, "[ss check] k_score_wave: bad block indices=%u (last desc 0x%x limit %u slice %u)\n", c[0], c[1], c[2], c[3]);
    }
#endif
#ifdef SS_DIAG
    static unsigned long long hw[256][32];
    unsigned long long h[32] = {};
    if (hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_wdiag), sizeof(hw)) == hipSuccess) {
        for (int r = 0; r < 256; r++)
            for (int i = 0; i < 32; i++) h[i] += hw[r][i];
        const char* names[25] = {"slices", "rounds", "rows", "-", "ev_flush", "ev_overflow", "ev_slow", "flushed_records", "handed_in", "blocks",
                                 "cyc_setup", "cyc_plan", "cyc_events", "cyc_stream", "cyc_epilogue", "cyc_total", "cyc_row_add", "cyc_row_read",
                                 "cyc_row_append_clear", "cyc_row_issue", "cyc_final_flush", "cyc_flush_loads", "cyc_flush_hash", "cyc_flush_score", "final_flushes"};
        fprintf(stderr, "[ss diag] k_score_wave (lane 0 of every slice):");
        for (int i = 0; i < 25; i++) fprintf(stderr, " %s=%llu", names[i], h[i]);
        fprintf(stderr, "\n");
    }
#endif
}
}  // namespace ss
