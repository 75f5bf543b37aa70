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
#ifndef SSW_MINW
#define SSW_MINW 2
#endif
#ifndef SSW_DEPTH
#define SSW_DEPTH 2
#endif
constexpr int WDEPTH = SSW_DEPTH;       // groups whose loads are in flight or in registers (2 or 3)
constexpr int WSK = 1024;               // sketch slots; slot WSK is a dummy that always holds 0
constexpr int WSE = 480;                // skip entries staged per round
constexpr int WGMAX = 64;               // group rows per round; 2 * ceil((WSE + WL) / WCW) <= WGMAX
constexpr int WPW = 128;                // pending survivors = capacity of the wave's exact stage
constexpr int WHT = 256;                // exact-stage hash slots
constexpr int WL = 12;                  // lists per query
constexpr int WCB = 256;                // candidate buffer (>= 2k)
constexpr uint32_t WINF = 0xFFFFFFFFu;
constexpr uint32_t D_EMPTY = 0xFFFFu;   // group slot without a block
constexpr uint32_t D_FIRST = 0x10u, D_LAST = 0x20u;    // the block is the first / last one of its list: part of it belongs to a neighbouring list
constexpr uint32_t FX_CLAMP_SLOW = 1u << 17;           // oversize windows: (WSE + WL) * 64 records of this much stay below 2^32
static_assert((uint64_t)(WSE + WL) * 64 * FX_CLAMP_SLOW < (1ull << 32), "slow-path sums must not wrap");
static_assert((uint64_t)WCW * 64 * FX_CLAMP < (1ull << 32), "fast-path sums must not wrap");
static_assert(2 * ((WSE + WL + WCW - 1) / WCW) <= WGMAX, "a round's plan must fit its rows");
enum : uint32_t { M_NORMAL = 1, M_SLOW = 2, M_SKIP = 0 };

#ifdef SS_DIAG
__device__ unsigned long long g_wdiag[32];
#define WDIAG_ADD(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_wdiag[i], (unsigned long long)(v)); } while (0)
#else
#define WDIAG_ADD(i, v) do { } while (0)
#endif

__device__ __forceinline__ uint32_t rl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

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

// number of entries of the global u32 array a[lo, hi) (ascending) that are < v, as an offset from lo: interpolation
// steps with two independent probes each, then an 8-ary finish (lower_bound_interp of score_common.hpp for 4-byte entries)
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

__device__ __forceinline__ uint32_t w_slot(uint32_t doc) { return (doc ^ (doc >> 10)) & (uint32_t)(WSK - 1); }

// What a slice needs to know about one of its lists, resolved by k_wave_prep for all slices of the batch at once (the
// chains of dependent loads and the searches in the skip index then run side by side instead of at the head of every slice).
struct __attribute__((aligned(16))) WPrep {
    uint32_t g0, g1;          // first / last block of the list
    uint32_t cg, ge;          // blocks that hold the slice's doc range
    uint32_t p0_lo;           // first posting of the list (low 32 bits)
    uint32_t lanes;           // lo_lane | hi_lane << 8 | active << 16
    float kth;                // k'-th largest impact of the list (threshold floor)
    uint32_t mult;
};

struct WaveLds {
    double2* s_rec;      // [WPW] exact stage: {addend, magnitude}
    uint4* pend;         // [WPW] the same bytes while pending: {doc, index in list, list, -}
    uint64_t* l_w;       // [WL] address of the list's first float32 weight
    uint32_t* l_mult;    // [WL]
    uint32_t* l_field;   // [WL]
    uint32_t* ht_key;    // [WHT]
    uint32_t* ht_rec;    // [WHT]
    uint32_t* overflow;
};

// ---- exact stage of one wave (flush_pending / score_owned of score.hip for 64 threads) ------------------------------
__device__ __forceinline__ void wave_score_owned(const WaveLds& S, const TopK& tk, const SliceQuery& Q, const ScoreParams& p, int lane,
                                                 uint32_t doc, uint32_t slot) {
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
            if (fkey(fin) >= thr0 || fin != fin) final_rank(T, B, mt, mb, Q.qmag, topic_dot(p.prior, Q.probs, p.k_topics, e_doc), title, body, fin);
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
        topk_compact(tk, p.k);
        if (lane == 0) *S.overflow = 0;
        lds_wait();
    }
}

__device__ __noinline__ void wave_flush(const WaveLds S, const TopK tk, const SliceQuery Q, const ScoreParams& p, int lane, uint32_t n) {
    uint32_t pdoc[2], pl[2], own[2];
    float pw[2];
    double pm[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint32_t i = lane + r * 64;
        pl[r] = EMPTY;
        pdoc[r] = 0;
        pw[r] = 0.f;
        pm[r] = 1.0;
        if (i < n) {
            const uint4 e = S.pend[i];
            pdoc[r] = e.x;
            pl[r] = e.z;
            pw[r] = load_w(S.l_w[e.z], e.y);
            pm[r] = (S.l_field[e.z] ? p.t_mag : p.b_mag)[e.x];
        }
    }
    lds_wait();                                     // every pending entry has been read: the bytes may be rewritten
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint32_t l = pl[r];
        own[r] = EMPTY;
        if (l != EMPTY) {
            const uint32_t i = lane + r * 64;
            uint32_t h = (pdoc[r] * 2654435761u) >> 24;             // 8 hash bits -> [0, WHT)
            for (;;) {
                const uint32_t prev = atomicCAS(&S.ht_key[h], EMPTY, pdoc[r]);
                if (prev == EMPTY) own[r] = h;
                if (prev == EMPTY || prev == pdoc[r]) break;
                h = (h + 1) & (uint32_t)(WHT - 1);
            }
            const uint32_t field = S.l_field[l];
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
#pragma unroll
    for (int r = 0; r < 2; r++) wave_score_owned(S, tk, Q, p, lane, pdoc[r], own[r]);
}

// per-list constants, lane l < L holds list l
struct WList {
    uint32_t tb_lo, tb_hi;      // address of the table's record array (t_rec or b_rec)
    uint32_t g0, g1;            // first / last 64-record block of the list
    uint32_t lo_lane, hi_lane;  // lanes of block g0 below lo_lane and lanes of block g1 from hi_lane on belong to other lists
    uint32_t ge;                // last block that can hold a doc of the slice
    uint32_t cg;                // block that holds the frontier
    uint32_t q, soff;           // this round: skip entries staged, where in `se`
    uint32_t pos0;              // 64 * cg - (first posting of the list): index in the list of lane 0 of block cg (may wrap for block g0)
    uint32_t rb_lo, rb_hi;      // address of block cg
    float coef;                 // filter coefficient in fixed-point units
    bool active;
};

__device__ __forceinline__ void set_round_base(WList& w, uint32_t p0_lo) {
    const uint64_t a = (((uint64_t)w.tb_hi << 32) | w.tb_lo) + (uint64_t)w.cg * 512u;
    w.rb_lo = (uint32_t)a;
    w.rb_hi = (uint32_t)(a >> 32);
    w.pos0 = w.cg * 64u - p0_lo;
}

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

__device__ __forceinline__ uint64_t block_addr(uint32_t d, const WList& w) {
    const int l = (int)(d & 15u);
    return (((uint64_t)rl(w.rb_hi, l) << 32) | rl(w.rb_lo, l)) + (uint64_t)(d >> 6) * 512u;
}

struct WaveCtx {
    uint32_t* sk;
    uint32_t (*ghdr)[4];
    uint16_t (*gdesc)[WCW];
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
__device__ __noinline__ void slow_window(const WaveCtx C, const ScoreParams& p, const WList w, int lane, uint32_t row, uint32_t n_blocks,
                                         uint32_t b_lo, uint32_t span, bool need_add) {
    const uint32_t clamp = need_add ? FX_CLAMP_SLOW : FX_CLAMP;
    if (need_add) {
        for (uint32_t i = 0; i < n_blocks; i++) {
            const uint32_t d = rfl(C.gdesc[row + i / WCW][i % WCW]);
            const u32x2 rec = *(gptr_u2)(block_addr(d, w) + (uint32_t)lane * 8u);
            const float coef = __uint_as_float(rl(__float_as_uint(w.coef), (int)(d & 15u)));
            if (block_active(d, rec.x, b_lo, span, w, lane))
                atomicAdd(&C.sk[w_slot(rec.x)], min(fx_share(__uint_as_float(rec.y), coef), clamp));
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
                const u32x2 rec = *(gptr_u2)(block_addr(d, w) + (uint32_t)lane * 8u);
                const bool act = block_active(d, rec.x, b_lo, span, w, lane) && (rec.x - (b_lo + cur)) < sub;
                const uint32_t u = act ? C.sk[w_slot(rec.x)] : 0u;
                cnt += (uint32_t)__popcll(__ballot(act && u >= thr));
            }
            if (cnt <= (uint32_t)WPW || sub == 1) break;
            sub = (sub + 1) >> 1;
        }
        uint32_t n = 0;
        for (uint32_t i = 0; i < n_blocks; i++) {
            const uint32_t d = rfl(C.gdesc[row + i / WCW][i % WCW]);
            const u32x2 rec = *(gptr_u2)(block_addr(d, w) + (uint32_t)lane * 8u);
            const bool act = block_active(d, rec.x, b_lo, span, w, lane) && (rec.x - (b_lo + cur)) < sub;
            const uint32_t u = act ? C.sk[w_slot(rec.x)] : 0u;
            const bool surv = act && u >= thr;
            const unsigned long long m = __ballot(surv);
            if (m) {
                const int l = (int)(d & 15u);
                const uint32_t idx0 = rl(w.pos0, l) + (d >> 6) * 64u;
                const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
                const uint32_t pos = n + (uint32_t)__popcll(m & below);
                if (surv && pos < (uint32_t)WPW) C.S.pend[pos] = make_uint4(rec.x, idx0 + (uint32_t)lane, (uint32_t)l, 0u);
                n += (uint32_t)__popcll(m);
            }
        }
        lds_wait();
        if (n) wave_flush(C.S, C.tk, C.Q, p, lane, min(n, (uint32_t)WPW));
        cur += sub;
    }
    if (need_add) {
        for (int i = lane; i < WSK; i += 64) C.sk[i] = 0u;
    } else {
        // the caller's adds: clear exactly the slots of the window's records
        for (uint32_t i = 0; i < n_blocks; i++) {
            const uint32_t d = rfl(C.gdesc[row + i / WCW][i % WCW]);
            const u32x2 rec = *(gptr_u2)(block_addr(d, w) + (uint32_t)lane * 8u);
            if (block_active(d, rec.x, b_lo, span, w, lane)) C.sk[w_slot(rec.x)] = 0u;
        }
    }
    lds_wait();
}

// The ring's loads are hidden from the compiler (inline asm): its s_waitcnt bookkeeping drains a 48-deep ring at every
// step (it emitted vmcnt(15..0) where vmcnt(47..32) is right), which exposes one memory latency per row.  So the loads
// are counted by hand: a group is ALWAYS WCW loads, nothing else issues vector-memory operations inside the hot loop,
// and block c of the group being processed has exactly (WCW - 1 - c) + (WDEPTH - 1) * WCW younger loads in flight.
// ring_wait<N> names the destination "+v" so that no consumer is scheduled above it (cdna_hip_programming.md §5.7 (ii)).
#ifdef SSW_ASM_RING
template <int N>
__device__ __forceinline__ void ring_wait(u32x2& r) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N)); }
__device__ __forceinline__ void ring_keep(u32x2& r) { asm volatile("" : "+v"(r)); }
#else
template <int N>
__device__ __forceinline__ void ring_wait(u32x2&) {}
__device__ __forceinline__ void ring_keep(u32x2&) {}
#endif

template <int S_, int C_>
__device__ __forceinline__ void block_issue(const WList& w, uint32_t dv, int lane, uint64_t dummy, u32x2 (&rec)[WDEPTH][WCW]) {
    const uint32_t d = rl(dv, C_);
    uint64_t base = dummy;
    uint32_t voff = 0;
    if (d != D_EMPTY) {
        base = block_addr(d, w);
        voff = (uint32_t)lane * 8u;
    }
#ifdef SSW_ASM_RING
    const uint64_t addr = base + voff;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rec[S_][C_]) : "v"(addr));
#else
    rec[S_][C_] = *(gptr_u2)(base + voff);
#endif
}
template <int S_>
__device__ __forceinline__ void group_issue(const WaveCtx& C, const WList& w, uint32_t row, int lane, uint64_t dummy,
                                            u32x2 (&rec)[WDEPTH][WCW], uint32_t (&dvr)[WDEPTH]) {
    const uint32_t dv = C.gdesc[row][lane & (WCW - 1)];
    dvr[S_] = dv;
    block_issue<S_, 0>(w, dv, lane, dummy, rec);  block_issue<S_, 1>(w, dv, lane, dummy, rec);
    block_issue<S_, 2>(w, dv, lane, dummy, rec);  block_issue<S_, 3>(w, dv, lane, dummy, rec);
    block_issue<S_, 4>(w, dv, lane, dummy, rec);  block_issue<S_, 5>(w, dv, lane, dummy, rec);
    block_issue<S_, 6>(w, dv, lane, dummy, rec);  block_issue<S_, 7>(w, dv, lane, dummy, rec);
    block_issue<S_, 8>(w, dv, lane, dummy, rec);  block_issue<S_, 9>(w, dv, lane, dummy, rec);
    block_issue<S_, 10>(w, dv, lane, dummy, rec); block_issue<S_, 11>(w, dv, lane, dummy, rec);
    block_issue<S_, 12>(w, dv, lane, dummy, rec); block_issue<S_, 13>(w, dv, lane, dummy, rec);
    block_issue<S_, 14>(w, dv, lane, dummy, rec); block_issue<S_, 15>(w, dv, lane, dummy, rec);
    static_assert(WCW == 16, "group_issue is written out for 16 block slots");
}

// block C_ of ring slot S_: wait for its records, add every record of the window into the sketch; returns the slot (WSK: none)
template <int S_, int C_>
__device__ __forceinline__ uint32_t block_add(uint32_t* sk, const WList& w, uint32_t dv, uint32_t b_lo, uint32_t span, int lane,
                                              u32x2 (&rec)[WDEPTH][WCW]) {
    ring_wait<(WCW - 1 - C_) + (WDEPTH - 1) * WCW>(rec[S_][C_]);
    const uint32_t d = rl(dv, C_);
    uint32_t h = (uint32_t)WSK;
    if (d != D_EMPTY) {
        const float coef = __uint_as_float(rl(__float_as_uint(w.coef), (int)(d & 15u)));
        if (block_active(d, rec[S_][C_].x, b_lo, span, w, lane)) {
            h = w_slot(rec[S_][C_].x);
            atomicAdd(&sk[h], fx_share(__uint_as_float(rec[S_][C_].y), coef));
        }
    }
    return h;
}

// ---- planning of one round (not hot: once per ~WSE blocks) ---------------------------------------------------------
// Stages the skip entries of the blocks after every list's cursor, cuts [F, e) into windows of `s` driver blocks and
// writes the rows (ghdr / gdesc) the streaming loop reads.  Returns {n_rows, e}; adv_out[l] = how many blocks list l's
// cursor advances when the next round starts at e.
struct RoundPlan { uint32_t n_rows, e; };
__device__ __noinline__ RoundPlan plan_round(WList w, const ScoreParams& p, uint32_t* se, uint32_t (*ghdr)[4], uint16_t (*gdesc)[WCW],
                                             uint32_t* adv_out, int L, unsigned long long act_mask, uint32_t F, uint32_t dhi, int lane) {
    const uint32_t rem = w.active && lane < L ? w.ge - w.cg + 1u : 0u;
    const uint32_t tot_rem = wave_sum(rem);
    uint32_t ql = 0;
    if (rem > 1) ql = min(rem - 1u, max(1u, (uint32_t)(((uint64_t)(WSE - WL) * rem) / tot_rem)));
    w.q = ql;
    w.soff = wave_excl_scan(ql, lane);
    for (int l = 0; l < L; l++) {
        const uint32_t n = rl(w.q, l), so = rl(w.soff, l), cgl = rl(w.cg, l);
        const uint32_t* sp = (l & 1) ? p.t_skip : p.b_skip;
        for (uint32_t i = lane; i < n; i += 64) se[so + i] = sp[cgl + 1u + i];
    }
    lds_wait();
    // the round ends where the first list runs out of staged entries
    uint32_t last = WINF;
    if (rem && w.cg + w.q < w.ge) last = se[w.soff + w.q - 1u];
    uint32_t e = min(wave_min(last), dhi);
    // driver = the list with the most staged entries; a window = s of its blocks
    const uint32_t dkey = wave_max((w.q << 6) | (uint32_t)(63 - lane));
    const int drv = 63 - (int)(dkey & 63u);
    const uint32_t q_drv = dkey >> 6, tot_q = wave_sum(w.q);
    const uint32_t so_drv = rl(w.soff, drv);
    const int l_act = __popcll(act_mask);
    uint32_t s = 1;
    if (q_drv) s = max(1u, (uint32_t)(((uint64_t)(WCW > l_act ? WCW - l_act : 1) * q_drv) / max(tot_q, 1u)));
    const uint32_t nd_e = q_drv ? lds_lower_bound(se + so_drv, q_drv, e) : 0u;      // driver blocks that start inside (F, e)
    uint32_t nw = nd_e / s + 1u;
    if (nw > 63u) { nw = 63u; e = se[so_drv + 63u * s - 1u]; }
    // lane j <= nw: boundary B_j (B_0 = F, B_nw = e)
    uint32_t Bj = WINF;
    if ((uint32_t)lane <= nw) Bj = lane == 0 ? F : ((uint32_t)lane == nw ? e : se[so_drv + (uint32_t)lane * s - 1u]);
    // per list: blocks [first, first + num) of window j, relative to the list's cursor block
    uint32_t fst[WL], num[WL];
    uint32_t cnt = 0;
#pragma unroll
    for (int l = 0; l < WL; l++) {
        fst[l] = 0; num[l] = 0;
        if (l < L && ((act_mask >> l) & 1ull)) {
            const uint32_t n = rl(w.q, l), so = rl(w.soff, l);
            const uint32_t lb = (uint32_t)lane <= nw ? lds_lower_bound(se + so, n, Bj) : 0u;     // entries < B_j
            const uint32_t ub = lb + ((lb < n && se[so + lb] == Bj) ? 1u : 0u);                   // entries <= B_j
            const uint32_t lb_next = (uint32_t)__shfl_down((int)lb, 1, 64);
            fst[l] = ub;                                                                       // also: cursor advance if the round ends at B_j
            num[l] = (uint32_t)lane < nw ? lb_next - ub + 1u : 0u;
            cnt += num[l];
        }
    }
    // rows: one per regular window; an oversize window takes ceil(cnt / WCW) rows (its block list, walked by the slow path)
    const uint32_t rows = (uint32_t)lane < nw ? (cnt + WCW - 1) / WCW : 0u;
    const uint32_t roff = wave_excl_scan(rows, lane);
    {
        // windows 0 .. nw2-1 fit (roff is monotone, so the fitting windows are a prefix; one window alone always fits)
        const uint32_t nw2 = (uint32_t)__popcll(__ballot((uint32_t)lane < nw && roff + rows <= (uint32_t)WGMAX));
        if (nw2 < nw) { nw = max(nw2, 1u); e = rl(Bj, (int)nw); }
    }
    const uint32_t n_rows = rl(roff + rows, (int)nw - 1);
    const uint32_t b_hi = (uint32_t)__shfl_down((int)Bj, 1, 64);
    if ((uint32_t)lane < nw) {
        uint32_t n = 0;
        const uint32_t row0 = roff;
#pragma unroll
        for (int l = 0; l < WL; l++) {
            if (l < L && ((act_mask >> l) & 1ull)) {
                const uint32_t cgl = rl(w.cg, l), g0l = rl(w.g0, l), g1l = rl(w.g1, l);
                for (uint32_t b = 0; b < num[l]; b++) {
                    const uint32_t rel = fst[l] + b;
                    uint32_t d = (uint32_t)l | (rel << 6);
                    if (cgl + rel == g0l) d |= D_FIRST;
                    if (cgl + rel == g1l) d |= D_LAST;
                    gdesc[row0 + n / WCW][n % WCW] = (uint16_t)d;
                    n++;
                }
            }
        }
        for (uint32_t i = n; i < rows * WCW; i++) gdesc[row0 + i / WCW][i % WCW] = (uint16_t)D_EMPTY;
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
    for (int i = lane; i < 2 * WDEPTH * WCW; i += 64) gdesc[n_rows + i / WCW][i % WCW] = (uint16_t)D_EMPTY;
    if (lane < 2 * WDEPTH) { ghdr[n_rows + lane][0] = 0; ghdr[n_rows + lane][1] = 0; ghdr[n_rows + lane][2] = M_SKIP; ghdr[n_rows + lane][3] = 0; }
    lds_wait();
    return RoundPlan{n_rows, e};
}

}  // namespace

namespace ssw {

// one thread per (wave slice, list): list bounds, the slice's block range in the list (two searches in the skip index)
__global__ __launch_bounds__(256) void k_wave_prep(ScoreParams p, uint32_t n_slices, WPrep* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slices * (uint32_t)WL) return;
    const uint32_t si = i / WL, l = i % WL;
    const uint32_t slice_id = p.order[si];
    const SliceDesc sd = p.slices[slice_id];
    const uint32_t t0 = p.q_off[sd.q], nd = p.q_off[sd.q + 1] - t0;
    WPrep r{};
    if (l < 2 * nd) {
        const int field = l & 1;
        const uint32_t term = p.dterm[t0 + (l >> 1)];
        const uint64_t* ptr = field ? p.t_ptr : p.b_ptr;
        const uint64_t p0 = ptr[term], p1 = ptr[term + 1];
        const uint32_t* sp = field ? p.t_skip : p.b_skip;
        const bool active = p1 > p0;
        r.g0 = (uint32_t)(p0 >> 6);
        r.g1 = active ? (uint32_t)((p1 - 1) >> 6) : r.g0;
        r.p0_lo = (uint32_t)p0;
        r.lanes = (uint32_t)(p0 & 63) | ((active ? (uint32_t)((p1 - 1) & 63) + 1u : 0u) << 8) | ((active ? 1u : 0u) << 16);
        r.kth = (field ? p.t_kth : p.b_kth)[(size_t)term * KTH_N + p.kth_j];
        r.mult = p.dmult[t0 + (l >> 1)];
        uint32_t lo = 0, hi = 0;
        if (r.g1 > r.g0) {
            // entries skip[g0+1 .. g1] are first docs of the list's own blocks
            lo = sd.dlo == 0 ? 0u : skip_lower_bound(sp, r.g0 + 1, r.g1 + 1, sd.dlo + 1u);     // entries <= dlo
            hi = sd.dhi == WINF ? r.g1 - r.g0 : skip_lower_bound(sp, r.g0 + 1, r.g1 + 1, sd.dhi);   // entries < dhi
        }
        r.cg = r.g0 + lo;
        r.ge = r.g0 + hi;
        if (r.ge < r.cg) r.lanes &= 0xFFFFu;
    }
    out[i] = r;
}

__global__ __launch_bounds__(64, SSW_MINW) void k_score_wave(ScoreParams p, const WPrep* __restrict__ prep) {
    __shared__ uint32_t sk[WSK + 64];
    __shared__ uint32_t se[WSE + 32];
    __shared__ __attribute__((aligned(16))) uint32_t ghdr[WGMAX + 2 * WDEPTH][4];
    __shared__ uint16_t gdesc[WGMAX + 2 * WDEPTH][WCW];
    __shared__ __attribute__((aligned(16))) unsigned char pend_raw[WPW * 16];
    __shared__ uint32_t ht_key[WHT], ht_rec[WHT];
    __shared__ uint64_t cd_key[WCB];
    __shared__ uint32_t cd_doc[WCB];
    __shared__ uint64_t l_w[WL];
    __shared__ uint32_t l_mult[WL], l_field[WL], l_adv[WL];
    __shared__ uint64_t sc64[2];
    __shared__ uint32_t sc32[8];

    const int lane = threadIdx.x;
    DIAG_NOW(t_w0);
    const uint32_t slice_id = p.order[blockIdx.x];
    const SliceDesc sd = p.slices[slice_id];
    const uint32_t q = sd.q;
    const uint32_t t0 = p.q_off[q], nd = p.q_off[q + 1] - t0;
    const int L = (int)(2 * nd);                       // <= WL (host)

    WaveCtx C;
    C.sk = sk;
    C.ghdr = ghdr;
    C.gdesc = gdesc;
    C.S.s_rec = reinterpret_cast<double2*>(pend_raw);
    C.S.pend = reinterpret_cast<uint4*>(pend_raw);
    C.S.l_w = l_w;
    C.S.l_mult = l_mult;
    C.S.l_field = l_field;
    C.S.ht_key = ht_key;
    C.S.ht_rec = ht_rec;
    C.S.overflow = &sc32[1];
    C.Q.qmag = p.qmag[q];
    C.Q.probs = (p.probs && p.prior) ? p.probs + (size_t)q * p.k_topics : nullptr;
    C.Q.sqd_ub = C.Q.probs ? p.sqd_ub[q] : 0.0;
    C.Q.sqd_ub_f = C.Q.probs ? __double2float_ru(C.Q.sqd_ub) : 0.0f;
    C.Q.qmag_f = (float)C.Q.qmag;
    C.r_ub = C.Q.probs ? __double2float_ru(33.0 * C.Q.sqd_ub * (1.0 + 0x1p-12)) : 0.0f;

    for (int i = lane; i < WSK + 64; i += 64) sk[i] = 0u;
    for (int i = lane; i < WHT; i += 64) { ht_key[i] = EMPTY; ht_rec[i] = EMPTY; }
    if (lane == 0) { sc32[0] = 0; sc32[1] = 0; sc64[0] = 0ull; *reinterpret_cast<float*>(&sc32[2]) = -INFINITY; }

    // ---- the lists (lane l: list l = field l&1 of distinct term l>>1), resolved by k_wave_prep ----
    WList w{};
    uint32_t p0_lo = 0;
    float coef_raw = 0.f, floor_l = 0.f;
    if (lane < L) {
        const WPrep r = prep[(size_t)blockIdx.x * WL + lane];
        const int field = lane & 1;                    // 0 = body, 1 = title
        const uint64_t tb = (uint64_t)(field ? p.t_rec : p.b_rec);
        w.tb_lo = (uint32_t)tb;
        w.tb_hi = (uint32_t)(tb >> 32);
        w.active = (r.lanes >> 16) != 0;
        w.g0 = r.g0;
        w.g1 = r.g1;
        w.lo_lane = r.lanes & 0xFFu;
        w.hi_lane = (r.lanes >> 8) & 0xFFu;
        w.cg = r.cg;
        w.ge = r.ge;
        p0_lo = r.p0_lo;
        // first float32 weight of the list: the table's weights + the list's first posting (64-bit; the prep record keeps the low word)
        {
            const uint64_t* ptr = field ? p.t_ptr : p.b_ptr;
            const uint32_t term = p.dterm[t0 + (lane >> 1)];
            l_w[lane] = (uint64_t)((field ? p.t_w : p.b_w) + ptr[term]);
        }
        l_mult[lane] = r.mult;
        l_field[lane] = (uint32_t)field;
        // filter coefficient and threshold floor exactly as in k_score_slices (get_metadata.go:57-58,69)
        const double share = (field ? 38.0 : 29.0) * (double)r.mult / C.Q.qmag;
        coef_raw = __double2float_ru(share * (1.0 + 0x1p-12));
        if (r.kth > 0.0f) floor_l = fmaxf(0.0f, __double2float_rd(share * (1.0 - 0x1p-12) * (double)r.kth));
    }
    const float coef_max = __uint_as_float(wave_max((coef_raw > 0.0f && coef_raw < INFINITY) ? __float_as_uint(coef_raw) : 0u));
    C.fx_scale = coef_max > 0.0f ? (float)FX_ONE / coef_max : 1.0f;
    w.coef = coef_raw * C.fx_scale * (1.0f + 0x1p-20f);
    const float thr0_f = __uint_as_float(wave_max(__float_as_uint(floor_l)));
    const uint64_t thr0_key = thr0_f > 0.0f ? fkey((double)thr0_f) : 0ull;
    C.tk = TopK{cd_key, cd_doc, &sc32[0], &sc64[0], reinterpret_cast<float*>(&sc32[2]), thr0_key, thr0_f > 0.0f ? thr0_f : -INFINITY, (uint32_t)WCB};
    if (lane == 0 && thr0_f > 0.0f) { sc64[0] = thr0_key; *reinterpret_cast<float*>(&sc32[2]) = thr0_f; }
    const unsigned long long act_mask = __ballot(lane < L && w.active);
    lds_wait();
    const uint64_t dummy = (uint64_t)p.q_off;          // 8 readable bytes for block slots without a block

    DIAG_NOW(t_w1);
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
        const RoundPlan rp = plan_round(w, p, se, ghdr, gdesc, l_adv, L, act_mask, F, sd.dhi, lane);
        set_round_base(w, p0_lo);
        const uint32_t n_rows = rp.n_rows;
        DIAG_NOW(t_p1);
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
            group_issue<0>(C, w, r0, lane, dummy, rec, dvr);
            group_issue<1>(C, w, r0 + 1, lane, dummy, rec, dvr);
#if SSW_DEPTH == 3
            group_issue<2>(C, w, r0 + 2, lane, dummy, rec, dvr);
#endif
#ifdef SS_DIAG
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
                    _Pragma("unroll") for (int c = 0; c < WCW; c++) u[c] = sk[h[c]];                                       \
                    uint32_t tot = 0;                                                                                      \
                    _Pragma("unroll") for (int c = 0; c < WCW; c++) tot += (uint32_t)__popcll(__ballot(u[c] >= thr_fx));   \
                    WSTAMP(ts2);                                                                                           \
                    WACC(0, ts0, ts1); WACC(1, ts1, ts2);                                                                  \
                    if (tot) {                                                                                             \
                        if (pend_n + tot > (uint32_t)WPW) { ev = EV_OVERFLOW; ev_row = r_; goto ssw_event; }               \
                        _Pragma("unroll") for (int c = 0; c < WCW; c++) {                                                  \
                            const bool surv = u[c] >= thr_fx;                                                              \
                            const unsigned long long m = __ballot(surv);                                                   \
                            if (m) {                                                                                       \
                                const uint32_t d = rl(dvr[S_], c);                                                         \
                                const int l = (int)(d & 15u);                                                              \
                                const uint32_t idx0 = rl(w.pos0, l) + (d >> 6) * 64u;                                      \
                                const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));                \
                                if (surv) C.S.pend[pend_n + (uint32_t)__popcll(m & below)] = make_uint4(rec[S_][c].x, idx0 + (uint32_t)lane, (uint32_t)l, 0u); \
                                pend_n += (uint32_t)__popcll(m);                                                           \
                            }                                                                                              \
                        }                                                                                                  \
                    }                                                                                                      \
                    _Pragma("unroll") for (int c = 0; c < WCW; c++) if (h[c] != (uint32_t)WSK) sk[h[c]] = 0u;              \
                    WSTAMP(ts3);                                                                                           \
                    WACC(2, ts2, ts3);                                                                                     \
                } else {                                                                                                   \
                    if (mode == M_SLOW) { ev = EV_SLOW; ev_row = r_; goto ssw_event; }                                     \
                    /* a row without work (pad row, tail of an oversize window): its dummy loads still count */           \
                    ring_wait<(WDEPTH - 1) * WCW>(rec[S_][WCW - 1]);                                                       \
                }                                                                                                          \
                WSTAMP(ts4);                                                                                               \
                group_issue<S_>(C, w, r_ + WDEPTH, lane, dummy, rec, dvr);                                                 \
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
                const uint4 hv = *reinterpret_cast<const uint4*>(ghdr[ev_row]);
                const uint32_t b_lo = rfl(hv.x), span = rfl(hv.y), n_blk = rfl(hv.w);
                if (pend_n) wave_flush(C.S, C.tk, C.Q, p, lane, pend_n);
                pend_n = 0;
                r0 = ev_row;                            // EV_FLUSH: the row has not been touched: it runs again
                if (ev == EV_OVERFLOW) { slow_window(C, p, w, lane, ev_row, n_blk, b_lo, span, false); r0 = ev_row + 1; }
                if (ev == EV_SLOW) { slow_window(C, p, w, lane, ev_row, n_blk, b_lo, span, true); r0 = ev_row + 1; }
                thr_fx = wave_thr_fx(C);
                DIAG_NOW(t_e1);
                WDIAG_ADD(12, t_e1 - t_e0);
            }
        }
        DIAG_NOW(t_p2);
        WDIAG_ADD(13, t_p2 - t_p1);
        // ---- next round starts at e ----
        if (lane < L && w.active) w.cg += l_adv[lane];
        F = rp.e;
        lds_wait();
    }
    WDIAG_ADD(7, pend_n);
    if (pend_n) wave_flush(C.S, C.tk, C.Q, p, lane, pend_n);
    DIAG_NOW(t_w2);

    // hand the candidates in: appended to the query's list (k_merge_flat sorts; a slice sorts only if it holds more than k)
    if (sc32[0] > (uint32_t)p.k) topk_compact(C.tk, p.k);
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
void launch_score_wave(const void* params, unsigned n_slices, void* prep, hipStream_t st) {
    const ScoreParams& p = *reinterpret_cast<const ScoreParams*>(params);
    hipLaunchKernelGGL(ssw::k_wave_prep, dim3((n_slices * WL + 255) / 256), dim3(256), 0, st, p, n_slices, reinterpret_cast<WPrep*>(prep));
    hipLaunchKernelGGL(ssw::k_score_wave, dim3(n_slices), dim3(64), 0, st, p, reinterpret_cast<const WPrep*>(prep));
}
int score_wave_max_lists() { return WL; }
int score_wave_max_k() { return WCB / 2; }
void score_wave_diag_dump() {
#ifdef SS_DIAG
    unsigned long long h[32];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wdiag), sizeof(h)) == hipSuccess) {
        const char* names[20] = {"slices", "rounds", "rows", "-", "ev_flush", "ev_overflow", "ev_slow", "flushed_records", "handed_in", "blocks",
                                 "cyc_setup", "cyc_plan", "cyc_events", "cyc_stream", "cyc_epilogue", "cyc_total", "cyc_row_add", "cyc_row_read",
                                 "cyc_row_append_clear", "cyc_row_issue"};
        fprintf(stderr, "[ss diag] k_score_wave (lane 0 of every slice):");
        for (int i = 0; i < 20; i++) fprintf(stderr, " %s=%llu", names[i], h[i]);
        fprintf(stderr, "\n");
    }
#endif
}
}  // namespace ss
