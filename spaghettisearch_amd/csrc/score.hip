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
// Device design (HBM-bound streaming of 8-byte posting records — exactly SURVEY.md §8d's 8 bytes per posting):
//   * scoring layout: one 8-byte record per posting {doc u32, impact f32}, impact = float32 UPPER bound of
//     w / magnitude(doc, field).  The float32 weights and float64 magnitudes stay in the index's own arrays and
//     are only read for documents that can still enter the top-k.
//   * the host plans (it keeps df per term): duplicates -> multiplicities, unknown terms dropped, each query's
//     doc range cut into slices, longest first; one 512-thread workgroup per (query, slice).
//   * k_score_slices builds the slice's window plan in LDS (window = a doc range holding <= CAP records; every
//     record of a doc lies in one window), then streams the windows, window j+1's records in flight while window
//     j is processed.  Per window TWO stages:
//       1. FILTER (every record; one LDS integer atomic, one LDS read, one barrier per window, no probing, no keys):
//          c = impact * coef(list), in fixed point and rounded up, is added into slot hash(doc) of a small table
//          ("sketch": collisions only make the sum larger); after the barrier every record reads its slot back: the
//          value is an upper bound of its document's FinalRank.  If it is below the running threshold the record is
//          dropped.  Three tables rotate so that clearing the slots a window touched needs no second barrier.
//       2. EXACT (survivors only, batched): surviving records are appended to a pending list {doc, index in
//          list, list}; when the list fills (or the slice ends) all threads gather the survivors' float32
//          weights and float64 magnitudes from the index (one round of HBM latency for the whole batch), aggregate
//          them per document in an LDS hash table (float32 addends in float64: exact, order-free), compute the
//          reference's float64 arithmetic literally, and feed the running top-k (exact keys, bitonic compaction).
//     The running threshold starts at a bound known before any posting is read: the index keeps, per term and
//     field, the k'-th largest impact for k' = 1,2,4..1024, and k' >= k postings of ONE list are k' distinct
//     documents whose FinalRank is at least their own contribution.
//     Inputs outside the filter's assumptions (negative or non-finite weights, priors or probabilities, zero
//     magnitudes under non-zero weights, queryLength <= 0) switch the filter off for the call: every record
//     survives and stage 2 alone decides — same results, no separate code path.
//   * k_merge_topk: one workgroup per query merges its slices' top-k lists, re-derives
//     title/body/pagerank of the k winners by binary search and writes ss_hit rows.
//   Ties: ascending doc id (Q10); NaN finals last.
#include <chrono>
#include "score_common.hpp"

namespace {

// ---- K4: score one (query, doc-range slice) --------------------------------------
#ifndef SS_WGS_PER_CU
#define SS_WGS_PER_CU 2
#endif
struct ScoreLds {                     // byte offsets into the dynamic LDS block
    size_t s_rec, l_rec, l_w, sc64, cd_key, cd_doc, ht_key, ht_rec, sk, tbl, l_mult, l_field, f_cur, f_nxt, l_coef, sc32, sel, sel64, off, total;
};
__host__ __device__ inline ScoreLds score_lds_layout(int cb) {
    ScoreLds o{};
    size_t p = 0;
    o.s_rec = p;   p += (size_t)PC * 16;            // pending survivors {doc, index, list, -}; rewritten in place as {addend f64, magnitude f64}
    o.l_rec = p;   p += (size_t)MAXL * 8;
    o.l_w = p;     p += (size_t)MAXL * 8;
    o.sc64 = p;    p += 4 * 8;
    o.cd_key = p;  p += (size_t)cb * 8;
    o.cd_doc = p;  p += (size_t)cb * 4;
    o.ht_key = p;  p += (size_t)HT * 4;
    o.ht_rec = p;  p += (size_t)HT * 4;
    o.sk = p;      p += (size_t)3 * SK * 4;
    o.tbl = p;     p += (size_t)TBL_CAP * 4;
    o.l_mult = p;  p += (size_t)MAXL * 4;
    o.l_field = p; p += (size_t)MAXL * 4;
    o.f_cur = p;   p += (size_t)MAXL * 4;
    o.f_nxt = p;   p += (size_t)MAXL * 4;
    o.l_coef = p;  p += (size_t)MAXL * 4;
    o.sc32 = p;    p += 16 * 4;
    o.sel = p;     p += (256 + 8) * 4;              // topk_select_inl: histogram + scalars
    o.sel64 = p;   p += 4 * 8;
    o.off = p;     p += (size_t)OFF_CAP * 2;
    o.total = (p + 15) & ~(size_t)15;
    return o;
}

// The workgroup's LDS arrays as typed pointers (passed by reference into force-inlined helpers: after inlining they
// are plain SSA values derived from the LDS block, so every access stays a ds_* instruction).
struct SliceLds {
    double2* s_rec;      // [PC] exact stage: {addend (summed: BodyRank/TitleRank of the doc), magnitude}
    uint4* pend;         // [PC] the same bytes while pending: {doc, index in list, list, -}
    uint64_t* l_rec;     // [MAXL] address of the list's first record
    uint64_t* l_w;       // [MAXL] address of the list's first float32 weight
    uint32_t* ht_key;    // [HT]
    uint32_t* ht_rec;    // [HT] first body record (low 16 bits) and first title record (high 16) of the slot's doc
    uint32_t* sk;        // [3][SK] filter tables (fixed point)
    uint32_t* tbl;       // [TBL_CAP] cursor of list l at the start of window j: tbl[j*L+l]
    uint32_t* l_mult;    // [MAXL] multiplicity of the term in the query
    uint32_t* l_field;   // [MAXL] 0 = body, 1 = title
    uint32_t* f_cur;     // [MAXL] oversize fallback: sub-window start
    uint32_t* f_nxt;     // [MAXL] oversize fallback: sub-window end
    float* l_coef;       // [MAXL] filter coefficient in fixed-point units: upper bound of (38|29)*mult/sqrt(queryLength) * scale
    uint16_t* off;       // [OFF_CAP] offset of list l inside window j: off[j*OS+l], count at [j*OS+OS-1]
    uint32_t* overflow;  // shared scalar
    uint32_t* sel;       // [256 + 8] scratch of topk_select_inl
    uint64_t* sel64;     // [4]
};

// generic window loop (L > LCH lists): records of one window as raw 8-byte vectors (one global_load_dwordx2 per posting)
// with their list, index in the list and filter coefficient; the list of record i by binary search in the window's offsets
__device__ __forceinline__ void load_window(const SliceLds& S, int j, int L, int OS, int tid,
                                            u32x2 (&rec)[PPT], uint32_t (&rl)[PPT], uint32_t (&ri)[PPT], float (&rc)[PPT]) {
    const uint32_t* tbl_j = S.tbl + j * L;
    const uint16_t* off_j = S.off + j * OS;
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
            ri[r] = tbl_j[lo] + (i - off_j[lo]);
            rec[r] = load_rec(S.l_rec[lo], ri[r]);
            rl[r] = lo;
            rc[r] = S.l_coef[lo];
        }
    }
}

// ---- exact stage -----------------------------------------------------------------------------------------------
// score ONE doc of the exact stage's table (get_metadata.go:31-69) — the thread that claimed the doc's slot owns it —
// reset its slot and offer it to the running top-k.  All threads call (slot = EMPTY: nothing to score): the admit
// loop's barrier is workgroup-wide.
__device__ __forceinline__ void score_owned(const SliceLds& S, const TopK& tk, const SliceQuery& Q, const ScoreParams& p, int tid,
                                            uint32_t doc, uint32_t slot) {
    uint32_t e_doc = EMPTY;
    uint64_t e_key = 0;
    if (slot != EMPTY) {
        e_doc = doc;
        const uint32_t fr = S.ht_rec[slot];
        // no posting of a field => its sum is 0 and 0/(m*q) is 0 (or NaN -> 0) for every m
        double2 rb = make_double2(0.0, 1.0), rt = make_double2(0.0, 1.0);
        const uint32_t ib = fr & 0xFFFFu, it = fr >> 16;
        if (ib != NOREC) rb = S.s_rec[ib];
        if (it != NOREC) rt = S.s_rec[it];
        S.ht_key[slot] = EMPTY;
        S.ht_rec[slot] = EMPTY;
        const double B = rb.x, T = rt.x, mb = rb.y, mt = rt.y;
        const uint64_t thr0 = *tk.thr;
        const float thr_f = *tk.thr_f;
        // cheap float estimate first; skipped only if the estimate, with a 1e-4 relative margin, is clearly below the
        // threshold; anything non-finite falls through to the exact path.
        // v_rcp_f32 (1 ulp) instead of an IEEE division: the 1e-4 margin below covers it; 0*inf = NaN falls through
        const float ea = 38.0f * ((float)T * __builtin_amdgcn_rcpf((float)mt * Q.qmag_f)), eb = 29.0f * ((float)B * __builtin_amdgcn_rcpf((float)mb * Q.qmag_f)),
                    ec = 33.0f * Q.sqd_ub_f;
        if ((ea + eb + ec) + (fabsf(ea) + fabsf(eb) + fabsf(ec)) * 1e-4f + 1e-30f < thr_f) {
            e_doc = EMPTY;
        } else {
            double title, body, fin;
            if (Q.probs) {
                // the prior row (128 B) is only fetched if the doc can still make the top-k:
                // every operation of final_rank is monotone in sqd, so sqd_ub bounds the score
                final_rank(T, B, mt, mb, Q.qmag, Q.sqd_ub, title, body, fin);
                if (fkey(fin) >= thr0 || fin != fin) final_rank(T, B, mt, mb, Q.qmag, topic_dot(p.prior, Q.probs, p.k_topics, e_doc), title, body, fin);
                else e_doc = EMPTY;
            } else {
                final_rank(T, B, mt, mb, Q.qmag, 0.0, title, body, fin);
            }
            e_key = fkey(fin);
        }
    }
    // threshold filter into the candidate buffer; overflow -> compact and retry
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
        lds_barrier();
        if (!*S.overflow) break;
        topk_cut(tk, p.k, S.sel, S.sel64);   // raises thr; count back to <= k (unordered: radix selection where one entry per thread fits)
        if (tid == 0) *S.overflow = 0;
        lds_barrier();
    }
}

// The pending survivors pend[0, n) -> per-document sums -> FinalRank -> running top-k.  All threads call with the
// same n (<= PC).  On return the pending list is empty and the hash table clean.
__device__ __forceinline__ void flush_pending(const SliceLds& S, const TopK& tk, const SliceQuery& Q, const ScoreParams& p, int tid, uint32_t n) {
    DIAG_ADD(2, 1);
    DIAG_ADD(3, n);
    DIAG_NOW(t_f0);
    uint32_t pdoc[PPX], pl[PPX], own[PPX];
    float pw[PPX];
    double pm[PPX];
#pragma unroll
    for (int r = 0; r < PPX; r++) {
        const uint32_t i = tid + r * TPB;
        pl[r] = EMPTY;
        if (i < n) {
            const uint4 e = S.pend[i];
            pdoc[r] = e.x;
            pl[r] = e.z;
            // one round of memory latency for the whole batch: the survivor's float32 weight and its doc's float64 magnitude
            pw[r] = load_w(S.l_w[e.z], e.y);
            pm[r] = (S.l_field[e.z] ? p.t_mag : p.b_mag)[e.x];
        }
    }
    lds_barrier();                                  // every pending entry has been read: the bytes may be rewritten
    // accumulate per doc (main_retrieve.go:61-69,170-187); float32 addends in float64: exact, any order.
    // A table slot names, per field, the FIRST record of its doc (one 32-bit CAS on the packed word); the rare later
    // records of the same (doc, field) add their addend into the first one's.
#pragma unroll
    for (int r = 0; r < PPX; r++) {
        const uint32_t l = pl[r];
        own[r] = EMPTY;
        if (l != EMPTY) {
            const uint32_t i = tid + r * TPB;
            uint32_t h = (((pdoc[r] * 2654435761u) >> 20) * (uint32_t)(HT / 256)) >> 4;      // 12 hash bits -> [0, HT)
            for (;;) {
                const uint32_t prev = atomicCAS(&S.ht_key[h], EMPTY, pdoc[r]);
                if (prev == EMPTY) own[r] = h;     // this thread claimed the doc's slot: it will score the doc
                if (prev == EMPTY || prev == pdoc[r]) break;
                h = h + 1 == HT ? 0 : h + 1;
            }
            const uint32_t field = S.l_field[l];        // 0 = body (low half), 1 = title (high half)
            const double v = (double)pw[r] * (double)S.l_mult[l];
            S.s_rec[i] = make_double2(v, pm[r]);
            asm volatile("" ::: "memory");         // program order: parked before it can be named (LDS runs a wave's operations in order)
            uint32_t old = S.ht_rec[h], first;
            for (;;) {
                first = field ? old >> 16 : old & 0xFFFFu;
                if (first != NOREC) break;
                const uint32_t want = field ? (old & 0xFFFFu) | (i << 16) : (old & 0xFFFF0000u) | i;
                const uint32_t prev = atomicCAS(&S.ht_rec[h], old, want);
                if (prev == old) break;            // first stays NOREC: this record is the first of its (doc, field)
                old = prev;
            }
            if (first != NOREC) atomicAdd(&S.s_rec[first].x, v);
        }
    }
    lds_barrier();
#pragma unroll
    for (int r = 0; r < PPX; r++) score_owned(S, tk, Q, p, tid, pdoc[r], own[r]);
    DIAG_NOW(t_f1);
    DIAG_ADD(8, t_f1 - t_f0);
}

// ---- oversize window (a list is locally much denser than planned): bisect its doc range until a piece fits and hand
//      every record of the piece to the exact stage.  Rare; no filter.  The pending list must be empty on entry. ----
__device__ __forceinline__ void oversize_window(const SliceLds& S, const TopK& tk, const SliceQuery& Q, const ScoreParams& p, int tid, int L,
                                                int j, int n_win, const SliceDesc& sd, uint32_t drv, uint64_t drv_addr) {
    if (tid < L) S.f_cur[tid] = S.tbl[j * L + tid];
    uint32_t flo = j == 0 ? sd.dlo : load_doc(drv_addr, S.tbl[j * L + drv]);
    const uint32_t fend = j + 1 == n_win ? sd.dhi : load_doc(drv_addr, S.tbl[(j + 1) * L + drv]);
    __syncthreads();
    for (;;) {
        uint32_t fhi = fend;
        uint32_t cnt;
        for (;;) {
            if (tid < L) {
                S.f_nxt[tid] = fhi == fend ? S.tbl[(j + 1) * L + tid]
                                           : lower_bound_addr(S.l_rec[tid], S.f_cur[tid], S.tbl[(j + 1) * L + tid], fhi);
            }
            __syncthreads();
            cnt = 0;
            for (int l = 0; l < L; l++) cnt += S.f_nxt[l] - S.f_cur[l];
            if (cnt <= (uint32_t)CAP) break;
            // a single doc has at most L <= 132 postings, so the bisection ends
            fhi = flo + (uint32_t)(((uint64_t)fhi - flo) >> 1);
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < PPT; r++) {
            const uint32_t i = tid + r * TPB;
            if (i < cnt) {
                uint32_t run = 0;
                int l = 0;
                for (; l < L; l++) {
                    const uint32_t len = S.f_nxt[l] - S.f_cur[l];
                    if (i < run + len) break;
                    run += len;
                }
                const uint32_t idx = S.f_cur[l] + (i - run);
                S.pend[i] = make_uint4(load_doc(S.l_rec[l], idx), idx, (uint32_t)l, 0u);
            }
        }
        lds_barrier();
        flush_pending(S, tk, Q, p, tid, cnt);
        bool done = true;
        for (int l = 0; l < L; l++) done = done && S.f_nxt[l] == S.tbl[(j + 1) * L + l];
        __syncthreads();
        if (done) break;
        if (tid < L) S.f_cur[tid] = S.f_nxt[tid];
        flo = fhi;
        __syncthreads();
    }
}

// ---- chunked windows (L <= LCH): a wave loads 64 consecutive records of ONE list per chunk ---------------------------
// Everything that depends on the list — record address, filter coefficient, how many lanes hold a record — is then
// wave-uniform and lives in scalar registers; a lane spends ~10 vector instructions per record.  A window is at most
// MAXCH chunks (sum over the lists of ceil(len/64)); its row in S.off holds the cumulative chunk counts (one byte per list).
struct ChunkRef { uint32_t l, start, cnt; };     // list, index of the chunk's first record in the list, records (0: no chunk)
struct WaveLists { uint32_t rec_lo, rec_hi; float coef; };   // lane l < L keeps list l's record address and filter coefficient
struct WinRow { uint32_t cum, t0, t1, n_ch; };   // lane l <= L: cumulative chunk count before list l, the list's cursors at both window ends

// one row of the window plan, read lane-parallel (lanes past L repeat lane L: no divergence)
__device__ __forceinline__ WinRow win_row(const SliceLds& S, int L, int j, int n_win, int lane) {
    const int li = min(lane, L);
    const int jr = min(j, n_win - 1);      // windows past the end: read the last row (stays in bounds), report no chunks
    const uint8_t* row = reinterpret_cast<const uint8_t*>(S.off) + jr * OSC;
    WinRow w;
    w.cum = row[li];
    w.t0 = S.tbl[jr * L + li];           // lane L reads one entry past the row's lists: the next row's first cursor (unused)
    w.t1 = S.tbl[(jr + 1) * L + min(li, L - 1)];
    w.n_ch = j < n_win ? (uint32_t)__builtin_amdgcn_readlane((int)w.cum, L) : 0u;
    return w;
}

__device__ __forceinline__ ChunkRef chunk_of(const WinRow& w, int L, uint32_t c, int lane) {
    ChunkRef cr{0u, 0u, 0u};
    if (c < w.n_ch && w.n_ch <= (uint32_t)MAXCH) {
        // list = the last one whose cumulative count is <= c (cum_0 = 0; lists without chunks repeat their predecessor's count)
        const unsigned long long m = __ballot(lane < L && w.cum <= c);
        const int l = __builtin_amdgcn_readfirstlane(__popcll(m) - 1);
        const uint32_t rem = c - (uint32_t)__builtin_amdgcn_readlane((int)w.cum, l);
        cr.l = (uint32_t)l;
        cr.start = (uint32_t)__builtin_amdgcn_readlane((int)w.t0, l) + rem * 64u;
        cr.cnt = min(64u, (uint32_t)__builtin_amdgcn_readlane((int)w.t1, l) - cr.start);
    }
    return cr;
}

// Requests window j's records into ring slot S_.  ALWAYS CPW loads per wave — lanes past the chunk's end repeat its last
// record, chunk slots without a chunk (and windows past the end) read a harmless dummy word — so that the number of loads
// in flight behind any slot is a compile-time constant and the wait in front of chunk_add is a counted s_waitcnt vmcnt(N),
// not a drain.  Address = wave-uniform list base + 32-bit lane offset (global_load saddr form: no 64-bit vector math).
template <int S_>
__device__ __forceinline__ void chunk_issue(const WinRow& w, const WaveLists& wl, int L, int wave, int lane, uint64_t dummy,
                                            u32x2 (&rec)[DEPTH][CPW], uint32_t (&cnt)[DEPTH][CPW], float (&cf)[DEPTH][CPW]) {
#pragma unroll
    for (int r = 0; r < CPW; r++) {
        const ChunkRef cr = chunk_of(w, L, (uint32_t)(wave + r * WAVES), lane);
        cnt[S_][r] = cr.cnt;
        uint64_t base = dummy;
        uint32_t voff = 0;
        cf[S_][r] = 0.f;
        if (cr.cnt) {
            base = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)wl.rec_hi, (int)cr.l) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane((int)wl.rec_lo, (int)cr.l);
            cf[S_][r] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(wl.coef), (int)cr.l));
            voff = (cr.start + min((uint32_t)lane, cr.cnt - 1u)) * (uint32_t)sizeof(Rec);
        }
        rec[S_][r] = *(gptr_u2)(base + voff);
    }
}


// filter slot of a doc: doc ids inside a window are a narrow range, so folding the id's low bits spreads them
__device__ __forceinline__ uint32_t fx_slot(uint32_t doc) { return (doc ^ (doc >> SS_SK_BITS)) & (uint32_t)(SK - 1); }

template <int S_>
__device__ __forceinline__ void chunk_add(uint32_t* sk_cur, int lane, const u32x2 (&rec)[DEPTH][CPW], const uint32_t (&cnt)[DEPTH][CPW],
                                          const float (&cf)[DEPTH][CPW], uint32_t (&h)[CPW]) {
#pragma unroll
    for (int r = 0; r < CPW; r++) {
        h[r] = EMPTY;
        if (cnt[S_][r]) {                                        // wave-uniform
            if ((uint32_t)lane < cnt[S_][r]) {
                h[r] = fx_slot(rec[S_][r].x);
                atomicAdd(&sk_cur[h[r]], fx_share(__uint_as_float(rec[S_][r].y), cf[S_][r]));
            }
        }
    }
}

template <int S_>
__device__ __forceinline__ void chunk_filter(const SliceLds& S, const uint32_t (&u)[CPW], uint32_t* sk_prv, uint32_t* surv_cnt_cur, int L, int j, int n_win, int wave, int lane,
                                             uint32_t thr_fx, uint32_t pbase, const u32x2 (&rec)[DEPTH][CPW],
                                             const uint32_t (&h)[CPW], uint32_t (&hprev)[CPW]) {
#pragma unroll
    for (int r = 0; r < CPW; r++) {
        const bool surv = h[r] != EMPTY && u[r] >= thr_fx;
        if (hprev[r] != EMPTY) sk_prv[hprev[r]] = 0u;
        hprev[r] = h[r];
        const unsigned long long m = __ballot(surv);
        if (m) {
            // rare: which list and where — looked up again instead of being carried through the pipeline
            const WinRow w = win_row(S, L, j, n_win, lane);
            const ChunkRef cr = chunk_of(w, L, (uint32_t)(wave + r * WAVES), lane);
            const int leader = __ffsll((long long)m) - 1;
            uint32_t b = 0;
            if (lane == leader) b = atomicAdd(surv_cnt_cur, (uint32_t)__popcll(m));
            b = (uint32_t)__shfl((int)b, leader, 64);
            if (surv) {
                const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
                S.pend[pbase + b + (uint32_t)__popcll(m & below)] = make_uint4(rec[S_][r].x, cr.start + (uint32_t)lane, cr.l, 0u);
            }
        }
    }
}

// NT = threads of the calling workgroup.  FUSED: called by the last slice of the query to finish inside k_score_slices — the
// slices' lists were stored write-through and are read with sc1 loads (no fence; see the hand-off at the end of
// k_score_slices).
// FLAT: the query's slices (k_score_wave) appended their candidates to ONE list per query (qc_cnt[q] entries from
// slice_base[q] * k on); otherwise every slice owns k entries and so_cnt[s] says how many it filled.
constexpr int MERGE_T = 16;     // >= the wave kernel's lists per query (score_wave_max_lists)
size_t merge_lds_bytes(int k, int cb);
template <int NT, bool FUSED, bool FLAT = false>
__device__ __forceinline__ void merge_query(const ScoreParams& p, const uint32_t q, unsigned char* smem, const int cbm) {
    double* accT = reinterpret_cast<double*>(smem);                    // [k]
    double* accB = accT + p.k;                                         // [k]
    double* mgT = accB + p.k;                                          // [k]
    double* mgB = mgT + p.k;                                           // [k]
    uint64_t* cd_key = reinterpret_cast<uint64_t*>(mgB + p.k);         // [cb]
    uint64_t* sc64 = cd_key + cbm;                                     // [1]
    uint32_t* cd_doc = reinterpret_cast<uint32_t*>(sc64 + 1);          // [cb]
    uint32_t* sc32 = cd_doc + cbm;                                     // [4]
    uint64_t* tm_p0 = reinterpret_cast<uint64_t*>(sc32 + 4);           // [MERGE_T] FLAT: the terms' combined lists and multiplicities,
    uint64_t* tm_p1 = tm_p0 + MERGE_T;                                 //           fetched while the candidates are on their way
    uint32_t* tm_mult = reinterpret_cast<uint32_t*>(tm_p1 + MERGE_T);
    TopK tk{cd_key, cd_doc, &sc32[0], &sc64[0], reinterpret_cast<float*>(&sc32[2]), 0ull, -INFINITY, (uint32_t)cbm};
    const int tid = threadIdx.x;
    const int k = p.k;
    if (tid == 0) { sc32[0] = 0; sc32[1] = 0; sc64[0] = 0ull; }
    DIAG_NOWX(t_g0);
    __syncthreads();
    if (FLAT) {
#ifdef SSM_EXP_NOGATHER        // (timing experiments only: wrong results)
        const uint32_t n = min(p.qc_cnt[q], 1u);
#else
        const uint32_t n = p.qc_cnt[q];
#endif
        DIAG_ADD(18, n);
        DIAG_ADD(17, 1);
        const size_t base = (size_t)p.slice_base[q] * k;
        uint32_t* overflow = &sc32[1];
        // two candidates per thread and pass, both loads issued before either is used (no branch around them: a lane without a
        // candidate re-reads entry 0): ~300 candidates per query at config 3 are ONE round of memory latency, not two
        uint64_t key[2];
        uint32_t doc[2];
        bool have[2];
        auto load_pass = [&](uint32_t i0) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const uint32_t i = i0 + (uint32_t)r * NT + tid;
                have[r] = i < n;
                const size_t ii = base + (have[r] ? i : 0u);
                key[r] = p.so_key[ii];
                doc[r] = p.so_doc[ii];
            }
        };
        load_pass(0);
        // what the explain stage needs of the query's terms is requested now, behind the first candidates: these loads (two
        // dependent levels) run beside them instead of after the sort
        {
            const uint32_t tq0 = p.q_off[q], tnd = p.q_off[q + 1] - tq0;
            if ((uint32_t)tid < min(tnd, (uint32_t)MERGE_T)) {
                const uint32_t term = p.dterm[tq0 + tid];
                tm_p0[tid] = p.c_ptr[term];
                tm_p1[tid] = p.c_ptr[term + 1];
                tm_mult[tid] = p.dmult[tq0 + tid];
            }
        }
        __syncthreads();
        if (tid == 0) p.qc_cnt[q] = 0u;                                  // every thread has its copy: zero again for the next batch (no memset between batches)
        for (uint32_t i0 = 0; i0 < n;) {
            for (;;) {
                const uint64_t thr = *tk.thr;
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    if (have[r]) {
                        if (key[r] >= thr) {
                            const uint32_t j = atomicAdd(tk.count, 1u);
                            if (j < tk.cb) { tk.key[j] = key[r]; tk.doc[j] = doc[r]; have[r] = false; }
                            else *overflow = 1;
                        } else {
                            have[r] = false;
                        }
                    }
                }
                __syncthreads();
                if (!*overflow) break;
                topk_compact_net(tk, k);
                if (tid == 0) *overflow = 0;
                __syncthreads();
            }
            i0 += 2 * NT;
            if (i0 < n) load_pass(i0);
        }
    } else
    for (uint32_t s = p.slice_base[q]; s < p.slice_base[q + 1]; s++) {
        const uint32_t n = FUSED ? __hip_atomic_load(&p.so_cnt[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p.so_cnt[s];   // <= k, and cb >= 2k: room after a compaction
        if (sc32[0] + n > tk.cb) topk_compact(tk, k);
        __syncthreads();
        const uint64_t thr = *tk.thr;
        for (uint32_t i = tid; i < n; i += NT) {
            const uint64_t key = FUSED ? __hip_atomic_load(&p.so_key[(size_t)s * k + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p.so_key[(size_t)s * k + i];
            if (key >= thr) {
                const uint32_t j = atomicAdd(tk.count, 1u);
                tk.key[j] = key;
                tk.doc[j] = FUSED ? __hip_atomic_load(&p.so_doc[(size_t)s * k + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p.so_doc[(size_t)s * k + i];
            }
        }
        __syncthreads();
    }
#ifdef SSM_EXP_NOSORT           // (timing experiments only: wrong results)
    __syncthreads();
    if (tid == 0) sc32[0] = min(sc32[0], (uint32_t)k);
    __syncthreads();
#else
    if (FLAT) topk_compact_net(tk, k);
    else topk_compact(tk, k);
#endif
    const uint32_t n_out = sc32[0];

    // explain: TitleRank/BodyRank of the winners, re-derived from the posting lists
    DIAG_NOWX(t_m0);
    DIAG_ADD(15, t_m0 - t_g0);
    for (uint32_t i = tid; i < n_out; i += NT) { accT[i] = 0.0; accB[i] = 0.0; mgT[i] = 1.0; mgB[i] = 1.0; }
    __syncthreads();
    const uint32_t t0 = p.q_off[q], nd = p.q_off[q + 1] - t0;
    const bool has_phrase = p.ph_off && p.ph_off[q + 1] > p.ph_off[q];
    const uint32_t L = 2 * nd + (has_phrase ? 4u : 0u);
    if (FLAT) {
        // k_score_wave's queries have their combined lists: ONE search per (winner, term) finds the doc's body and title posting
        // side by side, and it runs through the skip index (4 bytes per 64 postings: cache-resident) and then inside one 512-byte
        // block, instead of two interpolation searches over the whole lists (the merge's searches were 58 of its 98 us)
#ifdef SSM_EXP_NOEXPLAIN
        for (uint32_t task = tid; task < 0 * nd; task += NT) {
#else
        for (uint32_t task = tid; task < n_out * nd; task += NT) {
#endif
            const uint32_t i = task / nd, l = task % nd;
            const uint64_t p0 = tm_p0[l], p1 = tm_p1[l];
            if (p1 == p0) continue;
            const uint32_t d = cd_doc[i];
            const uint32_t g0 = (uint32_t)(p0 >> 6), g1 = (uint32_t)((p1 - 1) >> 6);
            // blocks g0+1 .. g1 start inside the list: the doc's first posting lies in the block before the first entry >= d,
            // or — if that entry IS d — at the very start of the next one: search [block start, next block start + 2)
            const uint32_t gb = g0 + (g1 > g0 ? skip_lower_bound<false>(p.c_skip, g0 + 1, g1 + 1, d) : 0u);
            uint64_t lo = max(p0, (uint64_t)gb << 6), hi = min(p1, ((uint64_t)(gb + 1) << 6) + 2);
            const uint64_t list_addr = (uint64_t)p.c_rec;
            // a plain binary search (seven dependent probes): the merge is bound by the NUMBER of scattered loads its 300 searches
            // per query issue, not by their latency — an 8-ary search in two steps (15 probes) made the explain 37 % slower
            while (lo < hi) {                                        // first position with doc >= d
                const uint64_t mid = (lo + hi) >> 1;
                if ((load_doc(list_addr, mid) & 0x7FFFFFFFu) < d) lo = mid + 1; else hi = mid;
            }
            const double mult = (double)tm_mult[l];
            for (uint64_t pos = lo; pos < min(p1, lo + 2); pos++) {  // body first, then title, of the same doc
                const uint32_t rd = load_doc(list_addr, pos);
                if ((rd & 0x7FFFFFFFu) != d) break;
                const int field = (int)(rd >> 31);
                const double v = (double)p.c_w[pos] * mult;
                const double mag = (field ? p.t_mag : p.b_mag)[d];
                if (field) { atomicAdd(&accT[i], v); mgT[i] = mag; }
                else { atomicAdd(&accB[i], v); mgB[i] = mag; }
            }
        }
    } else
    for (uint32_t task = tid; task < n_out * L; task += NT) {
        const uint32_t i = task / L, l = task % L;
        const int field = l & 1;
        uint64_t addr, waddr;
        uint32_t len;
        double mult;
        if (l < 2 * nd) {
            const uint32_t term = p.dterm[t0 + (l >> 1)];
            const uint64_t* ptr = field ? p.t_ptr : p.b_ptr;
            addr = (uint64_t)((field ? p.t_rec : p.b_rec) + ptr[term]);
            waddr = (uint64_t)((field ? p.t_w : p.b_w) + ptr[term]);
            len = (uint32_t)(ptr[term + 1] - ptr[term]);
            mult = (double)p.dmult[t0 + (l >> 1)];
        } else {
            addr = (uint64_t)(x_rec_of(p, (int)(l - 2 * nd)) + p.x_off[q]);
            waddr = (uint64_t)(x_w_of(p, (int)(l - 2 * nd)) + p.x_off[q]);
            len = p.x_cnt[(size_t)q * 4 + (l - 2 * nd)];
            mult = 1.0;
        }
        const uint32_t d = cd_doc[i];
        const uint32_t pos = lower_bound_interp(addr, 0, len, d);
        if (pos < len && load_doc(addr, pos) == d) {
            const double v = (double)load_w(waddr, pos) * mult;
            const double mag = (field ? p.t_mag : p.b_mag)[d];
            if (field) { atomicAdd(&accT[i], v); mgT[i] = mag; }
            else { atomicAdd(&accB[i], v); mgB[i] = mag; }
        }
    }
    __syncthreads();
    DIAG_NOWX(t_m1);
    DIAG_ADD(14, t_m1 - t_m0);
    const double qmag = p.qmag[q];
    const double* probs = (p.probs && p.prior) ? p.probs + (size_t)q * p.k_topics : nullptr;
    for (uint32_t i = tid; i < (uint32_t)k; i += NT) {
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
    DIAG_NOWX(t_m2);
    DIAG_ADD(16, t_m2 - t_m1);
}


__global__ __launch_bounds__(TPB, (TPB / 256) * SS_WGS_PER_CU) void k_score_slices(ScoreParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ScoreLds lo_ = score_lds_layout(p.cb);
    uint64_t* sc64 = reinterpret_cast<uint64_t*>(smem + lo_.sc64);        // [0]: thr
    uint64_t* cd_key = reinterpret_cast<uint64_t*>(smem + lo_.cd_key);    // [cb]
    uint32_t* cd_doc = reinterpret_cast<uint32_t*>(smem + lo_.cd_doc);    // [cb]
    uint32_t* sc32 = reinterpret_cast<uint32_t*>(smem + lo_.sc32);        // [16] scalars
    SliceLds S;
    S.s_rec = reinterpret_cast<double2*>(smem + lo_.s_rec);
    S.pend = reinterpret_cast<uint4*>(smem + lo_.s_rec);
    S.l_rec = reinterpret_cast<uint64_t*>(smem + lo_.l_rec);
    S.l_w = reinterpret_cast<uint64_t*>(smem + lo_.l_w);
    S.ht_key = reinterpret_cast<uint32_t*>(smem + lo_.ht_key);
    S.ht_rec = reinterpret_cast<uint32_t*>(smem + lo_.ht_rec);
    S.sk = reinterpret_cast<uint32_t*>(smem + lo_.sk);
    S.tbl = reinterpret_cast<uint32_t*>(smem + lo_.tbl);
    S.l_mult = reinterpret_cast<uint32_t*>(smem + lo_.l_mult);
    S.l_field = reinterpret_cast<uint32_t*>(smem + lo_.l_field);
    S.f_cur = reinterpret_cast<uint32_t*>(smem + lo_.f_cur);
    S.f_nxt = reinterpret_cast<uint32_t*>(smem + lo_.f_nxt);
    S.l_coef = reinterpret_cast<float*>(smem + lo_.l_coef);
    S.off = reinterpret_cast<uint16_t*>(smem + lo_.off);
    S.overflow = &sc32[1];
    S.sel = reinterpret_cast<uint32_t*>(smem + lo_.sel);
    S.sel64 = reinterpret_cast<uint64_t*>(smem + lo_.sel64);

    uint32_t& cand_count = sc32[0];
    float* thr_f_s = reinterpret_cast<float*>(&sc32[2]);
    uint32_t* surv_cnt = &sc32[4];                  // [3] survivors appended by window j: surv_cnt[j % 3]
    uint32_t* thr0_bits = &sc32[8];                 // slice set-up: max of the lists' bounds (non-negative floats order like their bits)
    uint32_t* coef_max_bits = &sc32[9];             // slice set-up: largest filter coefficient of the query's lists

    const int tid = threadIdx.x, lane = tid & 63;
    DIAG_NOW(t_k0);
#ifdef SS_DIAG
    const unsigned long long t_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t slice_id = p.order[blockIdx.x];
    const SliceDesc sd = p.slices[slice_id];
    const uint32_t q = sd.q;
    const uint32_t t0 = p.q_off[q], nd = p.q_off[q + 1] - t0;
    const bool has_phrase = p.ph_off && p.ph_off[q + 1] > p.ph_off[q];
    const int L = (int)(2 * nd) + (has_phrase ? 4 : 0);
    SliceQuery Q;
    Q.qmag = p.qmag[q];
    Q.probs = (p.probs && p.prior) ? p.probs + (size_t)q * p.k_topics : nullptr;
    Q.sqd_ub = Q.probs ? p.sqd_ub[q] : 0.0;
    Q.sqd_ub_f = Q.probs ? __double2float_ru(Q.sqd_ub) : 0.0f;
    Q.qmag_f = (float)Q.qmag;
    // filter: upper bound of the prior's share of FinalRank, 0.33*sqd*100 (get_metadata.go:69), with the filter's margin
    const float r_ub = Q.probs ? __double2float_ru(33.0 * Q.sqd_ub * (1.0 + 0x1p-12)) : 0.0f;
    const bool exact_all = p.exact_all != 0;

    for (int i = tid; i < HT; i += TPB) { S.ht_key[i] = EMPTY; S.ht_rec[i] = EMPTY; }
    for (int i = tid; i < 3 * SK; i += TPB) S.sk[i] = 0u;
    if (tid == 0) { cand_count = 0; *S.overflow = 0; sc64[0] = 0ull; *thr_f_s = -INFINITY; surv_cnt[0] = surv_cnt[1] = surv_cnt[2] = 0; *thr0_bits = 0; *coef_max_bits = 0; }

    // ---- slice set-up: where every list enters and leaves the slice's doc range ----
    uint32_t* t_lo = S.tbl;                 // row 0 of the table
    uint32_t* t_hi = S.f_nxt;               // parked here until n_win is known
    __syncthreads();
    if (tid < L) {
        const int field = tid & 1;                     // 0 = body, 1 = title (also for the phrase lists)
        uint64_t addr, waddr;
        uint32_t len, mult;
        float kth = 0.0f;
        if (tid < (int)(2 * nd)) {
            const uint32_t term = p.dterm[t0 + (tid >> 1)];
            const uint64_t* ptr = field ? p.t_ptr : p.b_ptr;
            const uint64_t p0 = ptr[term];
            addr = (uint64_t)((field ? p.t_rec : p.b_rec) + p0);
            waddr = (uint64_t)((field ? p.t_w : p.b_w) + p0);
            len = (uint32_t)(ptr[term + 1] - p0);
            mult = p.dmult[t0 + (tid >> 1)];
            kth = (field ? p.t_kth : p.b_kth)[(size_t)term * KTH_N + p.kth_j];
        } else {
            // phrase contributions are appended once, after the terms (main_retrieve.go:73-78)
            const int x = tid - (int)(2 * nd);
            addr = (uint64_t)(x_rec_of(p, x) + p.x_off[q]);
            waddr = (uint64_t)(x_w_of(p, x) + p.x_off[q]);
            len = p.x_cnt[(size_t)q * 4 + x];
            mult = 1;
        }
        S.l_rec[tid] = addr;
        S.l_w[tid] = waddr;
        S.l_mult[tid] = mult;
        S.l_field[tid] = field;
        S.f_cur[tid] = len;
        // FinalRank's share of one record is (38|29) * mult * (w/mag) / sqrt(queryLength) (get_metadata.go:57-58,69).
        // Filter coefficient: rounded up with a 2^-12 margin that covers every float rounding of the filter's sums;
        // threshold floor: the same rounded down, times the list's k'-th largest impact (k' >= k distinct documents
        // score at least their own contribution when every other addend is >= 0).
        const double share = (field ? 38.0 : 29.0) * (double)mult / Q.qmag;
        const float coef = __double2float_ru(share * (1.0 + 0x1p-12));
        S.l_coef[tid] = coef;
        if (coef > 0.0f && coef < INFINITY) atomicMax(coef_max_bits, __float_as_uint(coef));
        if (!exact_all && kth > 0.0f) {
            const float floor_l = __double2float_rd(share * (1.0 - 0x1p-12) * (double)kth);
            if (floor_l > 0.0f) atomicMax(thr0_bits, __float_as_uint(floor_l));
        }
    }
#ifdef SS_EXP_FLOOR      // variant build only (tools/floor_exp.py): a per-query floor handed in by the host
    if (tid == 0 && p.q_floor && !exact_all) {
        const float f = p.q_floor[q];
        if (f > 0.0f) atomicMax(thr0_bits, __float_as_uint(f));
    }
#endif
    __syncthreads();
    DIAG_NOW(t_s1);
    DIAG_ADD(6, t_s1 - t_k0);
    if (tid < 2 * L) {                      // the two bounds of a list by two threads: independent search chains
        const int l = tid >> 1;
        if (tid & 1) t_hi[l] = sd.dhi == 0xFFFFFFFFu ? S.f_cur[l] : lower_bound_interp(S.l_rec[l], 0, S.f_cur[l], sd.dhi);
        else t_lo[l] = sd.dlo == 0 ? 0u : lower_bound_interp(S.l_rec[l], 0, S.f_cur[l], sd.dlo);
    }
    __syncthreads();
    DIAG_NOW(t_s2);
    DIAG_ADD(7, t_s2 - t_s1);
    // fixed-point scale of the filter: the largest coefficient (an impact of 1.0 in that list) = FX_ONE units
    const float coef_max = __uint_as_float(*coef_max_bits);
    const float fx_scale = coef_max > 0.0f ? (float)FX_ONE / coef_max : 1.0f;
    const float thr0_f = __uint_as_float(*thr0_bits);
    const uint64_t thr0_key = thr0_f > 0.0f ? fkey((double)thr0_f) : 0ull;
    if (tid < L) S.l_coef[tid] = S.l_coef[tid] * fx_scale * (1.0f + 0x1p-20f);     // filter coefficients in fixed-point units from here on
    const TopK tk{cd_key, cd_doc, &sc32[0], &sc64[0], thr_f_s, thr0_key, thr0_f > 0.0f ? thr0_f : -INFINITY, (uint32_t)p.cb};
    if (tid == 0 && thr0_f > 0.0f) { sc64[0] = thr0_key; *thr_f_s = thr0_f; }
    // every thread: total, driver (longest list in the slice), number of windows
    uint32_t tot = 0, drv = 0, drv_len = 0;
    for (int l = 0; l < L; l++) {
        const uint32_t len = t_hi[l] - t_lo[l];
        tot += len;
        if (len > drv_len) { drv_len = len; drv = l; }
    }
    const bool chunked = L <= LCH;
    int n_win = 0;
    if (tot) {
        if (chunked) {
            // mean 13.5 of the MAXCH = 16 chunks per window: every non-empty list wastes half a chunk per window on
            // average, and the counts fluctuate (sigma ~0.6 chunks); more than MAXCH falls back to the oversize path
            int l_ne = 0;
            for (int l = 0; l < L; l++) l_ne += t_hi[l] != t_lo[l];
            const uint32_t tgt = (uint32_t)max(192, (MAXCH * 64 * 27) / 32 - 34 * l_ne);
            n_win = (int)((tot + tgt - 1) / tgt);
            n_win = min(n_win, min(MAX_WIN, min(TBL_CAP / L - 1, (int)(OFF_CAP * 2 / OSC))));
        } else {
            n_win = (int)((tot + TARGET - 1) / TARGET);
            n_win = min(n_win, min(MAX_WIN, min(TBL_CAP / L - 1, OFF_CAP / (L + 1))));
        }
        n_win = max(n_win, 1);
    }
    const int OS = L + 1;                   // generic loop: row stride of `off`; the window's record count sits in the row's last entry
    // window j covers docs [b_j, b_{j+1}), b_j = doc of the driver's record at j/n_win of its run:
    // all cursors are known up front (no per-window serial planning; windows fill evenly because
    // the other lists simply contribute whatever falls into the driver's doc range)
    const uint64_t drv_addr = S.l_rec[drv];
    for (int idx = tid; idx < (n_win - 1) * L; idx += TPB) {
        const int j = idx / L + 1, l = idx - (j - 1) * L;
        const uint32_t b = load_doc(drv_addr, t_lo[drv] + (uint64_t)j * drv_len / n_win);
        S.tbl[j * L + l] = lower_bound_interp(S.l_rec[l], t_lo[l], t_hi[l], b);
    }
    __syncthreads();
    if (tid < L && n_win) S.tbl[n_win * L + tid] = t_hi[tid];
    __syncthreads();
    if (chunked) {
        uint8_t* off8 = reinterpret_cast<uint8_t*>(S.off);
        for (int j = tid; j < n_win; j += TPB) {
            uint32_t cum = 0, recs = 0;
            for (int l = 0; l < L; l++) {
                off8[j * OSC + l] = (uint8_t)min(cum, 255u);
                const uint32_t len = S.tbl[(j + 1) * L + l] - S.tbl[j * L + l];
                cum += (len + 63u) >> 6;
                recs += len;
            }
            off8[j * OSC + L] = (uint8_t)min(cum, 255u);      // > MAXCH: oversize window
            off8[j * OSC + L + 1] = (uint8_t)min((recs + 7u) >> 3, 255u);   // records / 8, rounded up (<= MAXCH * 64 / 8 = 128 for a regular window)
        }
    } else {
        for (int j = tid; j < n_win; j += TPB) {
            uint32_t run = 0;
            for (int l = 0; l < OS - 1; l++) {
                S.off[j * OS + l] = (uint16_t)min(run, 0xFFFFu);
                run += S.tbl[(j + 1) * L + l] - S.tbl[j * L + l];
            }
            S.off[j * OS + OS - 1] = (uint16_t)min(run, 0xFFFFu);      // > CAP: oversize window
        }
    }
    __syncthreads();

    DIAG_NOW(t_k1);
    DIAG_ADD(0, 1);
    DIAG_ADD(1, n_win);
    DIAG_ADD(5, tot);
    DIAG_ADD(9, t_k1 - t_k0);
    // ---- the windows ----
    // the threshold in the filter's units: it only moves when the exact stage runs (recomputed after every flush)
    uint32_t thr_fx = exact_all ? 0u : fx_threshold(*thr_f_s, r_ub, fx_scale);
    uint32_t pbase = 0;                     // pending survivors of the windows before this one (same value in every thread)
    int cur = 0, prv = 2;                   // j % 3, (j - 1) % 3
    if (chunked && n_win > 0) {
        const int wave = tid >> 6;
        WaveLists wl{0u, 0u, 0.f};
        if (lane < L) {
            const uint64_t a = S.l_rec[lane];
            wl.rec_lo = (uint32_t)a;
            wl.rec_hi = (uint32_t)(a >> 32);
            wl.coef = S.l_coef[lane];
        }
        u32x2 rec[DEPTH][CPW];
        uint32_t cnt[DEPTH][CPW];
        float cf[DEPTH][CPW];
        uint32_t hprev[CPW];
#pragma unroll
        for (int r = 0; r < CPW; r++) {
            hprev[r] = EMPTY;
#pragma unroll
            for (int d = 0; d < DEPTH; d++) { rec[d][r] = u32x2{0u, 0u}; cnt[d][r] = 0; cf[d][r] = 0.f; }
        }
        const uint8_t* off8 = reinterpret_cast<const uint8_t*>(S.off);
        // windows 0 .. DEPTH-1 go in flight before the loop; window j + DEPTH is requested at the end of step j.
        // The loop is unrolled by DEPTH so that ring slots and filter tables are compile-time constants; the window
        // count is padded to a multiple of DEPTH with empty windows (no records, same instruction sequence).
        const uint64_t dummy = (uint64_t)p.q_off;      // 8 readable bytes for lanes without a record
        chunk_issue<0>(win_row(S, L, 0, n_win, lane), wl, L, wave, lane, dummy, rec, cnt, cf);
        chunk_issue<1>(win_row(S, L, 1, n_win, lane), wl, L, wave, lane, dummy, rec, cnt, cf);
        chunk_issue<2>(win_row(S, L, 2, n_win, lane), wl, L, wave, lane, dummy, rec, cnt, cf);
#define SS_CHUNK_STEP(S_, JJ)                                                                                              \
        {                                                                                                                  \
            const int j_ = (JJ);                                                                                           \
            constexpr int nxt_ = (S_ + 1) % 3, prv_ = (S_ + 2) % 3;                                                        \
            const uint32_t n_ch = j_ < n_win ? off8[j_ * OSC + L] : 0u;                                                    \
            const uint32_t n_rec = j_ < n_win ? 8u * off8[j_ * OSC + L + 1] : 0u;                                          \
            const bool normal = n_ch <= (uint32_t)MAXCH;                                                                   \
            uint32_t* sk_cur = S.sk + S_ * SK;                                                                             \
            uint32_t* sk_prv = S.sk + prv_ * SK;                                                                           \
            uint32_t h[CPW], u[CPW];                                                                                       \
            /* stage 1a: every record adds its share into its doc's slot (waits for THIS window's records only) */        \
            DIAG_NOWL(t_a0_);                                                                                              \
            chunk_add<S_>(sk_cur, lane, rec, cnt, cf, h);                                                                  \
            DIAG_NOWL(t_a1_);                                                                                              \
            lds_barrier();                                                                                                 \
            DIAG_NOWL(t_a2_);                                                                                              \
            DIAG_ADDL(12, t_a1_ - t_a0_);                                                                                   \
            DIAG_ADDL(13, t_a2_ - t_a1_);                                                                                   \
            /* ONE round of LDS latency for everything the rest of the step needs: the survivor count of the previous */  \
            /* window (complete now, stable until its counter is re-used), this window's slots, the plan row of the   */  \
            /* window that is requested next                                                                          */  \
            const uint32_t sc_ = surv_cnt[prv_];                                                                           \
            _Pragma("unroll") for (int r = 0; r < CPW; r++) u[r] = sk_cur[h[r] & (uint32_t)(SK - 1)];                      \
            const WinRow row_ = win_row(S, L, j_ + DEPTH, n_win, lane);                                                    \
            pbase += sc_;                                                                                                  \
            if (tid == 0) surv_cnt[nxt_] = 0;                                                                              \
            if (!normal || pbase + n_rec > (uint32_t)PC) {                                                                 \
                if (pbase) flush_pending(S, tk, Q, p, tid, pbase);                                                         \
                pbase = 0;                                                                                                 \
                if (!normal) oversize_window(S, tk, Q, p, tid, L, j_, n_win, sd, drv, drv_addr);                           \
                thr_fx = exact_all ? 0u : fx_threshold(*thr_f_s, r_ub, fx_scale);                                          \
            }                                                                                                              \
            /* stage 1b: the slot bounds the doc's FinalRank from above; below the threshold -> drop.  Then the records */ \
            /* of window j + DEPTH are requested into the registers this window has just released.                    */ \
            chunk_filter<S_>(S, u, sk_prv, &surv_cnt[S_], L, j_, n_win, wave, lane, thr_fx, pbase, rec, h, hprev);         \
            chunk_issue<S_>(row_, wl, L, wave, lane, dummy, rec, cnt, cf);                                                 \
        }
        for (int j = 0; j < n_win; j += DEPTH) {
            SS_CHUNK_STEP(0, j)
            SS_CHUNK_STEP(1, j + 1)
            SS_CHUNK_STEP(2, j + 2)
        }
#undef SS_CHUNK_STEP
        prv = 2;                                // the padded window count is a multiple of 3: the last step used counter 2
    } else if (!chunked) {
        u32x2 rec[PPT];
        uint32_t rl[PPT], ri[PPT];
        float rc[PPT];
#pragma unroll
        for (int r = 0; r < PPT; r++) { rl[r] = EMPTY; ri[r] = 0; rc[r] = 0.f; rec[r] = u32x2{0u, 0u}; }
        uint32_t hprev[PPT];                // filter slots this thread touched in the previous window (EMPTY: none)
#pragma unroll
        for (int r = 0; r < PPT; r++) hprev[r] = EMPTY;
        bool have = false;                  // window j's records are in flight / in registers
        for (int j = 0; j < n_win; j++) {
            const uint32_t n = S.off[j * OS + OS - 1];
            const int nxt = cur == 2 ? 0 : cur + 1;
            const bool normal = n <= (uint32_t)CAP;
            if (normal && !have) load_window(S, j, L, OS, tid, rec, rl, ri, rc);     // first window, or the one after an oversize window
            have = false;
            uint32_t h[PPT];
            u32x2 mine[PPT];
            uint32_t ml[PPT], mi[PPT];
            uint32_t* sk_cur = S.sk + cur * SK;
            if (normal) {
                // stage 1a: every record adds its share into its doc's slot (waits for window j's records)
#pragma unroll
                for (int r = 0; r < PPT; r++) {
                    h[r] = EMPTY;
                    if (rl[r] != EMPTY) {
                        h[r] = fx_slot(rec[r].x);
                        atomicAdd(&sk_cur[h[r]], fx_share(__uint_as_float(rec[r].y), rc[r]));
                    }
                    mine[r] = rec[r]; ml[r] = rl[r]; mi[r] = ri[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < PPT; r++) { h[r] = EMPTY; ml[r] = EMPTY; mi[r] = 0; mine[r] = u32x2{0u, 0u}; }
            }
            lds_barrier();
            // survivors of window j-1 are all appended now; their count is stable until window j+2 re-uses the counter
            pbase += surv_cnt[prv];
            if (tid == 0) surv_cnt[nxt] = 0;
            if (!normal || pbase + n > (uint32_t)PC) {
                if (pbase) flush_pending(S, tk, Q, p, tid, pbase);
                pbase = 0;
                if (!normal) oversize_window(S, tk, Q, p, tid, L, j, n_win, sd, drv, drv_addr);
                thr_fx = exact_all ? 0u : fx_threshold(*thr_f_s, r_ub, fx_scale);
            }
            // put the next window's loads in flight
            if (j + 1 < n_win && S.off[(j + 1) * OS + OS - 1] <= CAP) {
                load_window(S, j + 1, L, OS, tid, rec, rl, ri, rc);
                have = true;
            }
            // stage 1b: the slot now bounds the doc's FinalRank from above; below the threshold -> drop.
            // (An oversize window has no records here: only the previous window's slots are cleared.)
            uint32_t* sk_prv = S.sk + prv * SK;
#pragma unroll
            for (int r = 0; r < PPT; r++) {
                bool surv = false;
                if (ml[r] != EMPTY) surv = sk_cur[h[r]] >= thr_fx;
                if (hprev[r] != EMPTY) sk_prv[hprev[r]] = 0u;
                hprev[r] = h[r];
                const unsigned long long m = __ballot(surv);
                if (m) {
                    const int leader = __ffsll((long long)m) - 1;
                    uint32_t b = 0;
                    if (lane == leader) b = atomicAdd(&surv_cnt[cur], (uint32_t)__popcll(m));
                    b = (uint32_t)__shfl((int)b, leader, 64);
                    if (surv) {
                        const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
                        S.pend[pbase + b + (uint32_t)__popcll(m & below)] = make_uint4(mine[r].x, mi[r], ml[r], 0u);
                    }
                }
            }
            prv = cur;
            cur = nxt;
        }
    }
    lds_barrier();
    pbase += surv_cnt[prv];
    if (pbase) flush_pending(S, tk, Q, p, tid, pbase);
    DIAG_NOW(t_k2);
    DIAG_ADD(10, t_k2 - t_k1);

    topk_cut(tk, p.k, S.sel, S.sel64);  // (the slice's list goes to the merge unordered: merge_query filters and sorts what it gathers)
    const uint32_t n_out = cand_count;
    if (!p.q_ticket) {
        for (uint32_t i = tid; i < n_out; i += TPB) {
            p.so_key[(size_t)slice_id * p.k + i] = cd_key[i];
            p.so_doc[(size_t)slice_id * p.k + i] = cd_doc[i];
        }
        if (tid == 0) p.so_cnt[slice_id] = n_out;
    } else {
        // Fused merge: the slice's list goes out write-through (sc1), every wave drains its stores, one lane takes the
        // query's ticket; the slice that arrives last merges the query right here while other workgroups are still
        // scoring — no second launch, no fence (the readers use sc1 loads; MI355X_MICROARCH.md, hand-off forms).
        for (uint32_t i = tid; i < n_out; i += TPB) {
            __hip_atomic_store(&p.so_key[(size_t)slice_id * p.k + i], cd_key[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&p.so_doc[(size_t)slice_id * p.k + i], cd_doc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) __hip_atomic_store(&p.so_cnt[slice_id], n_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        uint32_t* last_flag = reinterpret_cast<uint32_t*>(smem);   // the slice's LDS is free from here on
        if (tid == 0) {
            const uint32_t n_sl = p.slice_base[sd.q + 1] - p.slice_base[sd.q];
            const uint32_t prev = __hip_atomic_fetch_add(&p.q_ticket[sd.q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = prev == n_sl - 1;
            if (last) __hip_atomic_store(&p.q_ticket[sd.q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
            *last_flag = last ? 1u : 0u;
        }
        __syncthreads();
        const bool last = *last_flag != 0;
        __syncthreads();
        if (last) merge_query<TPB, true>(p, sd.q, smem, p.cb);
    }
    DIAG_NOW(t_k3);
    DIAG_ADD(11, t_k3 - t_k0);
#ifdef SS_DIAG
    if (tid == 0 && blockIdx.x < 4096) {
        g_slice[blockIdx.x][0] = t_r0;
        g_slice[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
        g_slice[blockIdx.x][2] = (unsigned long long)n_win;
        g_slice[blockIdx.x][3] = tot;
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
//   * matches leave as scoring records {doc, impact of the float32 sum} + the sum itself, in doc order (ordered
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

// float32 upper bound of w / mag for the filter; 0 where the weight is 0 (0/0 = NaN -> 0 in the reference too)
__device__ __forceinline__ float impact_of(float w, double mag) {
    if (w == 0.0f) return 0.0f;
    return __double2float_ru((double)w / mag);
}

constexpr uint32_t PH_PART = 8192;   // candidates per k_phrase_match workgroup
__global__ __launch_bounds__(PH_TPB) void k_phrase_match(ScoreParams p) {
    __shared__ int32_t s_pb[PH_MAX * PH_TPB];     // body posting index of term i for this thread's doc, -1 = none
    __shared__ int32_t s_pt[PH_MAX * PH_TPB];
    __shared__ uint32_t s_wave[2][PH_TPB / 64];
    __shared__ uint32_t s_base[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint4 part = p.ph_parts[blockIdx.x];
    const uint32_t q = part.x;
    const int pass = (int)part.y;
    const uint32_t f0 = p.ph_off[q], m = p.ph_off[q + 1] - f0;
    const uint32_t drv_term = p.ph_terms[f0 + p.ph_drv[q]];
    const uint64_t* pos_ptr2[2] = {p.b_pos_ptr, p.t_pos_ptr};
    const float* pos2[2] = {p.b_pos, p.t_pos};
    int32_t* my_pb = s_pb + tid;
    int32_t* my_pt = s_pt + tid;
    if (tid < 2) s_base[tid] = 0;
    __syncthreads();
    // pass 0: driver's body postings; pass 1: driver's title postings whose doc has no driver body posting
    const uint64_t l0 = pass == 0 ? p.b_ptr[drv_term] : p.t_ptr[drv_term];
    const uint64_t c0 = l0 + part.z, c1 = c0 + part.w;
    const Rec* crec = pass == 0 ? p.b_rec : p.t_rec;
    const uint32_t out0 = p.x_off[q] + part.z;    // the part's matches go out compactly from its first candidate's slot
    for (uint64_t cb = c0; cb < c1; cb += PH_TPB) {
        const uint64_t ci = cb + tid;
        bool body_ok = false, title_ok = false;
        float sum_b = 0.0f, sum_t = 0.0f;
        uint32_t d = 0;
        if (ci < c1) {
            d = crec[ci].doc;
            bool all = true, body_all = true, title_all = true;
            if (pass == 1) {
                const uint64_t b0 = p.b_ptr[drv_term], b1 = p.b_ptr[drv_term + 1];
                const uint64_t pos = lower_bound_rec_interp(p.b_rec, b0, b1, d);
                if (pos < b1 && p.b_rec[pos].doc == d) all = false;       // already handled in pass 0
            }
            for (uint32_t i = 0; i < m && all; i++) {
                const uint32_t term = p.ph_terms[f0 + i];
                const uint64_t b0 = p.b_ptr[term], b1 = p.b_ptr[term + 1];
                const uint64_t t0 = p.t_ptr[term], t1 = p.t_ptr[term + 1];
                const uint64_t pb = lower_bound_rec_interp(p.b_rec, b0, b1, d);
                const uint64_t pt = lower_bound_rec_interp(p.t_rec, t0, t1, d);
                const bool hb = pb < b1 && p.b_rec[pb].doc == d, ht = pt < t1 && p.t_rec[pt].doc == d;
                my_pb[i * PH_TPB] = hb ? (int32_t)pb : -1;
                my_pt[i * PH_TPB] = ht ? (int32_t)pt : -1;
                if (!hb && !ht) all = false;                      // phrase.go:63
                if (hb) sum_b += p.b_w[pb]; else body_all = false;          // phrase.go:69,80-84
                if (ht) sum_t += p.t_w[pt]; else title_all = false;         // phrase.go:73,87-91
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
            const uint32_t o = out0 + ob + (uint32_t)__popcll(mb & below);
            const int x = pass == 0 ? 0 : 2;
            Rec r;
            r.doc = d; r.imp = impact_of(sum_b, p.b_mag[d]);
            x_rec_of(p, x)[o] = r;
            x_w_of(p, x)[o] = sum_b;
        }
        if (title_ok) {
            const uint32_t o = out0 + ot + (uint32_t)__popcll(mt & below);
            const int x = pass == 0 ? 1 : 3;
            Rec r;
            r.doc = d; r.imp = impact_of(sum_t, p.t_mag[d]);
            x_rec_of(p, x)[o] = r;
            x_w_of(p, x)[o] = sum_t;
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
    if (tid < 2) p.ph_pcnt[(size_t)blockIdx.x * 2 + tid] = s_base[tid];
}

// Per query: the parts' match runs, each compact from its own offset, are moved down into one contiguous doc-sorted list
// per result kind (in part order = doc order), and the four counts are written.  Only matches move.
__global__ __launch_bounds__(PH_TPB) void k_phrase_close(ScoreParams p) {
    const uint32_t q = blockIdx.x;
    const int tid = threadIdx.x;
    uint32_t run[4] = {0u, 0u, 0u, 0u};
    const uint32_t base = p.x_off[q];
    for (uint32_t pi = p.ph_pbase[q]; pi < p.ph_pbase[q + 1]; pi++) {
        const uint4 part = p.ph_parts[pi];
        for (int kind = 0; kind < 2; kind++) {
            const int x = (int)part.y * 2 + kind;
            const uint32_t n = p.ph_pcnt[(size_t)pi * 2 + kind];
            const uint32_t src = base + part.z, dst = base + run[x];
            if (n && src != dst) {
                Rec* xr = x_rec_of(p, x);
                float* xw = x_w_of(p, x);
                // dst < src and the ranges may overlap: a chunk is read whole before it is written
                for (uint32_t i0 = 0; i0 < n; i0 += PH_TPB) {
                    const uint32_t i = i0 + (uint32_t)tid;
                    Rec r{};
                    float w = 0.f;
                    if (i < n) { r = xr[src + i]; w = xw[src + i]; }
                    __syncthreads();
                    if (i < n) { xr[dst + i] = r; xw[dst + i] = w; }
                    __syncthreads();
                }
            }
            run[x] += n;
        }
    }
    if (tid < 4) p.x_cnt[(size_t)q * 4 + tid] = run[tid];
}

size_t score_lds_bytes(int cb) { return score_lds_layout(cb).total; }

// ---- K5: merge a query's slices, explain the winners ------------------------------
__global__ __launch_bounds__(TPB_M) void k_merge_topk(ScoreParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (p.q_fast && p.q_fast[blockIdx.x]) return;          // k_merge_flat's (bit 0) or k_score_small's (bit 1): their hits come from there
    merge_query<TPB_M, false>(p, blockIdx.x, smem, p.cb);
}
// the queries scored by k_score_wave: one candidate list per query
#ifndef SS_TPB_MF
#define SS_TPB_MF 320
#endif
constexpr int TPB_MF = SS_TPB_MF;   // 320: the 300 explain searches of a 3-term query at k = 100 are one pass (ms per batch at 256 / 320 / 384 / 512 threads: 0.405 / 0.399 / 0.410 / 0.420)
__global__ __launch_bounds__(TPB_MF) void k_merge_flat(ScoreParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    merge_query<TPB_MF, false, true>(p, p.merge_q[blockIdx.x], smem, p.cb_flat);
}

size_t merge_lds_bytes(int k, int cb) { return (size_t)k * 32 + (size_t)cb * 12 + 8 + 16 + 16 + (size_t)MERGE_T * 20; }

// scoring layout: {doc, float32 upper bound of w/mag[doc]} per posting; flags: bit 0 = a weight is negative or not finite,
// bit 1 = a magnitude is not a positive finite number under a non-zero weight (the filter's assumptions, see k_score_slices)
__global__ void k_pack_recs(const uint32_t* __restrict__ doc, const float* __restrict__ w, const double* __restrict__ mag,
                            uint64_t n, Rec* __restrict__ out, uint32_t* __restrict__ flags) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t f = 0;
    for (; i < n; i += stride) {
        Rec r;
        r.doc = doc[i];
        const float wi = w[i];
        const double m = mag[r.doc];
        if (!(wi >= 0.0f) || isinf(wi)) f |= 1u;
        if (wi != 0.0f && (!(m > 0.0) || isinf(m))) f |= 2u;
        r.imp = impact_of(wi, m);
        out[i] = r;
    }
    if (__ballot(f != 0)) {
        for (int o = 32; o > 0; o >>= 1) f |= (uint32_t)__shfl_xor((int)f, o, 64);
        if ((threadIdx.x & 63) == 0) atomicOr(flags, f);
    }
}

// Combined lists for k_score_wave: the title and body postings of a term merged into ONE doc-sorted list (body before title
// on the same doc), field in bit 31 of the doc word.  c_ptr[t] = t_ptr[t] + b_ptr[t]; c_w[i] = the posting's float32 weight, copied
// from its table (the exact stage reads it by the record's position: one load, where an index into the table's own list cost a
// second, dependent one per survivor).  One thread per posting of one table: its place is its own
// index plus the number of the OTHER field's postings of the term that come before it.  A block finds the terms its
// postings span with two binary searches; each posting then finds its term inside that short range.
constexpr int CM_TPB = 256, CM_PT = 8, CM_CHUNK = CM_TPB * CM_PT;
__global__ __launch_bounds__(CM_TPB) void k_merge_lists(const uint64_t* __restrict__ my_ptr, const Rec* __restrict__ my_rec, uint64_t n_my,
                                                        const uint64_t* __restrict__ ot_ptr, const Rec* __restrict__ ot_rec, uint64_t n_terms,
                                                        int my_field, const float* __restrict__ my_w, Rec* __restrict__ c_rec, float* __restrict__ c_w) {
    __shared__ uint64_t s_t[2];
    const uint64_t base = (uint64_t)blockIdx.x * CM_CHUNK;
    if (base >= n_my) return;
    const uint64_t last = min(base + CM_CHUNK, n_my) - 1;
    if (threadIdx.x < 2) {
        const uint64_t target = threadIdx.x == 0 ? base : last;      // largest t with my_ptr[t] <= target
        uint64_t lo = 0, hi = n_terms;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (my_ptr[mid] <= target) lo = mid; else hi = mid;
        }
        s_t[threadIdx.x] = lo;
    }
    __syncthreads();
    const uint64_t t_lo = s_t[0], t_hi = s_t[1];
    for (int j = 0; j < CM_PT; j++) {
        const uint64_t i = base + (uint64_t)j * CM_TPB + threadIdx.x;
        if (i > last) break;
        uint64_t lo = t_lo, hi = t_hi + 1;                            // my_ptr[lo] <= i < my_ptr[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (my_ptr[mid] <= i) lo = mid; else hi = mid;
        }
        const uint64_t t = lo;
        const Rec r = my_rec[i];
        // postings of the other field before this one: body (field 0) goes first on the same doc
        const uint64_t o0 = ot_ptr[t], o1 = ot_ptr[t + 1];
        const uint64_t before = lower_bound_rec(ot_rec, o0, o1, my_field == 0 ? r.doc : r.doc + 1u) - o0;
        const uint64_t pos = my_ptr[t] + o0 + (i - my_ptr[t]) + before;       // c_ptr[t] = my_ptr[t] + ot_ptr[t]
        Rec c = r;
        c.doc = r.doc | ((uint32_t)my_field << 31);
        // a title posting's impact is stored times 38/29 (rounded up twice): ONE filter coefficient per list then serves both
        // fields in k_score_wave's inner loop (get_metadata.go:69 weighs title 0.38, body 0.29) — still an upper bound
        if (my_field) {
            const float v = r.imp * 1.3103449f;                          // 1.3103449f > 38/29; then two ulps up: above the real product
            c.imp = v > 0.0f ? __uint_as_float(__float_as_uint(v) + 2u) : v;
        }
        c_rec[pos] = c;
        c_w[pos] = my_w[i];
    }
}
// pad behind the combined records ({doc 0x7FFFFFFF body, impact 0}: in no window) and the skip index: c_skip[g] = doc of record 64*g
__global__ void k_combined_finish(Rec* __restrict__ c_rec, uint64_t n, uint32_t* __restrict__ c_skip) {
    const uint64_t n_pad = ((n + 63) & ~(uint64_t)63) + 64;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (uint64_t)gridDim.x * blockDim.x) {
        if (i >= n) { Rec r; r.doc = 0x7FFFFFFFu; r.imp = 0.0f; c_rec[i] = r; }
        if ((i & 63) == 0) c_skip[i >> 6] = c_rec[i].doc & 0x7FFFFFFFu;
    }
}
__global__ void k_add_ptr_u64(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint64_t n, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// k'-th largest impact of every term's list for k' = 2^0 .. 2^10 (0 where the list is shorter), as LOWER bounds:
// a 4096-bin histogram over [2^-8, 1) (9 mantissa bits per binade: 0.2 % resolution; smaller impacts share bin 0,
// whose edge is 0), suffix-summed from the top; the bin edge at which the count reaches k' is the bound.
constexpr int KH_TPB = 256;
constexpr int KH_BINS = 4096;
constexpr uint32_t KH_B0 = 0x3B800000u;               // bits of 2^-8
__global__ __launch_bounds__(KH_TPB) void k_kth_impact(const uint64_t* __restrict__ term_ptr, const Rec* __restrict__ rec,
                                                        float* __restrict__ kth /*[T][KTH_N], zeroed*/) {
    __shared__ uint32_t s_hist[KH_BINS];
    __shared__ uint32_t s_part[KH_TPB];
    const uint64_t t = blockIdx.x;
    const uint64_t beg = term_ptr[t], end = term_ptr[t + 1];
    if (end == beg) return;
    for (int b = threadIdx.x; b < KH_BINS; b += KH_TPB) s_hist[b] = 0;
    __syncthreads();
    for (uint64_t i = beg + threadIdx.x; i < end; i += KH_TPB) {
        // stored impacts are rounded UP from w/mag: step down two float ulps for a lower bound
        const float lb = rec[i].imp * (1.0f - 0x1p-21f);
        const uint32_t xb = __float_as_uint(lb);
        uint32_t b = 0;
        if (lb > 0.0f && xb >= KH_B0) b = min((xb - KH_B0) >> 14, (uint32_t)(KH_BINS - 1));
        if (lb > 0.0f) atomicAdd(&s_hist[b], 1u);                    // NaN and non-positive impacts count for nothing
    }
    __syncthreads();
    // thread x owns bins [16x, 16x+16): its sum, then the number of postings in bins ABOVE its range
    constexpr int PER = KH_BINS / KH_TPB;
    uint32_t own = 0;
    for (int b = 0; b < PER; b++) own += s_hist[threadIdx.x * PER + b];
    s_part[threadIdx.x] = own;
    __syncthreads();
    uint32_t above = 0;
    for (int x = threadIdx.x + 1; x < KH_TPB; x++) above += s_part[x];
    uint32_t run = above;
    for (int b = PER - 1; b >= 0; b--) {
        const uint32_t bin = threadIdx.x * PER + b;
        const uint32_t c = s_hist[bin];
        if (c) {
            for (int j = 0; j < KTH_N; j++) {
                const uint32_t kk = 1u << j;
                if (run < kk && kk <= run + c) kth[t * KTH_N + j] = bin == 0 ? 0.0f : __uint_as_float(KH_B0 + (bin << 14));
            }
        }
        run += c;
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

namespace ss {
// score_wave.hip
size_t score_wave_prep_bytes(unsigned n_slices);
void launch_wave_prep(const void* params, unsigned n_slices, void* prep, hipStream_t st);
void launch_score_wave(const void* params, unsigned n_slices, const void* prep, hipStream_t st);
int score_wave_max_lists();
int score_wave_max_k();
// score_small.hip: one workgroup per small query (every posting scored exactly, hits written by the kernel itself)
uint32_t score_small_cap();
uint32_t score_small_cap_a();
int score_small_max_k();
int score_small_max_lists();
int32_t launch_score_small(const void* params, unsigned n_a, unsigned n_b, hipStream_t st);
void launch_small_copy(const void* params, unsigned n_small, hipStream_t st);
void score_small_report();
void score_wave_diag_dump();
}  // namespace ss

struct ss_scorer {
    ss_ctx* ctx = nullptr;
    ss_index* title = nullptr;
    ss_index* body = nullptr;
    uint64_t n_docs = 0, n_terms = 0;
    ss::DevBuf<Rec> t_rec, b_rec;              // scoring records {doc, impact}
    // combined lists (k_score_wave): title + body postings of a term merged by doc, field in bit 31 of the doc word
    ss::DevBuf<Rec> c_rec;
    ss::DevBuf<uint32_t> c_skip;
    ss::DevBuf<float> c_w;
    ss::DevBuf<uint64_t> c_ptr;
    bool has_combined = false;
    uint64_t c_pad_block = 0;
    ss::DevBuf<float> t_kth, b_kth;             // [T][KTH_N] k'-th largest impact per term (threshold floor)
    bool clean = true;                          // weights >= 0 and finite, magnitudes positive and finite where a weight is not 0
    bool prior_clean = true;                    // every prior value >= 0 and finite
    ss::DevBuf<double> prior;
    std::vector<double> prior_max, prior_min;   // per topic
    int k_topics = 0;
    int lds_attr = 0;
    // per-call workspaces, grow-only (no hipMalloc/hipFree on the steady-state query path)
    // Turns of per-batch buffers: the host runs at most TURNS batches ahead.  (Three were measured for the pipelined mode, so that a
    // batch's plan upload and k_wave_prep — which do not fit beside k_score_wave's three waves of 168 VGPRs per SIMD — are enqueued
    // one batch earlier: 0.395 against 0.399 ms per batch, not worth a third set of buffers.)
#ifndef SS_TURNS
#define SS_TURNS 3
#endif
    static constexpr int TURNS = SS_TURNS;
    unsigned wave_turn = 0;                    // which wave stream the next pipelined batch takes
    ss::DevBuf<unsigned char> d_plan2[TURNS], d_wprep2[TURNS];   // the plan on the device, one buffer per turn: batch i+1's upload runs beside batch i's kernels
    // pinned staging for the plan, double-buffered: a call that returns results in device memory does not wait
    // for the GPU, so the next call plans (and fills the other buffer) while this one's copy and kernels run
    unsigned char* h_plan[TURNS] = {};
    size_t h_plan_cap[TURNS] = {};
    std::vector<float> dbg_floor;    // experiment "score.debug_floor": the k-th best FinalRank of every query of the last host-output call, rounded down
    unsigned char* h_res = nullptr;  // pinned landing block of small host results (one device-to-host copy for hits + counts)
    static constexpr size_t H_RES_BYTES = 128 << 10;
    hipEvent_t plan_ev[TURNS] = {}; // recorded after the H2D copy of the buffer (on the context's second stream)
    hipEvent_t batch_ev[TURNS] = {};// recorded behind the kernels of the batch that read device buffer [turn]
    bool batch_ev_pending[TURNS] = {};
    size_t qcnt_zeroed2[TURNS] = {};           // counters known to be zero (k_merge_flat leaves its query's counter at zero)
    bool plan_ev_pending[TURNS] = {};
    int plan_turn = 0;
    ss::DevBuf<Rec> d_x[TURNS][4];              // phrase result lists: scoring records (one set per turn: batches overlap)
    ss::DevBuf<float> d_xw[TURNS][4];           // ... and their float32 weight sums
    ss::DevBuf<uint32_t> d_xcnt[TURNS], d_pcnt[TURNS];
    ss::DevBuf<uint64_t> d_so_key2[TURNS];           // the slices' candidates, one set per turn ("score.pipeline": batch i's merge reads its set while batch i+1 fills the other)
    ss::DevBuf<uint32_t> d_so_doc2[TURNS], d_so_cnt2[TURNS], d_qticket, d_qcnt2[TURNS];
    ss::DevBuf<ss_hit> d_small_stage[TURNS];         // k_score_small's rows of a pipelined batch (k_small_copy moves them on the caller's stream)
    ss::DevBuf<int32_t> d_small_stage_n[TURNS];
    hipEvent_t wave_ev[TURNS] = {};  // "score.pipeline": behind k_score_wave on the context's wave stream; the merge on the caller's stream waits for it
    hipEvent_t slice_ev[TURNS] = {}; // ... and behind the k_score_slices part of a split batch on ANOTHER wave stream
    size_t qticket_zeroed = 0;         // tickets known to be zero (every fused call leaves them so)
    ss::DevBuf<ss_hit> d_hits;
    // ss_score_topk_submit / _collect: batches in flight whose hits go to HOST memory.  A slot: device buffers the kernels write and
    // an event behind them.
    static constexpr int INFLIGHT = SS_SCORE_INFLIGHT;
    struct AsyncSlot {
        ss::DevBuf<ss_hit> hits;
        ss::DevBuf<int32_t> n_hits;
        hipEvent_t ev = nullptr;             // behind the batch's kernels on the caller's stream
        void* pin = nullptr;                 // "score.collect_pinned": the copy-out lands here first
        size_t pin_cap = 0;
        bool pin_mode = false;
        uint64_t ticket = 0;                 // 0 = free
        bool collecting = false;             // a collect call is waiting for / copying this slot outside the lock
        void* pin_n = nullptr;               // pinned landing block of the counts (a small copy into pageable memory costs ~20 us more)
        size_t pin_n_cap = 0;
        int32_t n_q = 0, k = 0;
    } aslot[INFLIGHT];
    uint64_t next_ticket = 1;
    hipStream_t out_stream = nullptr;        // collect's copies
    ss::DevBuf<int32_t> d_nhits;
    ~ss_scorer() {
        if (out_stream) { (void)hipStreamSynchronize(out_stream); (void)hipStreamDestroy(out_stream); }
        for (auto& a : aslot) {
            if (a.ev) (void)hipEventDestroy(a.ev);
            if (a.pin) ctx->pin_free(a.pin, a.pin_cap);
            if (a.pin_n) ctx->pin_free(a.pin_n, a.pin_n_cap);
        }
        for (int i = 0; i < TURNS; i++) {
            if (h_plan[i]) (void)hipHostFree(h_plan[i]);
            if (i == 0 && h_res) (void)hipHostFree(h_res);
            if (plan_ev[i]) (void)hipEventDestroy(plan_ev[i]);
            if (batch_ev[i]) (void)hipEventDestroy(batch_ev[i]);
            if (wave_ev[i]) (void)hipEventDestroy(wave_ev[i]);
            if (slice_ev[i]) (void)hipEventDestroy(slice_ev[i]);
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
    SS_HIP(ctx, s->t_rec.alloc(title->n_post));
    SS_HIP(ctx, s->b_rec.alloc(body->n_post));
    SS_HIP(ctx, s->t_kth.alloc((size_t)s->n_terms * KTH_N));
    SS_HIP(ctx, s->b_kth.alloc((size_t)s->n_terms * KTH_N));
    ss::DevBuf<uint32_t> flags;
    SS_HIP(ctx, flags.alloc(1));
    SS_HIP(ctx, hipMemsetAsync(flags.p, 0, sizeof(uint32_t), ctx->stream));
    SS_HIP(ctx, hipMemsetAsync(s->t_kth.p, 0, std::max<size_t>(s->t_kth.bytes(), 4), ctx->stream));
    SS_HIP(ctx, hipMemsetAsync(s->b_kth.p, 0, std::max<size_t>(s->b_kth.bytes(), 4), ctx->stream));
    if (title->n_post)
        hipLaunchKernelGGL(k_pack_recs, dim3(std::min<unsigned>(ss::div_up(title->n_post, 256), 16384u)), dim3(256), 0, ctx->stream,
                           (const uint32_t*)title->post_doc.p, (const float*)title->post_w.p, (const double*)title->mag.p,
                           title->n_post, s->t_rec.p, flags.p);
    if (body->n_post)
        hipLaunchKernelGGL(k_pack_recs, dim3(std::min<unsigned>(ss::div_up(body->n_post, 256), 16384u)), dim3(256), 0, ctx->stream,
                           (const uint32_t*)body->post_doc.p, (const float*)body->post_w.p, (const double*)body->mag.p,
                           body->n_post, s->b_rec.p, flags.p);
    {
        // combined lists: only while doc ids leave bit 31 free and the table stays below 2^32 postings (the wave kernel's limits)
        const uint64_t pc = title->n_post + body->n_post;
        if (s->n_docs < (1ull << 31) && pc < (1ull << 32) - 128 && s->n_terms && ctx->opt("score.wave", 1) != 0) {
            const uint64_t pc_pad = ((pc + 63) & ~(uint64_t)63) + 64;
            SS_HIP(ctx, s->c_rec.alloc(pc_pad));
            SS_HIP(ctx, s->c_w.alloc(pc_pad));
            SS_HIP(ctx, s->c_skip.alloc(pc_pad / 64 + 1));
            SS_HIP(ctx, s->c_ptr.alloc(s->n_terms + 1));
            hipLaunchKernelGGL(k_add_ptr_u64, dim3(ss::div_up(s->n_terms + 1, 256)), dim3(256), 0, ctx->stream, (const uint64_t*)title->term_ptr.p,
                               (const uint64_t*)body->term_ptr.p, s->n_terms + 1, s->c_ptr.p);
            if (body->n_post)
                hipLaunchKernelGGL(k_merge_lists, dim3(ss::div_up(body->n_post, CM_CHUNK)), dim3(CM_TPB), 0, ctx->stream, (const uint64_t*)body->term_ptr.p,
                                   (const Rec*)s->b_rec.p, body->n_post, (const uint64_t*)title->term_ptr.p, (const Rec*)s->t_rec.p, s->n_terms, 0,
                                   (const float*)body->post_w.p, s->c_rec.p, s->c_w.p);
            if (title->n_post)
                hipLaunchKernelGGL(k_merge_lists, dim3(ss::div_up(title->n_post, CM_CHUNK)), dim3(CM_TPB), 0, ctx->stream, (const uint64_t*)title->term_ptr.p,
                                   (const Rec*)s->t_rec.p, title->n_post, (const uint64_t*)body->term_ptr.p, (const Rec*)s->b_rec.p, s->n_terms, 1,
                                   (const float*)title->post_w.p, s->c_rec.p, s->c_w.p);
            hipLaunchKernelGGL(k_combined_finish, dim3(std::min<unsigned>(ss::div_up(pc_pad, 256), 16384u)), dim3(256), 0, ctx->stream, s->c_rec.p, pc, s->c_skip.p);
            s->has_combined = true;
            s->c_pad_block = ((pc + 63) & ~(uint64_t)63) / 64;
        }
    }
    if (s->n_terms) {
        hipLaunchKernelGGL(k_kth_impact, dim3((unsigned)s->n_terms), dim3(KH_TPB), 0, ctx->stream, (const uint64_t*)title->term_ptr.p,
                           (const Rec*)s->t_rec.p, s->t_kth.p);
        hipLaunchKernelGGL(k_kth_impact, dim3((unsigned)s->n_terms), dim3(KH_TPB), 0, ctx->stream, (const uint64_t*)body->term_ptr.p,
                           (const Rec*)s->b_rec.p, s->b_kth.p);
    }
    SS_HIP(ctx, hipGetLastError());
    uint32_t h_flags = 0;
    SS_HIP(ctx, ss::fetch(ctx, ctx->stream, &h_flags, flags.p, sizeof(uint32_t)));
    s->clean = h_flags == 0;
    if (ctx->opt("score.exact_all", 0) != 0) s->clean = false;   // tests: force the filter off
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
    for (hipStream_t ws : ctx->wave_stream)
        if (ws) (void)hipStreamSynchronize(ws);
#if defined(SSW_PHASES) && !defined(SS_DIAG)
    ss::score_wave_diag_dump();
#endif
    ss::score_small_report();               // (prints only in the -DSSS_PHASES build of score_small.hip)
#ifdef SS_DIAG
    ss::score_wave_diag_dump();
    {
        unsigned long long h[24];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_diag), sizeof(h)) == hipSuccess) {
            const char* names[19] = {"slices", "windows", "flushes", "flushed_records", "compactions", "records", "cyc_setup_lists", "cyc_setup_bounds", "cyc_flush", "cyc_setup", "cyc_windows", "cyc_total", "cyc_win_add", "cyc_win_barrier", "cyc_merge_explain", "cyc_merge_gather", "cyc_merge_final", "merges", "merge_candidates"};
            fprintf(stderr, "[ss diag] k_score_slices (thread 0 of every slice):");
            for (int i = 0; i < 19; i++) fprintf(stderr, " %s=%llu", names[i], h[i]);
            fprintf(stderr, "\n");
        }
        static unsigned long long hs[4096][4];
        if (hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_slice), sizeof(hs)) == hipSuccess) {
            if (FILE* f = fopen("gpurun_out/ss_diag_slices.csv", "w")) {
                fprintf(f, "launch,start,end,windows,records\n");
                for (int i = 0; i < 4096; i++)
                    if (hs[i][1]) fprintf(f, "%d,%llu,%llu,%llu,%llu\n", i, hs[i][0], hs[i][1], hs[i][2], hs[i][3]);
                fclose(f);
            }
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
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));     // (every k_score_wave on the wave stream has a merge behind it on this one)
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
    s->prior_clean = true;
    for (int t = 0; t < k_topics; t++)          // a NaN anywhere shows as max = +inf, min = -inf (k_prior_extrema)
        if (!(s->prior_min[t] >= 0.0) || !std::isfinite(s->prior_max[t])) s->prior_clean = false;
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

// Batches in flight with HOST results: submit runs the batch like a call with device outputs (nothing waits, consecutive batches
// overlap on the device) into the slot's own device buffers; collect waits for that batch alone and copies its rows to the caller.
// The host's plan for batch i+1 and the copy-out of batch i-1 then run under the kernels of batch i.
int32_t ss_score_topk_submit(ss_scorer* s, int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms, const uint32_t* p_ptr,
                             const uint32_t* p_terms, const int32_t* query_len, const double* topic_probs, int32_t k, uint64_t* ticket_out) {
    if (!s) return SS_ERR_INVALID;
    ss_ctx* ctx = s->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!ticket_out) return ctx->fail(SS_ERR_INVALID, "ss_score_topk_submit: ticket_out is NULL");
    *ticket_out = 0;
    if (n_q < 0 || k < 1 || k > SS_MAX_TOPK) return ctx->fail(SS_ERR_INVALID, "ss_score_topk_submit: n_q < 0 or k outside 1 .. %d", SS_MAX_TOPK);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    ss_scorer::AsyncSlot* a = nullptr;
    for (auto& c : s->aslot)
        if (!c.ticket) { a = &c; break; }
    if (!a) return ctx->fail(SS_ERR_STATE, "ss_score_topk_submit: %d batches in flight already (ss_score_topk_collect one first)", (int)ss_scorer::INFLIGHT);
    const size_t rows = (size_t)n_q * (size_t)k;
    if (a->hits.n < rows) SS_HIP(ctx, a->hits.alloc(rows + rows / 4));
    if (a->n_hits.n < (size_t)n_q) SS_HIP(ctx, a->n_hits.alloc((size_t)n_q + 64));
    if (!a->ev) SS_HIP(ctx, hipEventCreateWithFlags(&a->ev, hipEventDisableTiming));
    a->pin_mode = ctx->opt("score.collect_pinned", 0) != 0;
    if (a->pin_mode) {
        const size_t bytes = rows * sizeof(ss_hit) + (size_t)n_q * sizeof(int32_t);
        if (a->pin_cap < bytes) {
            ctx->pin_free(a->pin, a->pin_cap);
            a->pin = ctx->pin_alloc(bytes, &a->pin_cap);
            if (!a->pin) { a->pin_cap = 0; return ctx->fail(SS_ERR_OOM, "ss_score_topk_submit: no pinned host memory for %zu bytes of results", bytes); }
        }
        if (!s->out_stream) SS_HIP(ctx, hipStreamCreateWithFlags(&s->out_stream, hipStreamNonBlocking));
    }
    if (a->pin_n_cap < (size_t)n_q * sizeof(int32_t)) {
        if (a->pin_n) ctx->pin_free(a->pin_n, a->pin_n_cap);
        a->pin_n = ctx->pin_alloc(std::max<size_t>((size_t)n_q * sizeof(int32_t), 4096), &a->pin_n_cap);
        if (!a->pin_n) a->pin_n_cap = 0;                                   // (no pinned memory: collect copies the counts straight out)
    }
    if (n_q) {
        const int32_t rc = score_impl(s, n_q, q_ptr, q_terms, p_ptr, p_terms, query_len, topic_probs, k, a->hits.p, a->n_hits.p);
        if (rc != SS_OK) return rc;
        // Only an event behind the batch's kernels is recorded here; the rows are copied when they are COLLECTED.  [Enqueued at
        // submit, the device-to-host copy waits in the copy engine's in-order queue for this batch's merge and holds up the NEXT
        // batch's plan upload behind it (submit then took 0.41 ms instead of 0.13); done by a kernel on the caller's stream it sat
        // between two merges that the wave kernel stretches (period 0.48 ms instead of 0.345), on a stream of its own it shared a
        // hardware queue with a wave stream (0.59).  At collect time the batch is finished, the copy takes its 85 us and blocks
        // nothing: the host spends 0.13 ms in submit and 0.09 in collect per batch, under the 0.345 ms the device needs.]
        SS_HIP(ctx, hipEventRecord(a->ev, ctx->stream));
    }
    a->n_q = n_q;
    a->k = k;
    a->ticket = s->next_ticket++;
    *ticket_out = a->ticket;
    return SS_OK;
}

int32_t ss_score_topk_collect(ss_scorer* s, uint64_t ticket, ss_hit* hits_out, int32_t* n_hits_out) {
    if (!s) return SS_ERR_INVALID;
    ss_ctx* ctx = s->ctx;
    ss_scorer::AsyncSlot* a = nullptr;
    // everything the unlocked part needs is read HERE, under the context's lock: the options map, the slot's fields and the scorer's
    // stream may be written by a thread that submits or sets an option meanwhile (ctx->opt() is an unlocked map lookup)
    bool trace = false, pin_mode = false;
    int32_t n_q = 0, k = 0;
    hipEvent_t ev = nullptr;
    hipStream_t out_stream = nullptr;
    const ss_hit* d_hits = nullptr;
    const int32_t* d_n = nullptr;
    void* pin = nullptr;
    void* pin_n = nullptr;
    {
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        if (!hits_out || !n_hits_out) return ctx->fail(SS_ERR_INVALID, "ss_score_topk_collect: NULL output");
        for (auto& c : s->aslot)
            if (ticket && c.ticket == ticket && !c.collecting) { a = &c; break; }
        if (!a) return ctx->fail(SS_ERR_INVALID, "ss_score_topk_collect: no batch in flight with ticket %llu", (unsigned long long)ticket);
        SS_HIP(ctx, hipSetDevice(ctx->device));
        a->collecting = true;                   // (a second collect of the same ticket from another thread is refused, not raced)
        trace = ctx->opt("score.trace", 0) != 0;
        pin_mode = a->pin_mode;
        n_q = a->n_q; k = a->k; ev = a->ev; out_stream = s->out_stream;
        d_hits = a->hits.p; d_n = a->n_hits.p; pin = a->pin; pin_n = a->pin_n;
    }
    // (the wait and the copies run outside the context's lock: another thread may submit the next batch meanwhile; the slot itself
    // stays this call's until its ticket is cleared below)
    if (n_q) {
        const auto tw0 = std::chrono::steady_clock::now();
        hipError_t e = hipEventSynchronize(ev);
        const auto tw1 = std::chrono::steady_clock::now();
        const size_t rows = (size_t)n_q * (size_t)k;
        const size_t hb = rows * sizeof(ss_hit), nb = (size_t)n_q * sizeof(int32_t);
        if (pin_mode) {
            // through the slot's pinned block on the copy engine, then a host memcpy
            if (e == hipSuccess) e = hipMemcpyAsync(pin, d_hits, hb, hipMemcpyDeviceToHost, out_stream);
            if (e == hipSuccess) e = hipMemcpyAsync(static_cast<unsigned char*>(pin) + hb, d_n, nb, hipMemcpyDeviceToHost, out_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(out_stream);
            if (e == hipSuccess) {
                std::memcpy(hits_out, pin, hb);
                std::memcpy(n_hits_out, static_cast<unsigned char*>(pin) + hb, nb);
            }
        } else {
            if (e == hipSuccess) e = hipMemcpy(hits_out, d_hits, hb, hipMemcpyDeviceToHost);
            if (pin_n) {                                                   // the counts: device -> the slot's pinned block -> the caller
                if (e == hipSuccess) e = hipMemcpy(pin_n, d_n, nb, hipMemcpyDeviceToHost);
                if (e == hipSuccess) std::memcpy(n_hits_out, pin_n, nb);
            } else if (e == hipSuccess) {
                e = hipMemcpy(n_hits_out, d_n, nb, hipMemcpyDeviceToHost);
            }
        }
        if (trace) fprintf(stderr, "[score trace] collect: waited %.0f us for the batch, copies %.0f us\n", std::chrono::duration<double, std::micro>(tw1 - tw0).count(),
                           std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw1).count());
        if (e != hipSuccess) {
            std::lock_guard<std::recursive_mutex> lk(ctx->mu);
            a->ticket = 0;
            a->collecting = false;
            return ctx->fail(SS_ERR_HIP, "ss_score_topk_collect: %s", hipGetErrorString(e));
        }
    }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    a->ticket = 0;
    a->collecting = false;
    return SS_OK;
}

static int32_t score_impl_inner(ss_scorer* s, int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms, const uint32_t* p_ptr,
                                const uint32_t* p_terms, const int32_t* query_len, const double* topic_probs, int32_t k,
                                ss_hit* hits_out, int32_t* n_hits_out);

// no C++ exception may cross the C ABI: host allocation failures come back as SS_ERR_OOM
static int32_t score_impl(ss_scorer* s, int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms, const uint32_t* p_ptr,
                          const uint32_t* p_terms, const int32_t* query_len, const double* topic_probs, int32_t k,
                          ss_hit* hits_out, int32_t* n_hits_out) {
    if (!s) return SS_ERR_INVALID;
    try {
        return score_impl_inner(s, n_q, q_ptr, q_terms, p_ptr, p_terms, query_len, topic_probs, k, hits_out, n_hits_out);
    } catch (const std::bad_alloc&) {
        return s->ctx->fail(SS_ERR_OOM, "ss_score_topk: host allocation failed");
    } catch (const std::exception& e) {
        return s->ctx->fail(SS_ERR_INVALID, "ss_score_topk: %s", e.what());
    }
}

static int32_t score_impl_inner(ss_scorer* s, int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms, const uint32_t* p_ptr,
                                const uint32_t* p_terms, const int32_t* query_len, const double* topic_probs, int32_t k,
                                ss_hit* hits_out, int32_t* n_hits_out) {
    ss_ctx* ctx = s->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (n_q < 0 || !q_ptr || !hits_out || !n_hits_out) return ctx->fail(SS_ERR_INVALID, "ss_score_topk: NULL argument or n_q < 0");
    if (k < 1) return ctx->fail(SS_ERR_INVALID, "ss_score_topk: k < 1");
    if (k > SS_MAX_TOPK) return ctx->fail(SS_ERR_UNSUPPORTED, "ss_score_topk: k %d > SS_MAX_TOPK %d", k, SS_MAX_TOPK);
    if (topic_probs && s->k_topics == 0) return ctx->fail(SS_ERR_STATE, "ss_score_topk: topic_probs given but no prior set (ss_scorer_set_prior)");
    if (n_q == 0) return SS_OK;

    const bool trace = ctx->opt("score.trace", 0) != 0;
    auto t_now = [] { return std::chrono::steady_clock::now(); };
    auto t_us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::micro>(b - a).count();
    };
    const auto th0 = t_now();
    // ---- host-side plan (the host keeps df per term; queries are tiny) -------------
    std::vector<uint32_t> h_qptr(n_q + 1);
    SS_HIP(ctx, ss::copy_in(ctx->stream, h_qptr.data(), q_ptr, (n_q + 1) * sizeof(uint32_t)));
    const uint32_t n_tok = h_qptr[n_q];
    for (int q = 0; q < n_q; q++)
        if (h_qptr[q + 1] < h_qptr[q]) return ctx->fail(SS_ERR_INVALID, "ss_score_topk: q_ptr not non-decreasing");
    if (n_tok && !q_terms) return ctx->fail(SS_ERR_INVALID, "ss_score_topk: q_terms is NULL");
    std::vector<uint32_t> h_terms(n_tok);
    if (n_tok) SS_HIP(ctx, ss::copy_in(ctx->stream, h_terms.data(), q_terms, n_tok * sizeof(uint32_t)));
    // phrase part (retrieval/phrase.go): tokens of all quoted phrases of a query, concatenated
    std::vector<uint32_t> h_pptr(n_q + 1, 0), h_pterms, h_pdrv(n_q, 0xFFFFFFFFu), h_xoff(n_q + 1, 0), h_pbase(n_q + 1, 0);
    std::vector<uint4> h_parts;                 // k_phrase_match work: {query, pass, first candidate, candidates}
    if (p_ptr) {
        if (!s->title->pos_ptr.p || !s->body->pos_ptr.p)
            return ctx->fail(SS_ERR_STATE, "ss_score_topk_phrase: positional postings not loaded (ss_index_set_positions on both tables)");
        SS_HIP(ctx, ss::copy_in(ctx->stream, h_pptr.data(), p_ptr, (n_q + 1) * sizeof(uint32_t)));
        for (int q = 0; q < n_q; q++)
            if (h_pptr[q + 1] < h_pptr[q]) return ctx->fail(SS_ERR_INVALID, "ss_score_topk_phrase: p_ptr not non-decreasing");
        h_pterms.resize(h_pptr[n_q]);
        if (h_pptr[n_q]) {
            if (!p_terms) return ctx->fail(SS_ERR_INVALID, "ss_score_topk_phrase: p_terms is NULL");
            SS_HIP(ctx, ss::copy_in(ctx->stream, h_pterms.data(), p_terms, h_pterms.size() * sizeof(uint32_t)));
        }
    }
    std::vector<int32_t> h_qlen(n_q);
    if (query_len) SS_HIP(ctx, ss::copy_in(ctx->stream, h_qlen.data(), query_len, n_q * sizeof(int32_t)));
    else for (int q = 0; q < n_q; q++)     // len(queryTokenised)+len(phraseTokenised), main_retrieve.go:90
        h_qlen[q] = (int32_t)(h_qptr[q + 1] - h_qptr[q]) + (int32_t)(h_pptr[q + 1] - h_pptr[q]);
    std::vector<double> h_probs;
    const int K = s->k_topics;
    if (topic_probs) {
        h_probs.resize((size_t)n_q * K);
        SS_HIP(ctx, ss::copy_in(ctx->stream, h_probs.data(), topic_probs, h_probs.size() * sizeof(double)));
    }

    // The filter of k_score_slices assumes non-negative finite addends (see there).  Anything else — tables or priors
    // flagged at creation, a negative / non-finite topic probability, queryLength <= 0 — switches it off for this call:
    // every record then goes through the exact stage, which restates the reference's arithmetic for any input.
    bool exact_all = !s->clean || (topic_probs && !s->prior_clean);
    for (int q = 0; q < n_q && !exact_all; q++) exact_all = h_qlen[q] <= 0;
    for (size_t i = 0; i < h_probs.size() && !exact_all; i++) exact_all = !(h_probs[i] >= 0.0) || !std::isfinite(h_probs[i]);
    int kth_j = 0;
    while ((1 << kth_j) < k) kth_j++;
    const auto th1 = t_now();

    const std::vector<uint64_t>& tp = s->title->h_term_ptr;
    const std::vector<uint64_t>& bp = s->body->h_term_ptr;
    bool any_phrase = false;
    for (int q = 0; q < n_q && p_ptr; q++) {
        const uint32_t m = h_pptr[q + 1] - h_pptr[q];
        h_xoff[q + 1] = h_xoff[q];
        h_pbase[q + 1] = (uint32_t)h_parts.size();
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
        h_pbase[q + 1] = (uint32_t)h_parts.size();
        if (!known) { h_pdrv[q] = 0xFFFFFFFFu; continue; }
        if ((uint64_t)h_xoff[q] + best >= (1ull << 32))
            return ctx->fail(SS_ERR_UNSUPPORTED, "ss_score_topk_phrase: the phrases of this batch have more than 2^32 candidate documents (split the batch)");
        h_xoff[q + 1] = h_xoff[q] + (uint32_t)best;   // matches <= docs of the rarest term
        // candidates = the rarest term's body postings (pass 0), then its title postings (pass 1), PH_PART per workgroup
        const uint32_t dt = h_pterms[h_pptr[q] + h_pdrv[q]];
        const uint64_t nbody = bp[dt + 1] - bp[dt], ntitle = tp[dt + 1] - tp[dt];
        for (int pass = 0; pass < 2; pass++) {
            const uint64_t nc = pass == 0 ? nbody : ntitle;
            for (uint64_t c = 0; c < nc; c += PH_PART)
                h_parts.push_back(make_uint4((uint32_t)q, (uint32_t)pass, (uint32_t)c, (uint32_t)std::min<uint64_t>(PH_PART, nc - c)));
        }
        h_pbase[q + 1] = (uint32_t)h_parts.size();
    }
    // results straight into the caller's buffers when both live in device memory (then the call does not wait either)
    bool dev_out = false;
    {
        hipPointerAttribute_t a1{}, a2{};
        const bool d1 = hipPointerGetAttributes(&a1, hits_out) == hipSuccess && a1.type == hipMemoryTypeDevice;
        const bool d2 = hipPointerGetAttributes(&a2, n_hits_out) == hipSuccess && a2.type == hipMemoryTypeDevice;
        (void)hipGetLastError();                 // plain host memory is reported as an error: not one
        dev_out = d1 && d2;
    }
    // Slice size: SLICE_TARGET postings when the batch fills the chip several times over; smaller (down to
    // SLICE_MIN) for small batches, so that one query's lists are spread over many CUs instead of being
    // walked by a single workgroup (latency of a lone query: 0.72 ms -> see DESIGN.md K4).
    uint64_t slice_target = SLICE_TARGET;
    uint64_t grade_tot = 0, grade_seen = 0;
    const bool grade_slices = ctx->opt("score.grade_slices", 0) != 0;
    // The tokens' list lengths, looked up ONCE: the two host copies of term_ptr are 8 MB each at config 3 and a batch's terms are
    // scattered over them, so every look-up is a cache miss — and the plan used to make them in three passes (batch total, the
    // wave kernel's suitability test, the per-query slicing): 37k misses per 1024-query batch, most of the 43 us the plan took for a
    // batch of tail queries (round 5: `score.trace`, tools/host_tail.py).  0 / 0 for unknown terms.
    static thread_local std::vector<uint32_t> tok_lt, tok_lb;
    tok_lt.resize(n_tok);
    tok_lb.resize(n_tok);
    uint64_t tok_tot = 0;
    for (uint32_t i = 0; i < n_tok; i++) {
        const uint32_t t = h_terms[i];
        const bool known = (uint64_t)t < s->n_terms;
        tok_lt[i] = known ? (uint32_t)(tp[t + 1] - tp[t]) : 0u;       // (a posting list is shorter than 2^32: ss_index_create)
        tok_lb[i] = known ? (uint32_t)(bp[t + 1] - bp[t]) : 0u;
        tok_tot += (uint64_t)tok_lt[i] + tok_lb[i];
    }
    {
        const uint64_t batch_tot = tok_tot;
        const uint64_t slots = (uint64_t)std::max(ctx->cu_count, 1) * SS_WGS_PER_CU;
        // measured (10M docs, 3-term head queries, batches of 1..4096): one partial wave of slices is best — about
        // 1.5x the batch's postings per resident workgroup slot, never below SLICE_MIN (a slice costs ~45 us of
        // threshold warm-up whatever its size) nor above SLICE_TARGET
        slice_target = std::min<uint64_t>(SLICE_TARGET, std::max<uint64_t>(SLICE_MIN, batch_tot * 3 / (2 * slots)));
        // (results in device memory: consecutive batches overlap — "score.pipeline_slices" —, the next batch's kernel fills this one's
        //  tail and larger slices pay: mixed batch 0.164 ms at 2.6x this target against 0.170)
        if (dev_out && ctx->opt("score.pipeline", 2) != 0 && ctx->opt("score.pipeline_slices", 1) != 0)
            slice_target = std::min<uint64_t>(SLICE_TARGET, slice_target * 5 / 2);
        slice_target = (uint64_t)std::max<int64_t>(1024, ctx->opt("score.slice_target", (int64_t)slice_target));   // experiments only
        grade_tot = batch_tot;
    }
    // k_score_wave (one wave per slice) takes the plain OR queries: few lists, no phrase part, small k, inputs for which the
    // filter's assumptions hold, and a list long enough for the threshold floor (k'-th largest impact, k' >= k) to exist;
    // everything else runs k_score_slices.  Option "score.wave" = 0 switches the wave kernel off (tests, A/B).
    const bool wave_ok = ctx->opt("score.wave", 1) != 0 && s->has_combined && !exact_all && k <= ss::score_wave_max_k();
    uint64_t wave_target = 0;
    // (85 / 115 / 40 suited one batch at a time; with consecutive batches overlapping the next batch's kernel fills this one's tail
    //  and fewer, larger tail slices pay: 0.334-0.338 ms per batch at config 3 against 0.341-0.345, `tools/score_wall.py` with OPTS)
    const int64_t grade_pct = ctx->opt("score.wave_big_pct", 92), grade_big = ctx->opt("score.wave_big_x100", 115),
                  grade_small = ctx->opt("score.wave_small_x100", 60);
    if (wave_ok) {
        // about 5.5 slices per nine-wave-per-CU slot (four rounds of the 12 waves a CU holds), 8k .. 48k postings each
        const uint64_t slots = (uint64_t)std::max(ctx->cu_count, 1) * 9;
        const uint64_t batch_tot = tok_tot;
        wave_target = std::min<uint64_t>(49152, std::max<uint64_t>(8192, batch_tot * 2 / (11 * slots)));          // (config 3, ms per batch at 6k / 8k / 10k / 12k / 14k / 17k / 21k postings: 0.661 / 0.635 / 0.616 / 0.623 / 0.655 / 0.649 / 0.639)
        wave_target = (uint64_t)std::max<int64_t>(1024, ctx->opt("score.wave_slice_target", (int64_t)wave_target));
    }
    // Which queries suit k_score_wave: no phrase part, few lists, a list long enough for the threshold floor (k'-th largest
    // impact, k' = k rounded up to 2^j) to exist — and EVERY list long enough for that floor to be selective: the k'-th largest of
    // n impacts lets k'/n of a list's records through until the real threshold has risen (measured: batches of term ranks
    // U[1,100k] — long and short lists mixed — ran 3.5x slower here than in k_score_slices, with 100x the overflow events per
    // slice).  Option "score.wave_min_list" x k' postings (default 16; 0 = no such demand: tests reach the kernel with small tables).
    // The kernel is taken per BATCH: two scoring kernels one after the other each pay their ramp-up and tail (a batch split
    // between them measured slower than either alone), so it runs only when the queries it suits carry 90 % of the batch.
    // ... and few lists: a window is cut so that its blocks number 16 - 3 - (lists), every list adding a boundary block; with
    // 12 dense lists a window is one driver block and most windows overflow into the slow path (soak on the config-3 index,
    // wave / slices time: 0.4-0.9 at <= 5 terms, 1.0-1.1 at 9, 2-3 at 12).  Option "score.wave_max_terms", default 6.
    const uint32_t wave_max_terms = (uint32_t)std::min<int64_t>(ss::score_wave_max_lists(), std::max<int64_t>(1, ctx->opt("score.wave_max_terms", 6)));
    const int64_t wml = std::max<int64_t>(0, ctx->opt("score.wave_min_list", 16));
    const uint64_t wave_min_list = (uint64_t)wml << kth_j;
    std::vector<uint8_t> h_suits(n_q, 0);
    bool batch_wave = false;
    if (wave_ok) {
        uint64_t fit = 0, all = 0;
        for (int q = 0; q < n_q; q++) {
            uint64_t tot = 0, shortest = ~0ull, longest = 0;
            uint32_t n_known = 0;
            for (uint32_t i = h_qptr[q]; i < h_qptr[q + 1]; i++) {
                const uint32_t t = h_terms[i];
                if ((uint64_t)t >= s->n_terms) continue;
                n_known++;
                tot += (uint64_t)tok_lt[i] + tok_lb[i];
                const uint64_t len = std::max(tok_lt[i], tok_lb[i]);
                shortest = std::min(shortest, len);
                longest = std::max(longest, len);
            }
            all += tot;
            const bool phrase_q = p_ptr && h_pptr[q + 1] > h_pptr[q];
            // (n_known counts duplicate tokens too: an upper bound of the distinct terms, good enough for the choice)
            if (n_known && n_known <= wave_max_terms && !phrase_q && longest >= (uint64_t)4 * (uint64_t)k && longest >= 1024 &&
                shortest >= wave_min_list) {
                h_suits[q] = 1;
                fit += tot;
            }
        }
        // (a lone query or two — under ~400k postings — finish sooner in k_score_slices: 0.143 against 0.166 ms for two head queries,
        //  host in / host out; from four queries on the wave kernel leads, 0.21 against 0.31 ms)
        // "score.wave_share_pct": the share of the batch's postings the suited queries must carry (default 90; with device outputs
        // the two kernels of a split batch run side by side on two streams, so a split no longer pays two ramp-ups and tails one
        // after the other)
        const uint64_t share = (uint64_t)std::max<int64_t>(0, std::min<int64_t>(100, ctx->opt("score.wave_share_pct", 90)));
        batch_wave = wml == 0 ? fit > 0 : (fit * 100 >= all * share && fit >= 400000);
    }
    std::vector<uint8_t> h_fast(n_q, 0);
    // k_score_small (one workgroup per query, every posting scored exactly, no slices and no merge: score_small.hip) takes queries without
    // a phrase part whose lists hold at most score_small_cap() postings in all.  Bit-identical hits.  Round 5 measured both ways: a
    // workgroup is a chain of short phases (list table, postings -> hash table, magnitudes -> scores, radix selection, placement, hits:
    // ~20 us with nothing to overlap), so a 1024-query tail batch is SLOWER there than in the slices pipeline that runs batches side by
    // side (0.14 against 0.09 ms), while a short call — where latency is all there is — is faster (1 .. 32 tail queries host to host
    // 0.068-0.074 -> 0.050-0.058 ms): DESIGN K4c.
    // "score.small": 1 = every query that fits (tests, A/B); 2 (default) = only a call that consists of such queries and is at most
    // "score.small_max_batch" queries long: ONE launch that writes the hits, against a slices kernel and a merge — what a lone query or
    // a handful gain in latency a 1024-query batch loses in overlap (the slices pipeline runs batches side by side); 0 = never.
    const int64_t small_mode = ctx->opt("score.small", 2);
    const uint64_t small_cap = (uint64_t)std::min<int64_t>(ss::score_small_cap(), std::max<int64_t>(0, ctx->opt("score.small_cap", ss::score_small_cap())));
    bool small_ok = small_mode == 1 && k <= ss::score_small_max_k();
    // "score.small_batch" = 1 (with "score.small" = 2): in a LONGER call with device outputs every query that fits goes there too, the
    // kernel on an internal stream beside the neighbouring batches like the slices kernel, its rows staged and copied on the caller's stream
    const bool small_batch = small_mode == 2 && dev_out && k <= ss::score_small_max_k() && ctx->opt("score.pipeline", 2) != 0 &&
                             ctx->opt("score.small_batch", 0) != 0 && n_q > ctx->opt("score.small_max_batch", 64);
    if (small_batch) small_ok = true;
    if (small_mode == 2 && k <= ss::score_small_max_k() && n_q <= ctx->opt("score.small_max_batch", 64) && !p_ptr) {
        small_ok = true;                                    // (tokens counted with their repeats: an upper bound of a query's postings)
        for (int q = 0; q < n_q && small_ok; q++) {
            uint64_t tq = 0;
            for (uint32_t i = h_qptr[q]; i < h_qptr[q + 1]; i++) tq += (uint64_t)tok_lt[i] + tok_lb[i];
            small_ok = tq <= small_cap && h_qptr[q + 1] - h_qptr[q] <= (uint32_t)(ss::score_small_max_lists() / 2);
        }
    }
    struct SmallEnt { SmallHdr h; uint32_t loff; };
    std::vector<SmallEnt> h_small_a, h_small_b;         // by table size: up to score_small_cap_a() postings, and beyond
    std::vector<SmallList> h_small_lists;
    uint32_t small_lmax = 0;
    std::vector<uint32_t> h_qoff(n_q + 1, 0), h_dterm, h_dmult, h_sbase(n_q + 1, 0);
    std::vector<double> h_qmag(n_q), h_ub(n_q, 0.0);
    std::vector<SliceDesc> h_slices;
    std::vector<uint64_t> h_qcost(n_q, 0);          // postings per slice of the query (all its slices cost the same)
    h_slices.reserve((size_t)n_q * 16);
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
            tot += (uint64_t)tok_lt[i] + tok_lb[i];
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
        // the window plan holds (n_win + 1) * L cursors: keep a slice within what the plan can cut into regular windows
        const uint64_t n_lists = 2 * (h_dterm.size() - d0) + 4;
        const uint64_t plan_cap = std::max<uint64_t>(TARGET, (uint64_t)(TBL_CAP / n_lists > 2 ? TBL_CAP / n_lists - 2 : 1) * TARGET * 7 / 8);
        uint64_t q_target = slice_target;
        uint64_t max_slices = MAX_SLICES_PER_Q;
        if (small_ok && tot <= small_cap && !(p_ptr && h_pptr[q + 1] > h_pptr[q])) {
            // no slices: k_score_small reads the lists whole and writes the hits.  Its list table is made here — the host holds
            // term_ptr, and the kernel's own walk (query -> terms -> term_ptr) was three dependent loads at the head of every workgroup
            SmallList tmp[SS_SMALL_MAX_LISTS];
            uint32_t nl = 0, run = 0;
            bool fits = true;
            for (size_t j = d0; j < h_dterm.size() && fits; j++) {
                const uint32_t t = h_dterm[j];
                for (uint32_t field = 0; field < 2; field++) {
                    const uint64_t* pp = field ? tp.data() : bp.data();
                    const uint64_t b = pp[t], e = pp[t + 1];
                    if (e == b) continue;
                    if (nl == SS_SMALL_MAX_LISTS) { fits = false; break; }
                    run += (uint32_t)(e - b);
                    tmp[nl++] = SmallList{b, run, h_dmult[j] << 1 | field};
                }
            }
            if (fits) {
                SmallEnt en;
                en.h = SmallHdr{(uint32_t)q, nl, run, 0u, h_qmag[q], 0.0};
                en.loff = (uint32_t)h_small_lists.size();
                h_small_lists.insert(h_small_lists.end(), tmp, tmp + nl);
                small_lmax = std::max(small_lmax, nl);
                (run <= ss::score_small_cap_a() ? h_small_a : h_small_b).push_back(en);
                h_fast[q] = 2;
                h_sbase[q + 1] = (uint32_t)h_slices.size();
                continue;
            }
        }
        const bool fast = batch_wave && h_suits[q] && h_dterm.size() > d0 && (h_dterm.size() - d0) <= (size_t)ss::score_wave_max_lists();
        if (fast) {
            h_fast[q] = 1;
            q_target = wave_target;
            max_slices = 4096;
        }
        // graded slices: the first part of the batch's postings in larger slices, the rest in smaller ones — launched
        // longest first, the small ones fill the kernel's tail (both kernels; see DESIGN K4b)
        if (grade_pct > 0 && (fast || grade_slices)) {
            q_target = grade_seen * 100 < grade_tot * (uint64_t)grade_pct ? q_target * (uint64_t)grade_big / 100 : q_target * (uint64_t)grade_small / 100;
            q_target = std::max<uint64_t>(q_target, 1024);
            grade_seen += tot;
        }
        if (!fast) q_target = std::min<uint64_t>(q_target, plan_cap);
        uint64_t ns = std::max<uint64_t>(1, (tot + q_target - 1) / q_target);
        ns = std::min<uint64_t>(ns, std::min<uint64_t>(max_slices, s->n_docs));
        h_qcost[q] = tot / ns;
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
    if (small_mode == 2 && !(small_batch && ctx->opt("score.small_batch", 0) == 2) && !h_small_a.empty() && !h_small_b.empty()) {      // one launch: the larger table takes them all
        h_small_b.insert(h_small_b.end(), h_small_a.begin(), h_small_a.end());
        h_small_a.clear();
    }
    const size_t n_small_a = h_small_a.size(), n_small_b = h_small_b.size(), n_small = n_small_a + n_small_b;
    const size_t small_stride = sizeof(SmallHdr) + (size_t)small_lmax * sizeof(SmallList);
    const size_t n_slices = h_slices.size();
    const size_t n_d = h_dterm.size();
    // launch order: the wave kernel's slices first, then k_score_slices' (each group longest first); merge list = the wave queries
    // (the slices of a query are consecutive and cost the same, so the order is that of the QUERIES, stably sorted, with every
    //  query's slices in a row: sorting 14.6k slice indices took two thirds of the 0.22 ms a config-3 batch is planned in)
    std::vector<uint32_t> h_order(n_slices), h_mergeq;
    {
        // (a counting sort by a coarse cost class — the leading bit of the cost and the four bits behind it, 6 % steps — instead of
        //  std::stable_sort by the exact cost: the order only decides which slices are LAUNCHED first, and the comparison sort with its
        //  temporary buffer was most of the 35 us the plan of a 1024-query batch took (round 5); stable inside a class)
        std::vector<uint32_t> q_order(n_q);
        {
            constexpr int NCLS = 2048;
            auto cls_of = [&](uint32_t q) -> int {
                const uint64_t c = h_qcost[q];
                int kc = 0;
                if (c) {
                    const int msb = 63 - __builtin_clzll(c);
                    const uint64_t frac = msb >= 4 ? (c >> (msb - 4)) & 15u : (c << (4 - msb)) & 15u;
                    kc = (msb << 4 | (int)frac) + 1;                    // <= 64 * 16
                }
                return ((h_fast[q] & 1) ? 1024 : 0) + std::min(kc, 1023);
            };
            uint32_t cnt[NCLS + 1] = {};
            for (int q = 0; q < n_q; q++) cnt[NCLS - 1 - cls_of((uint32_t)q)]++;      // descending classes
            uint32_t run = 0;
            for (int c = 0; c < NCLS; c++) { const uint32_t v = cnt[c]; cnt[c] = run; run += v; }
            for (int q = 0; q < n_q; q++) q_order[cnt[NCLS - 1 - cls_of((uint32_t)q)]++] = (uint32_t)q;
        }
        size_t o = 0;
        for (int i = 0; i < n_q; i++)
            for (uint32_t sl = h_sbase[q_order[i]]; sl < h_sbase[q_order[i] + 1]; sl++) h_order[o++] = sl;
    }
    size_t n_fast_slices = 0;
    for (size_t i = 0; i < n_slices; i++) n_fast_slices += h_fast[h_slices[i].q] & 1;
    for (int q = 0; q < n_q; q++)
        if (h_fast[q] & 1) h_mergeq.push_back((uint32_t)q);

    int cb = SS_CB_MIN;
    while (cb < 2 * k) cb <<= 1;
    // k_merge_flat gathers ALL of a query's candidates before it sorts once (config 3: ~300 per query at k = 100; with room for
    // 2k only it sorted 2.6 times per query, and the sorts' barriers were half of the merge)
    const int cb_flat = std::max(cb, 512);
    const auto th2 = t_now();

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
    const size_t o_pbase = o;  o = align16(o + (n_q + 1) * sizeof(uint32_t));
    const size_t o_parts = o;  o = align16(o + h_parts.size() * sizeof(uint4));
    const size_t o_probs = o;  o = align16(o + h_probs.size() * sizeof(double));
    const size_t o_mergeq = o; o = align16(o + h_mergeq.size() * sizeof(uint32_t));
    const size_t o_qfast = o;  o = align16(o + (size_t)n_q);
    const size_t o_smalltab = o; o = align16(o + n_small * small_stride);
#ifdef SS_EXP_FLOOR
    const bool use_floor = ctx->opt("score.debug_floor", 0) != 0 && s->dbg_floor.size() == (size_t)n_q;
#else
    const bool use_floor = false;
#endif
    const size_t o_qfloor = o; o = align16(o + (use_floor ? (size_t)n_q * sizeof(float) : 0));
    const size_t plan_bytes = o;
    const int pb = s->plan_turn;
    s->plan_turn = (s->plan_turn + 1) % ss_scorer::TURNS;
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
    if (!s->batch_ev[pb]) SS_HIP(ctx, hipEventCreateWithFlags(&s->batch_ev[pb], hipEventDisableTiming));
    // the batch two calls ago used this turn's device buffers (plan, prep, candidates) and may still be running — in pipelined mode its
    // merge certainly may: the host runs at most two batches ahead, and waits here BEFORE any of those buffers is grown or rewritten
    if (s->batch_ev_pending[pb]) {
        SS_HIP(ctx, hipEventSynchronize(s->batch_ev[pb]));
        s->batch_ev_pending[pb] = false;
    }
    SS_HIP(ctx, ensure(s->d_plan2[pb], plan_bytes));
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
    std::memcpy(hp + o_pbase, h_pbase.data(), (n_q + 1) * sizeof(uint32_t));
    if (!h_parts.empty()) std::memcpy(hp + o_parts, h_parts.data(), h_parts.size() * sizeof(uint4));
    if (!h_probs.empty()) std::memcpy(hp + o_probs, h_probs.data(), h_probs.size() * sizeof(double));
    if (!h_mergeq.empty()) std::memcpy(hp + o_mergeq, h_mergeq.data(), h_mergeq.size() * sizeof(uint32_t));
    std::memcpy(hp + o_qfast, h_fast.data(), (size_t)n_q);
    if (use_floor) std::memcpy(hp + o_qfloor, s->dbg_floor.data(), (size_t)n_q * sizeof(float));
    {
        unsigned char* w = hp + o_smalltab;              // the 1024-slot queries first, then the larger ones (launch_score_small)
        for (const std::vector<SmallEnt>* v : {&h_small_a, &h_small_b})
            for (const SmallEnt& en : *v) {
                std::memcpy(w, &en.h, sizeof(SmallHdr));
                std::memcpy(w + sizeof(SmallHdr), h_small_lists.data() + en.loff, (size_t)en.h.n_lists * sizeof(SmallList));
                w += small_stride;
            }
    }
    const auto th3 = t_now();
    if (any_phrase) {
        for (int x = 0; x < 4; x++) {
            SS_HIP(ctx, ensure(s->d_x[pb][x], (size_t)h_xoff[n_q]));
            SS_HIP(ctx, ensure(s->d_xw[pb][x], (size_t)h_xoff[n_q]));
        }
        SS_HIP(ctx, ensure(s->d_xcnt[pb], (size_t)n_q * 4));
        SS_HIP(ctx, ensure(s->d_pcnt[pb], std::max<size_t>(h_parts.size(), 1) * 2));
    }
    SS_HIP(ctx, ensure(s->d_so_key2[pb], n_slices * k));
    SS_HIP(ctx, ensure(s->d_so_doc2[pb], n_slices * k));
    SS_HIP(ctx, ensure(s->d_so_cnt2[pb], n_slices));
    // A batch that is all k_score_slices, results in device memory ("score.pipeline" != 0 and "score.pipeline_slices", default on):
    // pipelined like the wave batches — the slices kernel on an internal stream, the merge (k_merge_topk, the kernel that writes
    // the hits) as a launch of its own on the caller's stream behind an event.  Small slices leave the last third of their kernel
    // on a thinning machine; the next batch's kernel now starts under it.  (The fused merge — the last slice of a query merges it
    // inside k_score_slices — cannot move off the caller's stream: it writes the hits, and a consumer the caller enqueued between
    // two calls must see the first call's hits before the second call's kernel touches the buffer.)
    const bool pipe_s = dev_out && n_fast_slices == 0 && n_slices > 0 && ctx->opt("score.pipeline", 2) != 0 &&
                        ctx->opt("score.pipeline_slices", 1) != 0;       // (phrase queries included: their match kernels go in front of the slices kernel)
    // ... and the k_score_slices part of a SPLIT batch (some queries on the wave kernel, the rest here): on a second internal stream
    // beside the wave kernel, unfused, its k_merge_topk on the caller's stream in front of the wave queries' k_merge_flat
    const bool pipe_split = dev_out && n_fast_slices > 0 && n_slices > n_fast_slices && !h_mergeq.empty() && ctx->opt("score.pipeline", 2) >= 2 &&
                            ctx->opt("score.pipeline_slices", 1) != 0;
    const bool fused = ctx->opt("score.separate_merge", 0) == 0 && !pipe_s && !pipe_split;
    // k_score_small of a pipelined batch: on the slices kernel's stream if there is one, else on a side stream beside the wave kernel,
    // else (every query small) on a wave stream of its own
    const bool small_staged = small_batch && n_small > 0;
    const bool small_side = small_staged && !pipe_split && dev_out && n_fast_slices > 0 && !h_mergeq.empty() && ctx->opt("score.pipeline", 2) >= 2;
    const bool small_alone = small_staged && !pipe_s && !pipe_split && !small_side;
    if (small_staged) {
        SS_HIP(ctx, ensure(s->d_small_stage[pb], n_small * (size_t)k));
        SS_HIP(ctx, ensure(s->d_small_stage_n[pb], n_small));
    }
    if (fused && s->qticket_zeroed < (size_t)n_q) {
        SS_HIP(ctx, ensure(s->d_qticket, (size_t)n_q));
        SS_HIP(ctx, hipMemsetAsync(s->d_qticket.p, 0, (size_t)n_q * sizeof(uint32_t), st));
        s->qticket_zeroed = (size_t)n_q;
    }
    if (!h_mergeq.empty() && s->qcnt_zeroed2[pb] < (size_t)n_q) {    // k_merge_flat hands every counter back at zero
        SS_HIP(ctx, ensure(s->d_qcnt2[pb], (size_t)n_q));
        SS_HIP(ctx, hipMemsetAsync(s->d_qcnt2[pb].p, 0, s->d_qcnt2[pb].bytes(), st));
        SS_HIP(ctx, hipStreamSynchronize(st));                        // (first use or growth only) k_score_wave may run on another stream
        s->qcnt_zeroed2[pb] = s->d_qcnt2[pb].n;
    }
    // Small results that go back to the host (a lone query, a handful): hits and counts in ONE device block, one copy into the context's
    // pinned scratch, two host memcpys — a second device-to-host copy costs a lone query ~8 us of its ~0.12 ms (round 5)
    const size_t res_rows = (size_t)n_q * k;
    const size_t res_bytes = res_rows * sizeof(ss_hit) + (size_t)n_q * sizeof(int32_t);
    // (up to 128 KB — 32 queries at k = 100 —: beyond that the extra host copy costs more than the second transfer; a 4 MB batch through
    //  a pinned block measured 0.79 against 0.655 ms in round 4)
    bool one_copy = !dev_out && res_bytes <= ss_scorer::H_RES_BYTES;
    if (one_copy && !s->h_res && hipHostMalloc(reinterpret_cast<void**>(&s->h_res), ss_scorer::H_RES_BYTES, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        s->h_res = nullptr;
        one_copy = false;
    }
    SS_HIP(ctx, ensure(s->d_hits, res_rows + (one_copy ? ((size_t)n_q * sizeof(int32_t) + sizeof(ss_hit) - 1) / sizeof(ss_hit) : 0)));
    SS_HIP(ctx, ensure(s->d_nhits, n_q));

    const unsigned char* dp = s->d_plan2[pb].p;
    ScoreParams p{};
    p.t_ptr = s->title->term_ptr.p; p.t_rec = s->t_rec.p; p.t_w = s->title->post_w.p; p.t_mag = s->title->mag.p; p.t_kth = s->t_kth.p;
    p.b_ptr = s->body->term_ptr.p; p.b_rec = s->b_rec.p; p.b_w = s->body->post_w.p; p.b_mag = s->body->mag.p; p.b_kth = s->b_kth.p;
    p.c_ptr = s->c_ptr.p; p.c_rec = s->c_rec.p; p.c_w = s->c_w.p; p.c_skip = s->c_skip.p;
    p.c_pad_block = (uint32_t)s->c_pad_block;
    p.t_pos_ptr = s->title->pos_ptr.p; p.t_pos = s->title->pos.p;
    p.b_pos_ptr = s->body->pos_ptr.p; p.b_pos = s->body->pos.p;
    if (any_phrase) {
        p.ph_off = reinterpret_cast<const uint32_t*>(dp + o_pptr);
        p.ph_terms = reinterpret_cast<const uint32_t*>(dp + o_pterms);
        p.ph_drv = reinterpret_cast<const uint32_t*>(dp + o_pdrv);
        p.x_off = reinterpret_cast<const uint32_t*>(dp + o_xoff);
        for (int x = 0; x < 4; x++) { p.x_rec[x] = s->d_x[pb][x].p; p.x_w[x] = s->d_xw[pb][x].p; }
        p.x_cnt = s->d_xcnt[pb].p;
        p.ph_parts = reinterpret_cast<const uint4*>(dp + o_parts);
        p.ph_pbase = reinterpret_cast<const uint32_t*>(dp + o_pbase);
        p.ph_pcnt = s->d_pcnt[pb].p;
    }
    p.prior = K ? s->prior.p : nullptr;
    p.k_topics = K;
    p.q_off = reinterpret_cast<const uint32_t*>(dp + o_qoff);
    p.dterm = reinterpret_cast<const uint32_t*>(dp + o_dterm);
    p.dmult = reinterpret_cast<const uint32_t*>(dp + o_dmult);
    p.qmag = reinterpret_cast<const double*>(dp + o_qmag);
    p.probs = topic_probs ? reinterpret_cast<const double*>(dp + o_probs) : nullptr;
    p.sqd_ub = reinterpret_cast<const double*>(dp + o_ub);
    p.slice_base = reinterpret_cast<const uint32_t*>(dp + o_sbase);
    p.slices = reinterpret_cast<const SliceDesc*>(dp + o_slices);
    p.order = reinterpret_cast<const uint32_t*>(dp + o_order);
    p.k = k;
    p.cb = cb;
    p.cb_flat = cb_flat;
    p.kth_j = kth_j;
    p.exact_all = exact_all ? 1 : 0;
    p.so_key = s->d_so_key2[pb].p; p.so_doc = s->d_so_doc2[pb].p; p.so_cnt = s->d_so_cnt2[pb].p;
    p.q_ticket = fused ? s->d_qticket.p : nullptr;
    p.qc_cnt = s->d_qcnt2[pb].p;
    p.merge_q = reinterpret_cast<const uint32_t*>(dp + o_mergeq);
    p.q_fast = reinterpret_cast<const uint8_t*>(dp + o_qfast);
    p.small_q = nullptr;
    p.small_tab = dp + o_smalltab;
    p.small_stride = (uint32_t)small_stride;
    p.q_floor = use_floor ? reinterpret_cast<const float*>(dp + o_qfloor) : nullptr;
    p.small_stage = small_staged ? s->d_small_stage[pb].p : nullptr;
    p.small_stage_n = small_staged ? s->d_small_stage_n[pb].p : nullptr;
    p.hits = dev_out ? hits_out : s->d_hits.p;
    p.n_hits = dev_out ? n_hits_out : one_copy ? reinterpret_cast<int32_t*>(s->d_hits.p + res_rows) : s->d_nhits.p;

    // "score.pipeline" (default): a batch that is all k_score_wave, results in device memory.  Its k_wave_prep and k_score_wave go
    // to the context's WAVE stream, its k_merge_flat to the caller's stream behind an event: the next batch's k_score_wave (which
    // needs nothing the caller's stream produces — queries arrive in host memory, the index is immutable while a scorer holds it,
    // candidate lists and counters exist once per turn) starts under this batch's merge, and since the merge — the kernel that
    // writes the hits — sits on the caller's stream, the results are complete in stream order like before.  [Round 3 had it the
    // other way round (merge on a side stream that the caller's stream did not wait for): faster by the same amount, but the hits
    // were only complete after ss_synchronize.]
    // A batch split between the two scoring kernels (queries that do not suit k_score_wave: short lists, phrases, many terms) is
    // pipelined too: its k_score_slices part — which writes its queries' hits itself — runs on the caller's stream BESIDE the wave
    // kernel, the wave queries' merge follows behind both.
    const bool pipe = dev_out && n_fast_slices > 0 && !h_mergeq.empty() && ctx->opt("score.pipeline", 2) != 0;
    // The upload goes out on the context's SECOND stream as soon as the plan is staged — beside the kernels of the previous
    // batch, which read the other device buffer.  (On the one stream the copy sat
    // between two batches: 39 us per batch in the kernel trace with the counter memset, 6 % of the wall time at config 3.)
    if (n_fast_slices) SS_HIP(ctx, ensure(s->d_wprep2[pb], ss::score_wave_prep_bytes((unsigned)n_fast_slices)));
    bool prep_done = false;
    if (!dev_out) {
        // results go back to the host: the call waits for them anyway, and a second wait in the middle would only add to a lone
        // query's latency (0.15 ms, of which 0.08 are kernels): copy, kernels and read-back follow each other on the one stream
        SS_HIP(ctx, hipMemcpyAsync(s->d_plan2[pb].p, hp, plan_bytes, hipMemcpyHostToDevice, st));
    } else {
        SS_HIP(ctx, hipMemcpyAsync(s->d_plan2[pb].p, hp, plan_bytes, hipMemcpyHostToDevice, ctx->comm_stream));
        // k_wave_prep reads the plan and the index, nothing of an earlier batch.  Unpipelined it follows the copy on the second
        // stream and so runs beside the previous batch's kernels (12 us of kernel and one launch gap per batch off the caller's
        // stream); pipelined it is enqueued on the wave stream in front of its k_score_wave (it finds no room beside the previous
        // batch's k_score_wave anyway: 3 x 168 VGPRs per SIMD).
        if (n_fast_slices && !pipe) {
            ss::launch_wave_prep(&p, (unsigned)n_fast_slices, s->d_wprep2[pb].p, ctx->comm_stream);
            prep_done = true;
        }
        // ... and the HOST waits for the second stream (~30 us; it has 0.4 ms to spare per batch): the kernels then go out with no
        // cross-stream dependency in front of them (a hipStreamWaitEvent there left 21 us between two batches, and two of
        // them per batch ran the runtime out of signals every ~80 batches: an 8 ms stall)
        SS_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));
    }
    const auto th4 = t_now();
    const size_t lds_score = score_lds_bytes(cb), lds_merge = merge_lds_bytes(k, cb);
    if (s->lds_attr < cb) {
        SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_score_slices), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_score));
        SS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_merge_topk), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)merge_lds_bytes(SS_MAX_TOPK, cb)));
        s->lds_attr = cb;
    }
    const bool timed = ctx->opt("score.timing", 1) != 0;       // the two timing events of ss_last_kernel_ms(1) (each costs the stream a few us)
    if (timed) SS_HIP(ctx, hipEventRecord(ctx->ev[1][0], st));
    hipStream_t wst = st;                        // where k_wave_prep / k_score_wave go
    hipStream_t sst = st;                        // ... and k_score_slices (with the phrase kernels in front of it)
    if (pipe || pipe_s || small_alone) {
        const int n_ws = (int)std::min<int64_t>(ss_ctx::N_WAVE_STREAMS, std::max<int64_t>(1, ctx->opt("score.pipeline", 2)));
        const int wi = (int)(s->wave_turn++ % (unsigned)n_ws);
        if (!ctx->wave_stream[wi]) SS_HIP(ctx, hipStreamCreateWithFlags(&ctx->wave_stream[wi], hipStreamNonBlocking));
        if (!s->wave_ev[pb]) SS_HIP(ctx, hipEventCreateWithFlags(&s->wave_ev[pb], hipEventDisableTiming));
        wst = ctx->wave_stream[wi];
        if (pipe_split || small_side) {
            const int si = (wi + 1) % n_ws;                        // (n_ws >= 2: "score.pipeline" >= 2)
            if (!ctx->wave_stream[si]) SS_HIP(ctx, hipStreamCreateWithFlags(&ctx->wave_stream[si], hipStreamNonBlocking));
            if (!s->slice_ev[pb]) SS_HIP(ctx, hipEventCreateWithFlags(&s->slice_ev[pb], hipEventDisableTiming));
            sst = ctx->wave_stream[si];
        }
    }
    if (pipe_s) sst = wst;
    if (n_small) {                     // writes its queries' hits itself: the caller's stream, like every kernel that does — or stages them
        const int32_t rc_s = ss::launch_score_small(&p, (unsigned)n_small_a, (unsigned)n_small_b, !small_staged ? st : small_alone ? wst : sst);
        if (rc_s != 0) return ctx->fail(SS_ERR_HIP, "k_score_small: %s", hipGetErrorString((hipError_t)rc_s));
    }
    if (any_phrase) {                            // the phrase matches, in front of the kernel that merges them in (k_score_slices)
        hipStream_t pst = sst;
        if (!h_parts.empty()) hipLaunchKernelGGL(k_phrase_match, dim3((unsigned)h_parts.size()), dim3(PH_TPB), 0, pst, p);
        hipLaunchKernelGGL(k_phrase_close, dim3((unsigned)n_q), dim3(PH_TPB), 0, pst, p);
    }
    if (n_fast_slices) {
        if (!prep_done) ss::launch_wave_prep(&p, (unsigned)n_fast_slices, s->d_wprep2[pb].p, wst);
        ss::launch_score_wave(&p, (unsigned)n_fast_slices, s->d_wprep2[pb].p, wst);
    }
    if (n_slices > n_fast_slices) {
        ScoreParams ps = p;
        ps.order = p.order + n_fast_slices;
        hipLaunchKernelGGL(k_score_slices, dim3((unsigned)(n_slices - n_fast_slices)), dim3(TPB), lds_score, sst, ps);
    }
    if (pipe_s) {                                // the merge, on the caller's stream, behind this batch's k_score_slices
        SS_HIP(ctx, hipEventRecord(s->wave_ev[pb], wst));
        SS_HIP(ctx, hipStreamWaitEvent(st, s->wave_ev[pb], 0));
    }
    if (pipe_split || small_side) {
        SS_HIP(ctx, hipEventRecord(s->slice_ev[pb], sst));
        SS_HIP(ctx, hipStreamWaitEvent(st, s->slice_ev[pb], 0));
    }
    if (small_alone) {
        SS_HIP(ctx, hipEventRecord(s->wave_ev[pb], wst));
        SS_HIP(ctx, hipStreamWaitEvent(st, s->wave_ev[pb], 0));
    }
    if (small_staged) ss::launch_small_copy(&p, (unsigned)n_small, st);
    if (!fused && n_slices > n_fast_slices) hipLaunchKernelGGL(k_merge_topk, dim3((unsigned)n_q), dim3(TPB_M), lds_merge, st, p);
    if (pipe) {                                  // the merge, on the caller's stream, behind this batch's k_score_wave
        SS_HIP(ctx, hipEventRecord(s->wave_ev[pb], wst));
        SS_HIP(ctx, hipStreamWaitEvent(st, s->wave_ev[pb], 0));
    }
    if (!h_mergeq.empty()) hipLaunchKernelGGL(k_merge_flat, dim3((unsigned)h_mergeq.size()), dim3(TPB_MF), merge_lds_bytes(k, cb_flat), st, p);
    if (timed) {                                 // (pipelined: from the end of the previous batch's merge to the end of this one's)
        SS_HIP(ctx, hipEventRecord(ctx->ev[1][1], st));
        ctx->ev_valid[1] = true;
    }
    SS_HIP(ctx, hipEventRecord(s->batch_ev[pb], st));
    s->batch_ev_pending[pb] = true;
    SS_HIP(ctx, hipGetLastError());
    if (trace)
        fprintf(stderr, "[score trace] copies in + checks %.0f us, plan (%zu slices) %.0f us, staging %.0f us, H2D + allocs + params %.0f us, launches %.0f us%s\n",
                t_us(th0, th1), n_slices, t_us(th1, th2), t_us(th2, th3), t_us(th3, th4), t_us(th4, t_now()), pipe ? " (k_score_wave on the wave stream)" : "");
    if (dev_out) return SS_OK;                   // ordered on the ctx stream; ss_synchronize (or the stream's owner) waits
    auto capture_floor = [&]() {                 // experiment only: the next call of the same batch starts from these thresholds
#ifndef SS_EXP_FLOOR
        return;
#endif
        if (ctx->opt("score.debug_floor", 0) == 0) return;
        s->dbg_floor.assign((size_t)n_q, 0.0f);
        for (int q = 0; q < n_q; q++)
            if (n_hits_out[q] == k) {
                const double f = hits_out[(size_t)q * k + (k - 1)].final;
                float ff = (float)f;
                if ((double)ff > f) ff = std::nextafterf(ff, -INFINITY);
                if (ff > 0.0f && f == f) s->dbg_floor[q] = ff;
            }
    };
    if (one_copy) {
        unsigned char* const hp1 = s->h_res;
        SS_HIP(ctx, hipMemcpyAsync(hp1, s->d_hits.p, res_bytes, hipMemcpyDeviceToHost, st));
        SS_HIP(ctx, hipStreamSynchronize(st));
        std::memcpy(hits_out, hp1, res_rows * sizeof(ss_hit));
        std::memcpy(n_hits_out, hp1 + res_rows * sizeof(ss_hit), (size_t)n_q * sizeof(int32_t));
        capture_floor();
        return SS_OK;
    }
    SS_HIP(ctx, hipMemcpyAsync(hits_out, s->d_hits.p, (size_t)n_q * k * sizeof(ss_hit), hipMemcpyDefault, st));
    // (the counts through the context's pinned scratch: a small copy into pageable memory is staged and waited for by the runtime on its
    //  own, ~20 us that a copy into pinned memory does not cost)
    const size_t nh_bytes = (size_t)n_q * sizeof(int32_t);
    if (nh_bytes <= ss_ctx::PIN_SCRATCH) {
        ctx->pin_used = 0;
        int32_t* const hn = ctx->pin<int32_t>((size_t)n_q);
        SS_HIP(ctx, hipMemcpyAsync(hn, s->d_nhits.p, nh_bytes, hipMemcpyDeviceToHost, st));
        SS_HIP(ctx, hipStreamSynchronize(st));
        std::memcpy(n_hits_out, hn, nh_bytes);
        capture_floor();
        return SS_OK;
    }
    SS_HIP(ctx, hipMemcpyAsync(n_hits_out, s->d_nhits.p, nh_bytes, hipMemcpyDefault, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    capture_floor();
    return SS_OK;
}

}  // extern "C"
