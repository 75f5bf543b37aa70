// pagerank.hip — Topic-Sensitive PageRank power iteration for gfx950 (MI355X).
//
// Replaces ranking/pagerank.go:85-145 (updatePagerank + computeRankInherited),
// all categories of pagerank.go:54-63 at once: one K-wide sweep per iteration.
//
// Reference arithmetic per topic (Q3-Q6 of SURVEY.md §7):
//   w_p      = d * last[p] / outdeg(p)            for every p with outdeg>0   (:136)
//   total    = sum_p w_p + (1-d) * N                                          (:137,:112)
//   cur[v]   = (1/n if iteration==1 else 0) + sum_{p->v} w_p                  (:97-107,:140-142)
//   cur[v]   = (cur[v] + (1-d)) / total                                       (:117)
//   change   = sum_v |cur[v]-last[v]| ; loop while change > eps               (:93,:118)
//
// HBM layout (node-major, the K topic values of a node are contiguous so that ONE
// index read serves K gathers and a K=16 gather is one 128-byte line):
//   x     [n_local][GW]   rank of this rank's rows, updated in place
//   table [nd_int+1][GW]  contributions w_p of ALL non-dangling nodes, read by
//                         random gather; written for the next sweep; the last row is
//                         all zero (where the unused slots of a turn gather from)
//                         (world==1: ping-pong pair; world>1: own slice -> `send`,
//                          ss_pr_exchange all-gathers it into `table`)
// One kernel per sweep (k_pr_sweep for K >= 3, k_pr_step for K <= 2 on large graphs).
// The pull SpMV, the normalise, the L1 delta, the next sweep's contributions and
// their sum (next `total`) are fused; block partial sums are handed to the last
// block to arrive (write-through stores + ticket, no fences) and combined in a fixed
// order; that block also applies the stop rule — the loop needs no host round trip
// per iteration.
//
// Algorithmic bytes per sweep: 4E + 8N + 16*K*N.
#include "graph.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>
#include <queue>

namespace {

constexpr int TPB = 256;
constexpr int WAVES = TPB / 64;
constexpr int MAXK = 16;

struct PrCtl {
    double S[MAXK];       // normaliser (`totalValue`) for the next sweep
    double delta[MAXK];   // last L1 change
    double csum[MAXK];    // last contribution sum (diagnostics)
    double xz[MAXK];      // rank of EVERY row without in-edges (they all share one value per topic)
    double xz_in[MAXK];   // topic-sensitive teleport only: that rank for the rows INSIDE the topic's teleport set (xz: outside)
    double tele[MAXK];    // two-vector form only ("pr.affine"): the teleport of each COLUMN for the next sweep
    int32_t active[MAXK];
    int32_t iters[MAXK];
    int32_t sweep;        // sweeps completed
    int32_t n_active;
    uint32_t ticket;      // last-group arrival counter
    uint32_t stuck;       // k_pr_multi_n: a wait between two sweeps ran out of patience (the grid was not resident): the state is void
    uint32_t gticket[8];  // last-block-of-a-group arrival counters
};

// The two-vector form of the reference's recurrence (option "pr.affine", opt-in; DESIGN K1b).  Every topic of pagerank.go:85-124 runs
// the SAME linear map on the same graph and differs only in its start value u_k = 1/n_k (:104); with x = (p*u + q) / (r*u + s)
// elementwise (p, q vectors over the nodes, r, s scalars) one iteration maps (p, q, r, s) to
//     p' = M p + tau*r*1,  q' = M q + tau*s*1,  r' = W p + tau*N*r,  s' = W q + tau*N*s        (M: the inherited part, W: the sum of
// the contributions, tau = 1 - d; iteration 1 adds the start vector: p += 1) — so TWO vectors carry every topic, whatever K is.
// The state is kept scaled to r + s = 1.  Per-topic ranks, L1 changes and stop decisions are evaluated from (p, q, r) by streaming
// kernels; a topic's ranks are written out in the iteration it stops.  Not the reference's float64 operation order: ranks agree
// with the oracle to ~1e-13, iteration counts where the stop rule is not at a rounding tie.
constexpr int AFF_MAXK = 256;
struct AffCtl {
    double u[AFF_MAXK];       // 1 / n_topic
    double delta[AFF_MAXK];   // last L1 change of the topic
    int32_t active[AFF_MAXK], iters[AFF_MAXK], just[AFF_MAXK];   // just: stopped in the iteration that has just been evaluated
    double r_x, r_prev, r_next;      // r of the stored vectors, of the previous ones, of the next sweep's result
    double s_x, s_prev, s_next;      // ... and s (the state is kept scaled to r + s = 1: s alone vanishes when d = 1)
    double xz_prev[2];               // the edge-less rows' (p, q) before the last sweep
    int32_t n_active, n_just, k_real, it;
};

enum : uint32_t { W_SEG = 0, W_WAVE = 1, W_GROUP = 2, W_ZERO = 3,   // GW < 8 (k_pr_step): block-owned items
                  // GW >= 8 (k_pr_sweep): every item belongs to ONE wave
                  V_SEG = 8,     // a <= SEGW-edge piece of a row with more than T_MULTI in-edges (partials + ticket)
                  V_ROWW = 9,    // a whole row, T_QUAD < in-edges <= T_MULTI
                  V_QUAD = 10,   // `count` rows, one per lane group and turn, each `nseg` (= chunks per row) 16-edge chunks long
                  V_DEG = 11,    // `count` rows of EXACTLY `nseg` (<= 8) in-edges: several rows per lane group and chunk
                  V_ZERO = 12 }; // `count` non-dangling rows without in-edges

struct WorkItem {
    uint32_t kind;
    uint32_t row;     // first local row
    uint32_t count;   // rows (WAVE/GROUP/ZERO) or segment index (SEG)
    uint32_t nseg;    // SEG: segments of this row
    uint32_t sbase;   // SEG: index of the row's first segment partial
    uint32_t tix;     // SEG: per-row ticket index
    uint32_t beg, end; // k_pr_sweep items: the item's in-edges are in_src[beg .. end)
};

struct PrParams {
    const uint32_t* in_ptr;
    const uint32_t* in_src;
    const uint32_t* outdeg;
    double* x;
    const double* tab_rd[2];
    double* tab_wr[2];
    const WorkItem* work;
    double* partials;     // [nblocks][2][GW]
    double* segpart;      // [nsegs][GW]
    uint32_t* rowticket;  // [n multi-segment rows]
    PrCtl* ctl;
    const double* x0;     // [GW] 1/n_topic
    double d, teleport, eps, tele_n;
    int32_t max_iter, k_topics, world;
    uint32_t sl_nd, cnt_nd, sl_d, cnt_d, seg_edges, n_items;
    uint32_t pos_nd, pos_d;   // rows WITH in-edges per class (they come first: rows are in-degree sorted)
    // opt-in true topic-sensitive teleport (ss_pr_set_teleport; null = the reference's uniform teleport):
    const uint32_t* memb;     // [n_local] bit k: the row's node is in topic k's teleport set
    const double* tin;        // [MAXK] teleport of a member: (1-d) * N / |set_k|
    const double* nz_in;      // [MAXK] rows without in-edges (this rank) inside topic k's set
    uint32_t ts_mask;         // bit k: topic k has a teleport set (others keep the uniform teleport)
    uint32_t zrow;            // index of the table's all-zero row (= nd_int): where the unused slots of a chunk gather from
    const uint32_t* woff;     // k_pr_sweep: [waves][8]: wave w's items of class c are work[woff[8w+c] .. woff[8w+c+1])
    uint32_t stagger_div;     // k_pr_sweep: blocks per arrival round (= CUs); 0 = every block walks the classes in the same order
    uint32_t stagger_code;    // start classes of the rounds as base-6 digits (0 = round r starts at position r of the class order)
    uint32_t class_order;     // k_pr_sweep: the order in which a wave walks its six work classes, 3 bits per position
    uint32_t n_order;         // k_pr_sweep_n: the order of its four phases, 2 bits per position
    double* x_alt;            // two-vector form: sweep s reads x (s even) / x_alt (s odd) and writes the other one; null otherwise
    AffCtl* aff;              // two-vector form ("pr.affine"): its control block; null otherwise
    const double* tele_col;   // ... and the per-column teleport (ctl->tele); null = the uniform p.teleport
#ifdef SS_PR_EXP_KINDMASK
    uint32_t kind_mask;       // experiment builds only: run just these work classes (bit = kind)
#endif
};

// ---- reductions --------------------------------------------------------------

// sum over the lanes of a wave that hold the same topic (lane % GW), fixed butterfly order
template <int GW>
__device__ __forceinline__ double wave_sum_topic(double v) {
#pragma unroll
    for (int off = GW; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Rows without in-edges inherit nothing: cur = (1/n if first sweep) + 0, so after the normalise they ALL
// hold the same value per topic.  They are never stored or streamed; this is that shared value.
__device__ __forceinline__ double zero_row_rank(const PrParams& p, int sweep, double S, double x0) {
    return ((sweep == 0 ? x0 : 0.0) + p.teleport) / S;            // pagerank.go:104,117
}
// Teleport of (row, topic).  Reference: the absolute (1-d) for every node (pagerank.go:117).  With a teleport set
// (Haveliwala's topic-sensitive PageRank, README.md:9 — opt-in) the same total mass (1-d)*N is spread
// over the set's nodes only, so the normaliser S = sum w + (1-d)*N (pagerank.go:112) keeps its meaning.
__device__ __forceinline__ double teleport_of(const PrParams& p, uint32_t lrow, int t) {
    if (!p.memb || !((p.ts_mask >> t) & 1u)) return p.teleport;
    return ((p.memb[lrow] >> t) & 1u) ? p.tin[t] : 0.0;
}
__device__ __forceinline__ double zero_row_rank_ts(const PrParams& p, int sweep, double S, double x0, double tele) {
    return ((sweep == 0 ? x0 : 0.0) + tele) / S;
}

// The control block as the persistent multi-sweep kernel (k_pr_multi_n) needs it: written by the last block of sweep i, read by every
// block of sweep i + 1 INSIDE one launch, i.e. across CUs and XCDs with no kernel boundary in between — write-through stores and
// L1-bypassing loads (sc1; scalar loads would come from the never-refreshed scalar cache).  The one-sweep kernels use plain accesses.
template <int PS, typename T>
__device__ __forceinline__ T ctl_ld(const T* q) {
    if constexpr (PS) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *q;
}
template <int PS, typename T>
__device__ __forceinline__ void ctl_st(T* q, T v) {
    if constexpr (PS) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *q = v;
}

template <int PS = 0>
__device__ __forceinline__ void finalize_ctl(const PrParams& p, const double* dl, const double* cs, bool is_begin) {
    PrCtl* ctl = p.ctl;
    if (p.aff) {
        // columns 0 / 1 = p / q.  Both are divided by the common sigma = r' + s' (r' = W p + tau*N*r, s' = W q + tau*N*s), their
        // teleports are tau*r and tau*s.
        AffCtl* a = p.aff;
        if (is_begin) {
            // start: p = 1, q = 0, r = 0, s = 1
            const double r1 = cs[0], s1 = cs[1] + p.tele_n, sigma = r1 + s1;
            for (int k = 0; k < MAXK; k++) {
                const bool real = k < 2;
                ctl->xz[k] = real ? p.x0[k] : 0.0;
                ctl->xz_in[k] = ctl->xz[k];
                ctl->S[k] = real ? sigma : 1.0;
                ctl->csum[k] = real ? cs[k] : 0.0;
                ctl->delta[k] = 0.0;
                ctl->active[k] = real ? 1 : 0;
                ctl->iters[k] = 0;
                ctl->tele[k] = 0.0;
            }
            ctl->tele[1] = p.teleport;                                // tau * s with s = 1; tele[0] = tau * r with r = 0
            a->r_x = 0.0; a->s_x = 1.0;
            a->r_prev = 0.0; a->s_prev = 1.0;
            a->r_next = r1 / sigma; a->s_next = s1 / sigma;
            a->xz_prev[0] = ctl->xz[0];
            a->xz_prev[1] = ctl->xz[1];
            a->it = 0;
            ctl->sweep = 0;
            ctl->n_active = 2;
            return;
        }
        const int it = ctl->sweep + 1;
        a->r_prev = a->r_x; a->s_prev = a->s_x;
        a->r_x = a->r_next; a->s_x = a->s_next;                       // (r, s) of the vectors this sweep has written
        for (int k = 0; k < 2; k++) {
            a->xz_prev[k] = ctl->xz[k];
            ctl->xz[k] = zero_row_rank_ts(p, ctl->sweep, ctl->S[k], p.x0[k], ctl->tele[k]);
            ctl->xz_in[k] = ctl->xz[k];
            ctl->iters[k] = it;
            ctl->csum[k] = cs[k];
        }
        const double r1 = cs[0] + p.tele_n * a->r_x;                  // W p + tau*N*r
        const double s1 = cs[1] + p.tele_n * a->s_x;                  // W q + tau*N*s
        const double sigma = r1 + s1;
        a->r_next = r1 / sigma;
        a->s_next = s1 / sigma;
        ctl->tele[0] = p.teleport * a->r_x;
        ctl->tele[1] = p.teleport * a->s_x;
        ctl->S[0] = ctl->S[1] = sigma;
        ctl->sweep = it;
        return;
    }
    if (is_begin) {
        for (int k = 0; k < MAXK; k++) {
            const bool real = k < p.k_topics;
            ctl->xz[k] = real ? p.x0[k] : 0.0;
            ctl->xz_in[k] = real ? p.x0[k] : 0.0;
            ctl->S[k] = real ? cs[k] + p.tele_n : 1.0;
            ctl->csum[k] = real ? cs[k] : 0.0;
            ctl->delta[k] = 0.0;
            ctl->active[k] = real ? 1 : 0;
            ctl->iters[k] = 0;
        }
        ctl->sweep = 0;
        ctl->n_active = p.k_topics;
        return;
    }
    const int sw = ctl_ld<PS>(&ctl->sweep);
    const int it = sw + 1;
    int na = 0;
    for (int k = 0; k < p.k_topics; k++) {
        if (ctl_ld<PS>(&ctl->active[k])) {
            const double Sk = ctl_ld<PS>(&ctl->S[k]);
            ctl_st<PS>(&ctl->iters[k], it);
            ctl_st<PS>(&ctl->delta[k], dl[k]);              // includes the rows without in-edges (added by the caller)
            if (p.memb && ((p.ts_mask >> k) & 1u)) {
                ctl_st<PS>(&ctl->xz[k], zero_row_rank_ts(p, sw, Sk, p.x0[k], 0.0));
                ctl_st<PS>(&ctl->xz_in[k], zero_row_rank_ts(p, sw, Sk, p.x0[k], p.tin[k]));
            } else {
                const double xz = zero_row_rank(p, sw, Sk, p.x0[k]);
                ctl_st<PS>(&ctl->xz[k], xz);
                ctl_st<PS>(&ctl->xz_in[k], xz);
            }
            bool cont = dl[k] > p.eps;                      // pagerank.go:93
            if (p.max_iter > 0 && it >= p.max_iter) cont = false;
            ctl_st<PS>(&ctl->active[k], cont ? 1 : 0);
            na += cont ? 1 : 0;
            ctl_st<PS>(&ctl->S[k], cs[k] + p.tele_n);       // pagerank.go:111-112
            ctl_st<PS>(&ctl->csum[k], cs[k]);
        }
    }
    ctl_st<PS>(&ctl->n_active, na);
    if constexpr (PS) {
        // `sweep` is what the other blocks poll between two sweeps: it goes last, behind everything else this thread has stored
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    ctl_st<PS>(&ctl->sweep, it);
}

// Block partial -> global partials; the last block to arrive sums all partials in a
// fixed order and either finalises the control block (world==1) or leaves this
// rank's totals in the tail rows of the send buffer (world>1).
template <int GW, int PS = 0>
__device__ __forceinline__ void block_reduce_and_publish(const PrParams& p, double dsum, double csum, double* tail,
                                                         bool is_begin) {
    __shared__ double red[WAVES][2][MAXK];
    __shared__ double tot[2][MAXK];
    __shared__ double colsum[TPB];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = lane % GW;
    dsum = wave_sum_topic<GW>(dsum);
    csum = wave_sum_topic<GW>(csum);
    if (lane < GW) {
        red[wave][0][t] = dsum;
        red[wave][1][t] = csum;
    }
    __syncthreads();
    // Hand-off of the block's partial sums to the last block to arrive, without fences (a release would write back the
    // XCD's whole dirty L2 — this sweep's rank and table stores — once per block; MI355X_MICROARCH.md, hand-off forms):
    // every partial is stored write-through (sc1), the storing wave drains its stores, one lane takes a ticket with an
    // agent-scope atomic, and the last block reads the partials with sc1 loads.
    if (threadIdx.x < 2 * GW) {
        const int which = threadIdx.x / GW, tt = threadIdx.x % GW;
        double v = red[0][which][tt];
#pragma unroll
        for (int w = 1; w < WAVES; w++) v += red[w][which][tt];
        __hip_atomic_store(&p.partials[(size_t)blockIdx.x * 2 * GW + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // Two levels, so that nobody sums a thousand rows alone: the blocks form NG groups (block % NG); the last block of a
    // group to arrive sums the group's rows (one batch of loads per thread) into a group row, the last group to finish
    // sums the NG group rows.  Fixed grouping, fixed order: deterministic.
    constexpr unsigned NG = 8;
    constexpr int NCOL = 2 * GW;
    constexpr int NPART = TPB / NCOL;
    const unsigned ng = min(NG, gridDim.x);
    const unsigned grp = blockIdx.x % ng;
    const unsigned members = (gridDim.x - grp + ng - 1) / ng;         // blocks b = grp, grp + ng, ...
    double* const gpart = p.partials + (size_t)gridDim.x * NCOL;      // [NG][NCOL] behind the block rows
    if (threadIdx.x == 0) {
        if constexpr (PS == 2) {
            // fence form of k_pr_multi_n: this block's table and rank stores (plain: they stay in the XCD's L2 for its own gathers) are
            // written back before the block counts as arrived; the wait behind the fence is spelled out (ROCm 7.2 can drop the fence's own)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned prev = __hip_atomic_fetch_add(&p.ctl->gticket[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == members - 1;
    }
    __syncthreads();
    if (!s_last) return;
    const int col = threadIdx.x % NCOL, part = threadIdx.x / NCOL;
    {
        double acc = 0.0;
        for (unsigned m0 = part; m0 < members; m0 += 16 * NPART) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const unsigned m = m0 + u * NPART;
                v[u] = __hip_atomic_load(&p.partials[(size_t)(grp + ng * (m < members ? m : m0)) * NCOL + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < 16; u++)
                if (m0 + u * NPART < members) acc += v[u];
        }
        colsum[threadIdx.x] = acc;
    }
    __syncthreads();
    if (threadIdx.x < NCOL) {
        double v = 0.0;
        for (int q = 0; q < NPART; q++) v += colsum[q * NCOL + threadIdx.x];
        __hip_atomic_store(&gpart[(size_t)grp * NCOL + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) ctl_st<PS>(&p.ctl->gticket[grp], 0u);       // every member has arrived: ready for the next sweep (drained below, in front of this block's ticket)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(&p.ctl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == ng - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < NCOL) {
        double v = 0.0;
        for (unsigned gq = 0; gq < ng; gq++)
            v += __hip_atomic_load(&gpart[(size_t)gq * NCOL + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tot[threadIdx.x / GW][threadIdx.x % GW] = v;
    }
    __syncthreads();
    // rows without in-edges: all equal, so their L1 change is count * |new - old| (not streamed, see zero_row_rank)
    if (!is_begin && threadIdx.x < GW && ctl_ld<PS>(&p.ctl->active[threadIdx.x])) {
        const double n_zero = (double)((p.cnt_nd - p.pos_nd) + (p.cnt_d - p.pos_d));
        if (p.memb && ((p.ts_mask >> threadIdx.x) & 1u)) {
            // two values per topic: inside and outside the teleport set
            const int k = threadIdx.x;
            const double out_new = zero_row_rank_ts(p, ctl_ld<PS>(&p.ctl->sweep), ctl_ld<PS>(&p.ctl->S[k]), p.x0[k], 0.0);
            const double in_new = zero_row_rank_ts(p, ctl_ld<PS>(&p.ctl->sweep), ctl_ld<PS>(&p.ctl->S[k]), p.x0[k], p.tin[k]);
            tot[0][k] += (n_zero - p.nz_in[k]) * fabs(out_new - ctl_ld<PS>(&p.ctl->xz[k])) + p.nz_in[k] * fabs(in_new - ctl_ld<PS>(&p.ctl->xz_in[k]));
        } else {
            const double xz_new = zero_row_rank(p, ctl_ld<PS>(&p.ctl->sweep), ctl_ld<PS>(&p.ctl->S[threadIdx.x]), p.x0[threadIdx.x]);
            tot[0][threadIdx.x] += n_zero * fabs(xz_new - ctl_ld<PS>(&p.ctl->xz[threadIdx.x]));
        }
    }
    __syncthreads();
    if (p.world == 1) {
        if (threadIdx.x == 0) {
            double dl[MAXK], cs[MAXK];
            for (int k = 0; k < MAXK; k++) {
                dl[k] = k < GW ? tot[0][k] : 0.0;
                cs[k] = k < GW ? tot[1][k] : 0.0;
            }
            ctl_st<PS>(&p.ctl->ticket, 0u);                          // (in front of finalize_ctl: its last store releases the next sweep)
            finalize_ctl<PS>(p, dl, cs, is_begin);
        }
    } else {
        // tail rows of this rank's all-gather piece: row sl_nd-2 = contribution sums, row sl_nd-1 = deltas
        if (threadIdx.x < GW) {
            tail[(size_t)(p.sl_nd - 2) * GW + threadIdx.x] = tot[1][threadIdx.x];
            tail[(size_t)(p.sl_nd - 1) * GW + threadIdx.x] = tot[0][threadIdx.x];
        }
        if (threadIdx.x == 0) p.ctl->ticket = 0;
    }
}

// Streaming data (ranks, indices, next contributions) is touched once per sweep: mark it
// non-temporal so that it does not push the randomly gathered table out of L2.
#ifndef SS_PR_NO_NT
#define NT_LOAD(p) __builtin_nontemporal_load(p)
#define NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define NT_LOAD(p) (*(p))
#define NT_STORE(v, p) (*(p) = (v))
#endif

// ---- gather ------------------------------------------------------------------
// T[row][t] addressed as table base (wave-uniform, scalar registers) + 32-bit byte offset: the contribution table of a rank
// stays below 4 GiB (n_nd * GW * 8 bytes, checked in ss_pr_create), so no 64-bit vector address arithmetic is needed
template <int GW>
__device__ __forceinline__ double tab_at(const double* __restrict__ T, uint32_t row, int t) {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(T) + (size_t)((row * (uint32_t)GW + (uint32_t)t) * 8u));
}


constexpr uint32_t SRC_MASK = 0x7FFFFFFFu;   // in_src bit 31 = "last in-edge of its row" (graph.hip)
constexpr int CH = 16;                       // edges per chunk of the GW=16 path

// sum of T[src][t] over edges beg+first, beg+first+stride, ... < end; 4 gathers in flight
template <int GW>
__device__ __forceinline__ double gather_sum(const double* __restrict__ T, const uint32_t* __restrict__ in_src,
                                             size_t beg, size_t end, unsigned first, unsigned stride, int t) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    size_t e = beg + first;
    for (; e + 3 * (size_t)stride < end; e += 4 * (size_t)stride) {
        const uint32_t s0 = in_src[e] & SRC_MASK, s1 = in_src[e + stride] & SRC_MASK,
                       s2 = in_src[e + 2 * (size_t)stride] & SRC_MASK, s3 = in_src[e + 3 * (size_t)stride] & SRC_MASK;
        a0 += T[(size_t)s0 * GW + t];
        a1 += T[(size_t)s1 * GW + t];
        a2 += T[(size_t)s2 * GW + t];
        a3 += T[(size_t)s3 * GW + t];
    }
    for (; e < end; e += stride) a0 += T[(size_t)(in_src[e] & SRC_MASK) * GW + t];
    return (a0 + a1) + (a2 + a3);
}

// ---- the sweep, K <= 2 (GW = 1 / 2) on graphs whose padded table would not stay cache-resident ---------------
// Persistent grid: a fixed number of blocks walks the work table round-robin, so the
// per-launch costs (partials, ticket) are paid ~2k times, not per work item.  (K >= 3: k_pr_sweep below.)
template <int GW>
__global__ __launch_bounds__(TPB) void k_pr_step(PrParams p) {
    constexpr int NSLOT = 64 / GW;
    __shared__ double rowred[WAVES][MAXK];
    __shared__ int s_rowlast;

    PrCtl* ctl = p.ctl;
    if (ctl->n_active == 0) return;   // every topic converged: the launch is a no-op
    const int sweep = ctl->sweep;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = lane % GW, slot = lane / GW;
    const double S = ctl->S[t];
    const bool act = ctl->active[t] != 0;
    const double x0 = sweep == 0 ? p.x0[t] : 0.0;      // Q4: iteration 1 accumulates onto 1/n
    const double* __restrict__ T = p.tab_rd[sweep & 1];
    double* __restrict__ Tw = p.tab_wr[sweep & 1];

    double dsum = 0.0, csum = 0.0;

    auto finish = [&](uint32_t lrow, double y) __attribute__((always_inline)) {
        const double xo = NT_LOAD(&p.x[(size_t)lrow * GW + t]);
        const uint32_t od = lrow < p.sl_nd ? NT_LOAD(&p.outdeg[lrow]) : 1u;
        y += x0;
        const size_t xi = (size_t)lrow * GW + t;
        double xn = (y + teleport_of(p, lrow, t)) / S;  // pagerank.go:117
        if (act) {
            NT_STORE(xn, &p.x[xi]);
            dsum += fabs(xn - xo);                      // pagerank.go:118
        } else {
            xn = xo;                                    // converged topic: frozen
        }
        if (lrow < p.sl_nd) {                           // non-dangling row: next sweep's contribution
            const double c = p.d * xn / (double)od;     // pagerank.go:136
            NT_STORE(c, &Tw[xi]);
            csum += c;                                  // pagerank.go:137
        }
    };

    for (uint32_t item = blockIdx.x; item < p.n_items; item += gridDim.x) {
        const WorkItem w = p.work[item];
#ifdef SS_PR_EXP_KINDMASK
        if (!((p.kind_mask >> w.kind) & 1u)) continue;
#endif
        if (w.kind == W_SEG) {
            // one block per segment of a long row
            const uint32_t lrow = w.row;
            const size_t rbeg = p.in_ptr[lrow], rend = p.in_ptr[lrow + 1];
            const size_t beg = rbeg + (size_t)w.count * p.seg_edges;
            const size_t end = min(rend, beg + (size_t)p.seg_edges);
            double acc = gather_sum<GW>(T, p.in_src, beg, end, wave * NSLOT + slot, WAVES * NSLOT, t);
            acc = wave_sum_topic<GW>(acc);
            __syncthreads();                            // rowred / s_rowlast reuse across items
            if (lane < GW) rowred[wave][t] = acc;
            __syncthreads();
            double y = 0.0;
            if (threadIdx.x < GW) {
                y = rowred[0][t];
#pragma unroll
                for (int q = 1; q < WAVES; q++) y += rowred[q][t];
            }
            if (w.nseg == 1) {
                if (threadIdx.x < GW) finish(lrow, y);
            } else {
                // several blocks share this row: publish the segment partial; the last
                // arriver adds the partials in segment order and finishes the row
                // (the fence-free hand-off of block_reduce_and_publish / long_rows: the partial is stored write-through by
                // wave 0, which drains its stores and then takes the ticket itself; the last arriver reads with sc1 loads)
                if (threadIdx.x < GW) __hip_atomic_store(&p.segpart[(size_t)(w.sbase + w.count) * GW + t], y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (threadIdx.x == 0) {
                    const unsigned prev = __hip_atomic_fetch_add(&p.rowticket[w.tix], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int last = prev == w.nseg - 1;
                    if (last) __hip_atomic_store(&p.rowticket[w.tix], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_rowlast = last;
                }
                __syncthreads();
                if (s_rowlast && threadIdx.x < GW) {
                    double ys = 0.0;
                    for (uint32_t q = 0; q < w.nseg; q++)
                        ys += __hip_atomic_load(&p.segpart[(size_t)(w.sbase + q) * GW + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    finish(lrow, ys);
                }
            }
        } else if (w.kind == W_WAVE) {
            // one wave per row, the wave's lane groups stride the row's in-edges
            if ((uint32_t)wave < w.count) {
                const uint32_t lrow = w.row + wave;
                const size_t beg = p.in_ptr[lrow], end = p.in_ptr[lrow + 1];
                double acc = gather_sum<GW>(T, p.in_src, beg, end, slot, NSLOT, t);
                acc = wave_sum_topic<GW>(acc);
                if (slot == 0) finish(lrow, acc);
            }
        } else if (w.kind == W_GROUP) {
            // one lane group (GW lanes) per row; rows are degree-sorted so trip counts match inside a wave
            for (uint32_t r = wave * NSLOT + slot; r < w.count; r += WAVES * NSLOT) {
                const uint32_t lrow = w.row + r;
                const size_t beg = p.in_ptr[lrow], end = p.in_ptr[lrow + 1];
                double acc = 0.0;
                for (size_t e = beg; e < end; e++) acc += T[(size_t)(p.in_src[e] & SRC_MASK) * GW + t];
                finish(lrow, acc);
            }
        } else {
            // non-dangling rows without in-edges: their rank is the shared value xz, only the next
            // contribution d*xz/outdeg has to be written (dangling ones need nothing at all)
            const bool ts = p.memb && ((p.ts_mask >> t) & 1u);
            const double xz_out = act ? (ts ? zero_row_rank_ts(p, sweep, S, p.x0[t], 0.0) : zero_row_rank(p, sweep, S, p.x0[t])) : ctl->xz[t];
            const double xz_inn = ts ? (act ? zero_row_rank_ts(p, sweep, S, p.x0[t], p.tin[t]) : ctl->xz_in[t]) : xz_out;
            const uint32_t nel = w.count * GW;
            for (uint32_t i = threadIdx.x; i < nel; i += TPB) {
                const uint32_t lrow = w.row + i / GW;
                const double xz = ts && ((p.memb[lrow] >> t) & 1u) ? xz_inn : xz_out;
                const double c = p.d * xz / (double)NT_LOAD(&p.outdeg[lrow]);   // pagerank.go:136
                NT_STORE(c, &Tw[(size_t)lrow * GW + t]);
                csum += c;                                                        // pagerank.go:137
            }
        }
    }

    block_reduce_and_publish<GW>(p, dsum, csum, Tw, false);
}

// ---- the sweep, K >= 5 (GW = 8 / 16): k_pr_sweep -------------------------------------------------------------
// Every work item belongs to ONE wave (no block barriers on the way), control flow is wave-uniform, and every path
// is the same software pipeline: a lane group (GW lanes = the GW topic values of one table row) takes 16 in-edges
// per turn; the index words of turn i+1 are requested before the 16 whole-row gathers of turn i are issued, so a turn
// costs ONE memory latency.  Slots of a turn that hold no edge gather the table's all-zero row (p.zrow) and add an
// exact 0.0 — there are no per-edge predicates, flags or LDS traffic anywhere.  Row ends are known from the item:
//   V_SEG / V_ROWW   the wave's lane groups share one long row (cross-group butterfly at the end)
//   V_QUAD           one row per lane group, all rows of the item `nch` turns long (rows are in-degree sorted)
//   V_DEG<R>         R rows of exactly D <= 16/R in-edges per lane group and turn, at fixed slots
// Measured on the 10M/50M R-MAT, K=16 (MI355X): 1.41 ms per sweep for the block-per-item / flag-driven kernel this
// replaces; every class alone was latency-bound (0.60 + 0.53 + 0.57 + 0.16 ms, tools/pr_kmask.sh).
constexpr uint32_t SEGW = 2048;      // edges per V_SEG piece

// TS: the state holds teleport sets (ss_pr_set_teleport).  A kernel of its own, so that the reference's path carries no
// membership loads (a load under a branch in finish_row makes the compiler drain the loads in flight: s_waitcnt vmcnt(0)).
template <int GW, bool TS>
struct SweepCtx {
    const PrParams& p;
    const double* __restrict__ T;
    double* __restrict__ Tw;
    double S, x0;
    bool act;
    int t, gbase, slot;
    double dsum, csum;
};

// edges of [epos, lim) that fall into a 16-slot turn starting at epos: saturating, so that a turn past the end has none
__device__ __forceinline__ uint32_t turn_fill(uint32_t epos, uint32_t lim) {
    return min((uint32_t)CH, __builtin_elementwise_sub_sat(lim, epos));
}

// the 16 index words of a lane group's turn: slot j = r*GW + t holds edge `epos + j` for j < n, the zero row otherwise
template <int GW>
__device__ __forceinline__ void idx_turn(const uint32_t* __restrict__ in_src, uint32_t epos, uint32_t n, uint32_t zrow, int t, uint32_t (&src)[CH / GW]) {
#pragma unroll
    for (int r = 0; r < CH / GW; r++) {
        const uint32_t j = (uint32_t)(r * GW + t);
        const uint32_t raw = NT_LOAD(&in_src[j < n ? epos + j : 0u]);       // unconditional load (edge 0 exists whenever an item has edges)
        src[r] = j < n ? (raw & SRC_MASK) : zrow;
    }
}
template <int GW>
__device__ __forceinline__ void gather_turn(const double* __restrict__ T, const uint32_t (&src)[CH / GW], int t, int gbase, double (&v)[CH]) {
#pragma unroll
    for (int j = 0; j < CH; j++) {
        const uint32_t sj = (uint32_t)__shfl((int)src[j / GW], gbase + (j % GW), 64);
        v[j] = tab_at<GW>(T, sj, t);
    }
}

template <int GW, bool TS>
__device__ __forceinline__ void finish_row(SweepCtx<GW, TS>& c, uint32_t lrow, double y, double xo, uint32_t od) {
    const PrParams& p = c.p;
    y += c.x0;
    const size_t xi = (size_t)lrow * GW + c.t;
    double tele = p.teleport;
    if constexpr (TS) tele = teleport_of(p, lrow, c.t);
    double xn = (y + tele) / c.S;                             // pagerank.go:117
    if (c.act) {
        NT_STORE(xn, &p.x[xi]);
        c.dsum += fabs(xn - xo);                              // pagerank.go:118
    } else {
        xn = xo;                                              // converged topic: frozen
    }
    if (lrow < p.sl_nd) {                                     // non-dangling row: next sweep's contribution
        const double cc = p.d * xn / (double)od;              // pagerank.go:136
        NT_STORE(cc, &c.Tw[xi]);
        c.csum += cc;                                         // pagerank.go:137
    }
}

// V_SEG / V_ROWW: the wave's items are long rows (or <= SEGW-edge pieces of the longest ones); turn i of an item
// gives lane group s the edges beg + 64*i + 16*s ...  The pipeline runs across the items: the last turn of one item
// requests the first index words of the next.
template <int GW, bool TS>
__device__ __forceinline__ void long_rows(SweepCtx<GW, TS>& c, const WorkItem* __restrict__ work, uint32_t i0, uint32_t i1, int lane) {
    constexpr int NS = 64 / GW;
    constexpr uint32_t TW = NS * CH;                          // edges per wave turn
    const PrParams& p = c.p;
    if (i0 >= i1) return;
    WorkItem cur = work[i0], nxt = work[i0 + 1];              // the table ends with two unused items: reading ahead is safe
    uint32_t src_n[CH / GW];
    {
        const uint32_t e0 = cur.beg + (uint32_t)c.slot * CH;
        idx_turn<GW>(p.in_src, e0, turn_fill(e0, cur.end), p.zrow, c.t, src_n);
    }
    for (uint32_t it = i0; it < i1; it++) {
        const WorkItem nn = work[it + 2];
        const uint32_t lrow = cur.row;
        // the row's old rank and out-degree: asked for now, used after the last turn
        const double xo = NT_LOAD(&p.x[(size_t)lrow * GW + c.t]);
        const uint32_t od = lrow < p.sl_nd ? NT_LOAD(&p.outdeg[lrow]) : 1u;
        const uint32_t turns = (cur.end - cur.beg + TW - 1) / TW;
        const bool more = it + 1 < i1;
        double acc = 0.0;
        for (uint32_t i = 0; i < turns; i++) {
            uint32_t src[CH / GW];
#pragma unroll
            for (int r = 0; r < CH / GW; r++) src[r] = src_n[r];
            const bool last = i + 1 == turns;                 // scalar
            const uint32_t e1 = (last ? nxt.beg : cur.beg + (i + 1) * TW) + (uint32_t)c.slot * CH;
            const uint32_t lim = last ? (more ? nxt.end : 0u) : cur.end;
            idx_turn<GW>(p.in_src, e1, turn_fill(e1, lim), p.zrow, c.t, src_n);
            double v[CH];
            gather_turn<GW>(c.T, src, c.t, c.gbase, v);
#pragma unroll
            for (int j = 0; j < CH; j++) acc += v[j];
        }
        const double y = wave_sum_topic<GW>(acc);
        if (cur.kind == V_ROWW) {
            if (lane < GW) finish_row<GW>(c, lrow, y, xo, od);
        } else {
            // several waves (of any blocks) share this row: publish the piece's sum; the last to arrive adds the
            // pieces in order and finishes the row
            // (write-through stores, drained, then the ticket; the last arriver reads with sc1 loads: no fences — see
            // block_reduce_and_publish)
            if (lane < GW) __hip_atomic_store(&p.segpart[(size_t)(cur.sbase + cur.count) * GW + c.t], y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned prev = 0;
            if (lane == 0) prev = __hip_atomic_fetch_add(&p.rowticket[cur.tix], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            prev = (unsigned)__builtin_amdgcn_readfirstlane((int)prev);
            if (prev == cur.nseg - 1) {
                if (lane == 0) __hip_atomic_store(&p.rowticket[cur.tix], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane < GW) {
                    double ys = 0.0;
                    for (uint32_t q = 0; q < cur.nseg; q++)
                        ys += __hip_atomic_load(&p.segpart[(size_t)(cur.sbase + q) * GW + c.t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    finish_row<GW>(c, lrow, ys, xo, od);
                }
            }
        }
        cur = nxt;
        nxt = nn;
    }
}

// V_QUAD: an item = rows row .. row+count-1, lane group s takes rows row + q*NS + s (q = 0 .. nq-1, nq <= GW), every row
// is walked in nch (= item.nseg) turns (its own length decides how many slots of a turn are real).  The bounds and
// out-degrees of ALL rows of an item come in one request (lane t of group s holds row `row + t*NS + s`), one item ahead;
// the pipeline runs across row groups and items.
template <int GW, bool TS>
__device__ __forceinline__ void quad_rows(SweepCtx<GW, TS>& c, const WorkItem* __restrict__ work, uint32_t i0, uint32_t i1) {
    constexpr int NS = 64 / GW;
    const PrParams& p = c.p;
    if (i0 >= i1) return;
    const uint32_t myq = (uint32_t)c.t * NS + (uint32_t)c.slot;
    auto bounds = [&](const WorkItem& w, bool live, uint32_t& bv, uint32_t& ev, uint32_t& ov) __attribute__((always_inline)) {
        const bool have = live && myq < w.count;
        const uint32_t rq = live ? w.row + (have ? myq : 0u) : 0u;
        const uint32_t b = p.in_ptr[rq], e = p.in_ptr[rq + 1];
        bv = b;
        ev = have ? e : b;
        ov = rq < p.sl_nd ? NT_LOAD(&p.outdeg[rq]) : 1u;
    };
    WorkItem cur = work[i0], nxt = work[i0 + 1];
    uint32_t cb, ce, co, nb, ne, no;
    bounds(cur, true, cb, ce, co);
    bounds(nxt, i0 + 1 < i1, nb, ne, no);
    uint32_t it = i0, q = 0, ch = 0;                          // scalar: item, row group and turn inside it
    uint32_t nq = (cur.count + NS - 1) / NS, nch = cur.nseg;
    uint32_t src_n[CH / GW];
    {
        const uint32_t b0 = (uint32_t)__shfl((int)cb, c.gbase, 64), e0 = (uint32_t)__shfl((int)ce, c.gbase, 64);
        idx_turn<GW>(p.in_src, b0, turn_fill(b0, e0), p.zrow, c.t, src_n);
    }
    double acc = 0.0;
    while (it < i1) {
        uint32_t src[CH / GW];
#pragma unroll
        for (int r = 0; r < CH / GW; r++) src[r] = src_n[r];
        // the turn after this one: same row group, the next one, or the first of the next item
        uint32_t qn = q, cn = ch + 1;
        bool cross = false;
        if (cn == nch) {
            cn = 0;
            qn = q + 1;
            if (qn == nq) { qn = 0; cross = true; }
        }
        const bool live_n = !cross || it + 1 < i1;
        {
            const uint32_t b_n = (uint32_t)__shfl((int)(cross ? nb : cb), c.gbase + (int)qn, 64);
            const uint32_t e_n = (uint32_t)__shfl((int)(cross ? ne : ce), c.gbase + (int)qn, 64);
            const uint32_t ep = b_n + cn * CH;
            idx_turn<GW>(p.in_src, ep, live_n ? turn_fill(ep, e_n) : 0u, p.zrow, c.t, src_n);
        }
        const bool ends = ch + 1 == nch;                      // scalar: this turn completes the rows of group q
        const uint32_t lrow = cur.row + q * NS + (uint32_t)c.slot;
        const bool valid = q * NS + (uint32_t)c.slot < cur.count;
        double xo = 0.0;
        if (ends) xo = NT_LOAD(&p.x[(size_t)(valid ? lrow : cur.row) * GW + c.t]);
        double v[CH];
        gather_turn<GW>(c.T, src, c.t, c.gbase, v);
#pragma unroll
        for (int j = 0; j < CH; j++) acc += v[j];
        if (ends) {
            const uint32_t od = (uint32_t)__shfl((int)co, c.gbase + (int)q, 64);
            if (valid) finish_row<GW>(c, lrow, acc, xo, od);
            acc = 0.0;
        }
        q = qn;
        ch = cn;
        if (cross) {
            it++;
            cur = nxt;
            cb = nb; ce = ne; co = no;
            nq = (cur.count + NS - 1) / NS;
            nch = cur.nseg;
            nxt = work[it + 1];
            bounds(nxt, it + 1 < i1, nb, ne, no);
        }
    }
}

// V_DEG: an item = `count` rows of exactly D (= item.nseg) in-edges from `row` (their edges are contiguous from item.beg);
// a lane group takes R rows per turn, row r of the turn at slots r*DM .. r*DM+D-1 (DM = 16/R >= D)
template <int GW, int R, bool TS>
__device__ __forceinline__ void deg_rows(SweepCtx<GW, TS>& c, const WorkItem* __restrict__ work, uint32_t i0, uint32_t i1) {
    constexpr int NS = 64 / GW;
    constexpr int DM = CH / R;
    constexpr int IR = CH / GW;
    const PrParams& p = c.p;
    if (i0 >= i1) return;
    auto idx = [&](const WorkItem& w, bool live, uint32_t turn, uint32_t (&src)[IR]) __attribute__((always_inline)) {
        const uint32_t rb = (turn * NS + (uint32_t)c.slot) * R;           // first row (relative) of this lane group's turn
#pragma unroll
        for (int r = 0; r < IR; r++) {
            const uint32_t j = (uint32_t)(r * GW + c.t);
            const uint32_t rr = rb + j / DM, u = j % DM;
            const bool ok = live && u < w.nseg && rr < w.count;
            const uint32_t raw = NT_LOAD(&p.in_src[ok ? w.beg + rr * w.nseg + u : 0u]);
            src[r] = ok ? (raw & SRC_MASK) : p.zrow;
        }
    };
    WorkItem cur = work[i0], nxt = work[i0 + 1];
    uint32_t src_n[IR];
    idx(cur, true, 0, src_n);
    for (uint32_t it = i0; it < i1; it++) {
        const WorkItem nn = work[it + 2];
        const uint32_t row0 = cur.row, count = cur.count;
        const uint32_t turns = (count + NS * R - 1) / (NS * R);
        for (uint32_t i = 0; i < turns; i++) {
            uint32_t src[IR];
#pragma unroll
            for (int r = 0; r < IR; r++) src[r] = src_n[r];
            if (i + 1 < turns) idx(cur, true, i + 1, src_n);
            else idx(nxt, it + 1 < i1, 0, src_n);
            const uint32_t rb = (i * NS + (uint32_t)c.slot) * R;
            // old ranks and out-degrees of the R rows travel with the gathers
            double xo[R];
#pragma unroll
            for (int r = 0; r < R; r++) xo[r] = NT_LOAD(&p.x[(size_t)(row0 + (rb + r < count ? rb + r : 0u)) * GW + c.t]);
            const uint32_t myr = row0 + (rb + (uint32_t)c.t < count ? rb + (uint32_t)c.t : 0u);
            const uint32_t odv = (c.t < R && myr < p.sl_nd) ? NT_LOAD(&p.outdeg[myr]) : 1u;
            double v[CH];
            gather_turn<GW>(c.T, src, c.t, c.gbase, v);
#pragma unroll
            for (int r = 0; r < R; r++) {
                double y = 0.0;
#pragma unroll
                for (int u = 0; u < DM; u++) y += v[r * DM + u];
                const uint32_t od = (uint32_t)__shfl((int)odv, c.gbase + r, 64);
                if (rb + r < count) finish_row<GW>(c, row0 + rb + r, y, xo[r], od);
            }
        }
        cur = nxt;
        nxt = nn;
    }
}

#ifdef SS_PR_EXP_KINDMASK
#define SS_PR_CLASS_ON(c) ((p.kind_mask >> (8 + (c))) & 1u)
#else
#define SS_PR_CLASS_ON(c) true
#endif
#ifndef SS_PR_MINW
#define SS_PR_MINW 1
#endif
#ifdef SS_PR_WAVETIME
// variant build (tools/build_variant.sh wt -DSS_PR_WAVETIME): when every wave of the last sweep started and ran out of items
// (100 MHz realtime counter), printed by ss_pr_destroy — how level the deal is in TIME, not in modelled turns
__device__ unsigned long long g_pr_wt[65536][2];
#endif
template <int GW, bool TS>
__global__ __launch_bounds__(TPB, SS_PR_MINW) void k_pr_sweep(PrParams p) {
    constexpr int NS = 64 / GW;
    PrCtl* ctl = p.ctl;
    if (ctl->n_active == 0) return;   // every topic converged: the launch is a no-op
#ifdef SS_PR_WAVETIME
    unsigned long long wt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wt0));
#endif
    const int sweep = ctl->sweep;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    SweepCtx<GW, TS> c{p, p.tab_rd[sweep & 1], p.tab_wr[sweep & 1], 0.0, 0.0, false, lane % GW, lane - lane % GW, lane / GW, 0.0, 0.0};
    c.S = ctl->S[c.t];
    c.act = ctl->active[c.t] != 0;
    c.x0 = sweep == 0 ? p.x0[c.t] : 0.0;      // Q4: iteration 1 accumulates onto 1/n

    // This wave's items: work[off[k] .. off[k+1]) for class k.  The host dealt the items to the waves so that every wave
    // gets the same number of turns (ss_pr_create); one loop per class, so that the register allocator sees each
    // pipeline on its own instead of the union of all of them.
    const uint32_t* __restrict__ off = p.woff + (size_t)(blockIdx.x * WAVES + wave) * 8;
    // The ORDER in which a wave walks its classes matters more than anything tried on the deal (round 5: all 720 orders, 4 blocks per
    // CU, config 4): short rows first, long rows last — 2, 3, 5, 4, 0, 1 = rows of <= 2 in-edges, <= 4, edge-less, <= 8, long, mid —
    // 0.902-0.904 ms per sweep against 0.947 in the order the classes are numbered (worst order 0.961).  Round 4's "stagger" (the
    // resident blocks of a CU start at different positions of the order: option "pr.stagger", now off by default) had found a part
    // of this by accident — its best start vectors were the ones that began most blocks at the short rows —; on top of the best
    // orders no start vector gains anything (every vector of 1296 measured for the best four orders: the all-equal one wins).
    // The loop below walks the order; "pr.class_order" = six digits, "pr.stagger" as before.
    int rot = p.stagger_div ? (int)((blockIdx.x / p.stagger_div) % 6u) : 0;
    if (p.stagger_code) {                       // experiments ("pr.stagger" >= 10): round r starts at base-6 digit r of the code
        uint32_t cdv = p.stagger_code;
        for (uint32_t r = blockIdx.x / p.stagger_div; r > 0; r--) cdv /= 6u;
        rot = (int)(cdv % 6u);
    }
    for (int s6 = 0; s6 < 6; s6++) {
    const int cls = (int)((p.class_order >> (3 * ((s6 + rot) % 6))) & 7u);
    {
        // everything a class pipeline derives from the lane id is recomputed behind an opaque copy per round: hoisted out of this loop,
        // the per-lane invariants of all six pipelines were live at once (164 VGPRs = 3 waves per SIMD instead of 114 = 4)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        c.t = ln % GW;
        c.gbase = ln - ln % GW;
        c.slot = ln / GW;
    }
    switch (cls) {
    case 0: if (SS_PR_CLASS_ON(0)) long_rows<GW>(c, p.work, off[0], off[1], lane); break;
    case 1: if (SS_PR_CLASS_ON(1)) quad_rows<GW>(c, p.work, off[1], off[2]); break;
    case 2: if (SS_PR_CLASS_ON(2)) deg_rows<GW, 2>(c, p.work, off[2], off[3]); break;
    case 3: if (SS_PR_CLASS_ON(3)) deg_rows<GW, 4>(c, p.work, off[3], off[4]); break;
    case 4: if (SS_PR_CLASS_ON(4)) deg_rows<GW, 8>(c, p.work, off[4], off[5]); break;
    default:
    if (SS_PR_CLASS_ON(5))
    for (uint32_t item = off[5]; item < off[6]; item++) {
        const WorkItem w = p.work[item];
        // V_ZERO: non-dangling rows without in-edges: their rank is the shared value xz, only the next contribution
        // d*xz/outdeg has to be written (dangling ones need nothing at all); 16 rows per lane group and item at most
        const bool ts = TS && p.memb && ((p.ts_mask >> c.t) & 1u);
        const double xz_out = c.act ? (ts ? zero_row_rank_ts(p, sweep, c.S, p.x0[c.t], 0.0) : zero_row_rank(p, sweep, c.S, p.x0[c.t])) : ctl->xz[c.t];
        const double xz_inn = ts ? (c.act ? zero_row_rank_ts(p, sweep, c.S, p.x0[c.t], p.tin[c.t]) : ctl->xz_in[c.t]) : xz_out;
        uint32_t od[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t rr = (uint32_t)(i * NS + c.slot);
            od[i] = NT_LOAD(&p.outdeg[w.row + (rr < w.count ? rr : 0u)]);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t rr = (uint32_t)(i * NS + c.slot);
            if (rr < w.count) {
                const uint32_t lrow = w.row + rr;
                const double xz = ts && ((p.memb[lrow] >> c.t) & 1u) ? xz_inn : xz_out;
                const double cc = p.d * xz / (double)od[i];                      // pagerank.go:136
                NT_STORE(cc, &c.Tw[(size_t)lrow * GW + c.t]);
                c.csum += cc;                                                     // pagerank.go:137
            }
        }
    }
    break;
    }
    }
#ifdef SS_PR_WAVETIME
    {
        unsigned long long wt1;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wt1));
        const uint32_t wid = (blockIdx.x * WAVES + wave) & 65535u;
        if (lane == 0) { g_pr_wt[wid][0] = wt0; g_pr_wt[wid][1] = wt1; }
    }
#endif
    block_reduce_and_publish<GW>(p, c.dsum, c.csum, c.Tw, false);
}

// ---- the sweep for K <= 2, wave-owned items (round 4) ------------------------------------------------------------------------
// k_pr_sweep's lane group holds the GW topic values of ONE table row; with one or two topics that geometry either pads to eight
// (seven of eight lanes gather, add and DIVIDE for padding: config 2 was issue-bound at 7.6 % of the roofline) or shrinks the
// group to one or two lanes (nothing coalesces).  Here the lanes hold ROWS and EDGES instead: the table row is one double
// (K = 1) or one double2 (K = 2), a lane gathers it whole, and
//   V_DEG    rows of exactly D <= 8 in-edges (almost all rows of a power-law graph): one LANE per row — D index words, D gathers,
//            and every lane finishes a row of its own (the two float64 divisions of pagerank.go:117,136 run on 64 real rows);
//   V_QUAD   9 .. 256 in-edges: one row per 8-lane group, the lanes stride the row's edges, three-step butterfly at its end;
//   V_ROWW / V_SEG   long rows and 2048-edge pieces of the longest: the wave strides the edges, six-step butterfly;
//   V_ZERO   one lane per row.
// The items, their classes and the static deal to the waves are those of the 8-wide sweep (build_work with NS = 8).  Nothing is
// software-pipelined: a lane holds a handful of registers, so eight waves per SIMD hide the latency instead.
// Summation order: a row's in-edges are added in a fixed order that depends only on the row's class — deterministic, and
// within the last bits of the other kernels' orders (parity gate 1e-6; iteration counts as the oracle's).
#ifndef SS_PRN_MINW
#define SS_PRN_MINW 6
#endif
template <int KW>
struct NVec { double v[KW]; };
// PS (k_pr_multi_n: several sweeps inside one launch): the table row was written by another CU earlier in this launch, so it is read
// with L1-bypassing sc1 loads (and stored write-through, tab_store below) instead of relying on a kernel boundary
template <int KW, int PS = 0>
__device__ __forceinline__ NVec<KW> ntab(const double* __restrict__ T, uint32_t row) {
    NVec<KW> r;
    if constexpr (PS == 1) {
        const double* q = reinterpret_cast<const double*>(reinterpret_cast<const char*>(T) + (size_t)(row * (8u * KW)));
#pragma unroll
        for (int k = 0; k < KW; k++) r.v[k] = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if constexpr (KW == 1) {
        r.v[0] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(T) + (size_t)(row * 8u));
    } else {
        const double2 t = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(T) + (size_t)(row * 16u));
        r.v[0] = t.x;
        r.v[1] = t.y;
    }
    return r;
}
template <int PS>
__device__ __forceinline__ void tab_store(double v, double* q) {
    if constexpr (PS == 1) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else NT_STORE(v, q);
}
template <int KW, bool TS>
struct NCtx {
    const PrParams& p;
    const double* __restrict__ T;
    double* __restrict__ Tw;
    double S[KW], x0[KW], dsum[KW], csum[KW], tele[KW];
    bool act[KW];
    const double* __restrict__ xr;      // ranks before this sweep
    double* __restrict__ xw;            // ... and after it (the same array, except in the two-vector form: PrParams::x_alt)
};
// XS: also the row's RANK is stored write-through — the rows that are cut into pieces (V_SEG) are finished by whichever wave hands its
// piece in last, a different wave (and CU) from sweep to sweep, so inside k_pr_multi_n their ranks are handed from CU to CU like the table
template <int KW, bool TS, int PS = 0, bool XS = false>
__device__ __forceinline__ void finish_n(NCtx<KW, TS>& c, uint32_t lrow, const NVec<KW>& y, const NVec<KW>& xo, uint32_t od) {
    const PrParams& p = c.p;
#pragma unroll
    for (int k = 0; k < KW; k++) {
        const double yk = y.v[k] + c.x0[k];
        double tele = c.tele[k];
        if constexpr (TS) tele = teleport_of(p, lrow, k);
        double xn = (yk + tele) / c.S[k];                         // pagerank.go:117
        const size_t xi = (size_t)lrow * KW + k;
        if (c.act[k]) {
            if constexpr (XS) __hip_atomic_store(&c.xw[xi], xn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else NT_STORE(xn, &c.xw[xi]);
            c.dsum[k] += fabs(xn - xo.v[k]);                      // pagerank.go:118
        } else {
            xn = xo.v[k];                                         // converged topic: frozen
        }
        if (lrow < p.sl_nd) {                                     // non-dangling row: next sweep's contribution
            const double cc = p.d * xn / (double)od;              // pagerank.go:136
            tab_store<PS>(cc, &c.Tw[xi]);
            c.csum[k] += cc;                                      // pagerank.go:137
        }
    }
}
template <int KW>
__device__ __forceinline__ NVec<KW> load_x(const double* __restrict__ x, uint32_t lrow) {
    NVec<KW> r;
#pragma unroll
    for (int k = 0; k < KW; k++) r.v[k] = NT_LOAD(&x[(size_t)lrow * KW + k]);
    return r;
}

// rows of exactly D = w.nseg <= ND in-edges: lane l of pass r0 owns row r0 + l of the item
template <int KW, bool TS, int ND, int PS = 0>
__device__ __forceinline__ void deg_lane_rows(NCtx<KW, TS>& c, const WorkItem& w, int lane) {
    const PrParams& p = c.p;
    const uint32_t D = w.nseg;
    for (uint32_t r0 = 0; r0 < w.count; r0 += 64) {
        const uint32_t rr = r0 + (uint32_t)lane;
        const bool valid = rr < w.count;
        const uint32_t lrow = w.row + (valid ? rr : 0u);
        const uint32_t e0 = w.beg + (valid ? rr : 0u) * D;
        uint32_t src[ND];
#pragma unroll
        for (int u = 0; u < ND; u++) {
            const bool ok = valid && (uint32_t)u < D;
            const uint32_t raw = NT_LOAD(&p.in_src[ok ? e0 + (uint32_t)u : w.beg]);
            src[u] = ok ? (raw & SRC_MASK) : p.zrow;
        }
        const NVec<KW> xo = load_x<KW>(c.xr, lrow);
        const uint32_t od = lrow < p.sl_nd ? NT_LOAD(&p.outdeg[lrow]) : 1u;
        NVec<KW> v[ND];
#pragma unroll
        for (int u = 0; u < ND; u++) v[u] = ntab<KW, PS>(c.T, src[u]);
        NVec<KW> acc;
#pragma unroll
        for (int k = 0; k < KW; k++) acc.v[k] = 0.0;
#pragma unroll
        for (int u = 0; u < ND; u++)
#pragma unroll
            for (int k = 0; k < KW; k++) acc.v[k] += v[u].v[k];
        if (valid) finish_n<KW, TS, PS>(c, lrow, acc, xo, od);
    }
}

// one sweep of this block's waves; `sweep` = ctl->sweep as the caller read it.  PS: inside k_pr_multi_n (see ntab)
template <int KW, bool TS, int PS>
__device__ __forceinline__ void sweep_n_body(const PrParams& p, const int sweep, const double (&S_in)[KW], const int (&act_in)[KW]) {
    PrCtl* ctl = p.ctl;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (two-vector form: the vectors alternate between x and x_alt, so that the previous ones are still there for the topics' L1 changes)
    NCtx<KW, TS> c{p, p.tab_rd[sweep & 1], p.tab_wr[sweep & 1], {}, {}, {}, {}, {}, {},
                   (p.x_alt && (sweep & 1)) ? p.x_alt : p.x, p.x_alt ? ((sweep & 1) ? p.x : p.x_alt) : p.x};
#pragma unroll
    for (int k = 0; k < KW; k++) {
        c.tele[k] = p.tele_col ? p.tele_col[k] : p.teleport;          // (per column only in the two-vector form)
        c.S[k] = S_in[k];
        c.act[k] = act_in[k] != 0;
        c.x0[k] = sweep == 0 ? p.x0[k] : 0.0;                     // Q4: iteration 1 accumulates onto 1/n
        c.dsum[k] = 0.0;
        c.csum[k] = 0.0;
    }
    const uint32_t* __restrict__ off = p.woff + (size_t)(blockIdx.x * WAVES + wave) * 8;
    const uint32_t* __restrict__ in_src = p.in_src;

    // The four phases in the order `p.n_order` names (2 bits per position: 0 = long rows, 1 = mid rows, 2 = rows of <= 8 in-edges, 3 = edge-less
    // rows; option "pr.n_class_order"): as in k_pr_sweep the order matters more than the deal (round 5).
    for (int s4 = 0; s4 < 4; s4++) {
    switch ((p.n_order >> (2 * s4)) & 3u) {
    case 0: {
        // ---- V_SEG / V_ROWW: the wave strides the row's (piece's) edges, four gathers per lane in flight
        if (SS_PR_CLASS_ON(0))
        for (uint32_t it = off[0]; it < off[1]; it++) {
            const WorkItem w = p.work[it];
            const uint32_t lrow = w.row;
            NVec<KW> xo = load_x<KW>(c.xr, lrow);
            const uint32_t od = lrow < p.sl_nd ? NT_LOAD(&p.outdeg[lrow]) : 1u;
            NVec<KW> acc;
    #pragma unroll
            for (int k = 0; k < KW; k++) acc.v[k] = 0.0;
            // (the index words of the next 256 edges are requested before the gathers of the current ones are consumed)
            uint32_t src_n[4];
            auto idx256 = [&](uint32_t e, uint32_t (&src)[4]) __attribute__((always_inline)) {
    #pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t j = e + (uint32_t)(u * 64 + lane);
                    const uint32_t raw = NT_LOAD(&in_src[j < w.end ? j : w.beg]);
                    src[u] = j < w.end ? (raw & SRC_MASK) : p.zrow;
                }
            };
            idx256(w.beg, src_n);
            for (uint32_t e = w.beg; e < w.end; e += 256) {
                uint32_t src[4];
    #pragma unroll
                for (int u = 0; u < 4; u++) src[u] = src_n[u];
                idx256(e + 256, src_n);                                       // (past the end: four loads of the first edge, dropped)
                NVec<KW> v[4];
    #pragma unroll
                for (int u = 0; u < 4; u++) v[u] = ntab<KW, PS>(c.T, src[u]);
    #pragma unroll
                for (int u = 0; u < 4; u++)
    #pragma unroll
                    for (int k = 0; k < KW; k++) acc.v[k] += v[u].v[k];
            }
    #pragma unroll
            for (int k = 0; k < KW; k++)
    #pragma unroll
                for (int o = 1; o < 64; o <<= 1) acc.v[k] += __shfl_xor(acc.v[k], o, 64);
            if (w.kind == V_ROWW) {
                if (lane == 0) finish_n<KW, TS, PS>(c, lrow, acc, xo, od);
            } else {
                // several waves (of any blocks) share this row: publish the piece's sum write-through, drain, take the ticket; the
                // last to arrive adds the pieces in order with sc1 loads (no fences — see block_reduce_and_publish)
                const double mine = (KW == 2 && (lane & 1)) ? acc.v[KW - 1] : acc.v[0];
                if (lane < KW) __hip_atomic_store(&p.segpart[(size_t)(w.sbase + w.count) * KW + lane], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                unsigned prev = 0;
                if (lane == 0) prev = __hip_atomic_fetch_add(&p.rowticket[w.tix], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                prev = (unsigned)__builtin_amdgcn_readfirstlane((int)prev);
                if (prev == w.nseg - 1) {
                    if (lane == 0) {
                        __hip_atomic_store(&p.rowticket[w.tix], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        NVec<KW> ys;
    #pragma unroll
                        for (int k = 0; k < KW; k++) {
                            ys.v[k] = 0.0;
                            for (uint32_t q = 0; q < w.nseg; q++)
                                ys.v[k] += __hip_atomic_load(&p.segpart[(size_t)(w.sbase + q) * KW + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if constexpr (PS == 1) {
                            // (the rank this row got in the previous sweep may have been written by another CU: read it now, past its L1)
                            NVec<KW> xs;
    #pragma unroll
                            for (int k = 0; k < KW; k++) xs.v[k] = __hip_atomic_load(&c.xr[(size_t)lrow * KW + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            finish_n<KW, TS, PS, true>(c, lrow, ys, xs, od);
                        } else {
                            finish_n<KW, TS, PS>(c, lrow, ys, xo, od);
                        }
                    }
                }
            }
        }

    } break;
    case 1: {
        // ---- V_QUAD: one row per 8-lane group; the rows of an item are all nch 16-edge turns long
        {
            const int gl = lane & 7, grp = lane >> 3;
            if (SS_PR_CLASS_ON(1))
            for (uint32_t it = off[1]; it < off[2]; it++) {
                const WorkItem w = p.work[it];
                const uint32_t nq = (w.count + 7) / 8;
                for (uint32_t q = 0; q < nq; q++) {
                    const uint32_t rr = q * 8 + (uint32_t)grp;
                    const bool valid = rr < w.count;
                    const uint32_t lrow = w.row + (valid ? rr : 0u);
                    const uint32_t b = p.in_ptr[lrow], e_end = valid ? p.in_ptr[lrow + 1] : b;
                    NVec<KW> xo = load_x<KW>(c.xr, lrow);
                    const uint32_t od = lrow < p.sl_nd ? NT_LOAD(&p.outdeg[lrow]) : 1u;
                    NVec<KW> acc;
    #pragma unroll
                    for (int k = 0; k < KW; k++) acc.v[k] = 0.0;
                    uint32_t src_n[4];
                    auto idx32 = [&](uint32_t ch, uint32_t (&src)[4]) __attribute__((always_inline)) {
    #pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const uint32_t j = b + ch * 16u + (uint32_t)(u * 8 + gl);
                            const uint32_t raw = NT_LOAD(&in_src[j < e_end ? j : b]);
                            src[u] = j < e_end ? (raw & SRC_MASK) : p.zrow;
                        }
                    };
                    idx32(0, src_n);
                    for (uint32_t ch = 0; ch < w.nseg; ch += 2) {           // two turns (32 edge slots of the row) per trip: four gathers per lane
                        uint32_t src[4];
    #pragma unroll
                        for (int u = 0; u < 4; u++) src[u] = src_n[u];
                        idx32(ch + 2, src_n);                               // the next trip's index words travel with this trip's gathers
                        NVec<KW> v[4];
    #pragma unroll
                        for (int u = 0; u < 4; u++) v[u] = ntab<KW, PS>(c.T, src[u]);
    #pragma unroll
                        for (int u = 0; u < 4; u++)
    #pragma unroll
                            for (int k = 0; k < KW; k++) acc.v[k] += v[u].v[k];
                    }
    #pragma unroll
                    for (int k = 0; k < KW; k++)
    #pragma unroll
                        for (int o = 1; o < 8; o <<= 1) acc.v[k] += __shfl_xor(acc.v[k], o, 64);
                    if (valid && gl == 0) finish_n<KW, TS, PS>(c, lrow, acc, xo, od);
                }
            }
        }

    } break;
    case 2: {
        // ---- V_DEG (all three classes): rows of exactly D <= 8 in-edges, their edges contiguous from item.beg: one lane per row
        if (SS_PR_CLASS_ON(2))
        for (uint32_t it = off[2]; it < off[5]; it++) {
            const WorkItem w = p.work[it];
            if (w.nseg <= 2) deg_lane_rows<KW, TS, 2, PS>(c, w, lane);          // (wave-uniform: most rows of a power-law graph)
            else if (w.nseg <= 4) deg_lane_rows<KW, TS, 4, PS>(c, w, lane);
            else deg_lane_rows<KW, TS, 8, PS>(c, w, lane);
        }

    } break;
    default: {
        // ---- V_ZERO: non-dangling rows without in-edges: their rank is the shared value, only the next contribution is written
        if (SS_PR_CLASS_ON(5))
        for (uint32_t it = off[5]; it < off[6]; it++) {
            const WorkItem w = p.work[it];
            for (uint32_t r0 = 0; r0 < w.count; r0 += 64) {
                const uint32_t rr = r0 + (uint32_t)lane;
                if (rr >= w.count) continue;
                const uint32_t lrow = w.row + rr;
                const uint32_t od = NT_LOAD(&p.outdeg[lrow]);
    #pragma unroll
                for (int k = 0; k < KW; k++) {
                    const bool ts = TS && p.memb && ((p.ts_mask >> k) & 1u);
                    const double xz_out = c.act[k] ? (ts ? zero_row_rank_ts(p, sweep, c.S[k], p.x0[k], 0.0) : zero_row_rank_ts(p, sweep, c.S[k], p.x0[k], c.tele[k])) : ctl_ld<PS>(&ctl->xz[k]);
                    const double xz_inn = ts ? (c.act[k] ? zero_row_rank_ts(p, sweep, c.S[k], p.x0[k], p.tin[k]) : ctl_ld<PS>(&ctl->xz_in[k])) : xz_out;
                    const double xz = ts && ((p.memb[lrow] >> k) & 1u) ? xz_inn : xz_out;
                    const double cc = p.d * xz / (double)od;                      // pagerank.go:136
                    tab_store<PS>(cc, &c.Tw[(size_t)lrow * KW + k]);
                    c.csum[k] += cc;                                               // pagerank.go:137
                }
            }
        }

    } break;
    }
    }

    // block_reduce_and_publish<KW> expects lane l to hold a partial of topic l % KW
    double ds = c.dsum[0], cs = c.csum[0];
    if constexpr (KW == 2) {
        const double d0o = __shfl_xor(c.dsum[0], 1, 64), d1o = __shfl_xor(c.dsum[1], 1, 64);
        const double c0o = __shfl_xor(c.csum[0], 1, 64), c1o = __shfl_xor(c.csum[1], 1, 64);
        ds = (lane & 1) ? c.dsum[1] + d1o : c.dsum[0] + d0o;
        cs = (lane & 1) ? c.csum[1] + c1o : c.csum[0] + c0o;
    }
    block_reduce_and_publish<KW, PS>(p, ds, cs, c.Tw, false);
}

template <int KW, bool TS>
__global__ __launch_bounds__(TPB, SS_PRN_MINW) void k_pr_sweep_n(PrParams p) {
    const PrCtl* ctl = p.ctl;
    // (everything this wave needs of the control block is requested before the first of it is looked at: one scalar-load latency
    //  at the start of a sweep that is mostly fixed cost on a small graph, instead of two)
    const int n_active = ctl->n_active, sweep = ctl->sweep;
    double S_in[KW];
    int act_in[KW];
#pragma unroll
    for (int k = 0; k < KW; k++) { S_in[k] = ctl->S[k]; act_in[k] = ctl->active[k]; }
    if (n_active == 0) return;        // every topic converged: the launch is a no-op
    sweep_n_body<KW, TS, false>(p, sweep, S_in, act_in);
}

// ---- several sweeps in ONE launch (round 5: graphs whose sweep is all fixed cost) -------------------------------------------------
// BASELINE config 2 (2^20 nodes / 5M edges, one vector) sweeps in 54 us of which 33 us are an EMPTY grid's: launch, control-block
// reads, the two-level hand-in of the partial sums, kernel end.  pagerank.go:93-119 is a loop; here the loop runs inside the launch:
// every block walks its waves' items, hands its partial sums in exactly as k_pr_sweep_n does, and then WAITS until the last block to
// arrive has applied the stop rule and published the next sweep's number (finalize_ctl<true>: write-through stores, drained, the
// sweep counter last) — that wait is the grid-wide barrier between two sweeps.  Nothing is fenced: whatever one sweep writes for
// another CU to read in the next (the contribution table, the control block, the partial sums, the row pieces' tickets) is stored
// write-through (sc1) and read with L1-bypassing sc1 loads (MI355X_MICROARCH.md, hand-offs without fences; the ranks x are read and
// written by the same lane of the same wave in every sweep — the deal is static — and need nothing).  The arithmetic and its order
// are those of k_pr_sweep_n: ranks and iteration counts are bit-identical to one launch per sweep.
// Residency: the grid is sized by the host to HALF of what the occupancy query admits (ss_pr_create), so that it is resident whatever
// else runs; a wait that still ends without the counter moving (SPIN_MAX polls, seconds) sets ctl->stuck and every block leaves — the
// host reports SS_ERR_STATE instead of a hung device.
constexpr uint32_t MULTI_SPIN_MAX = 1u << 24;
// PSM 1: write-through / sc1 form (above).  PSM 2: plain stores and loads with an agent-scope release in front of every block's arrival
// and an acquire behind every block's wait (the table then stays in the XCD's L2 for the block's own gathers, as between launches).
template <int KW, int PSM>
__global__ __launch_bounds__(TPB, SS_PRN_MINW) void k_pr_multi_n(PrParams p, int n_steps) {
    __shared__ int s_go, s_na, s_sw, s_act[KW];
    __shared__ double s_S[KW];
    PrCtl* ctl = p.ctl;
    // ONE lane per block reads the control block (write-through data: sc1 loads that all land on one L2 channel — every lane of 4096 waves
    // reading it was 20k requests to that channel per sweep) and hands it to the block through LDS
    auto read_ctl = [&]() __attribute__((always_inline)) {
        s_na = ctl_ld<1>(&ctl->n_active) == 0 || ctl_ld<1>(&ctl->stuck) ? 0 : 1;
        s_sw = ctl_ld<1>(&ctl->sweep);
#pragma unroll
        for (int k = 0; k < KW; k++) { s_S[k] = ctl_ld<1>(&ctl->S[k]); s_act[k] = ctl_ld<1>(&ctl->active[k]); }
    };
    if (threadIdx.x == 0) read_ctl();
    __syncthreads();
    for (int s = 0; s < n_steps; s++) {
        // (every block reads the same control block: it only changes when ALL blocks have handed in the sweep)
        if (s_na == 0) return;                                    // every topic has stopped: the remaining sweeps are no-ops
        const int sweep = s_sw;
        double S_in[KW];
        int act_in[KW];
#pragma unroll
        for (int k = 0; k < KW; k++) { S_in[k] = s_S[k]; act_in[k] = s_act[k]; }
        __syncthreads();                                          // (everybody has its copy: lane 0 may rewrite the LDS words below)
        sweep_n_body<KW, false, PSM>(p, sweep, S_in, act_in);
        if (s + 1 == n_steps) return;                             // the kernel boundary is the last barrier
        if (threadIdx.x == 0) {
            uint32_t spins = 0;
            while (ctl_ld<1>(&ctl->sweep) == sweep && ++spins < MULTI_SPIN_MAX) __builtin_amdgcn_s_sleep(8);
            const bool ok = spins < MULTI_SPIN_MAX;
            if (!ok) ctl_st<1>(&ctl->stuck, 1u);
            s_go = ok ? 1 : 0;
            if (ok) read_ctl();
            if constexpr (PSM == 2) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // drops this CU's L1 lines: the other blocks' table rows and ranks
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (holds the barrier until the invalidate has completed)
            }
        }
        __syncthreads();
        if (!s_go) return;
    }
}

// k_pr_sweep's items: their in-edge ranges from the device's in_ptr (the host deals the items by their turn counts, which it
// knows from the sorted in-degrees; copying in_ptr itself to the host cost 14 of the 17 ms of ss_pr_create at 10M nodes)
__global__ void k_pr_item_ranges(WorkItem* __restrict__ work, uint32_t n_items, const uint32_t* __restrict__ in_ptr) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    WorkItem w = work[i];
    switch (w.kind) {
        case V_ROWW: w.beg = in_ptr[w.row]; w.end = in_ptr[w.row + 1]; break;
        case V_SEG:
            w.beg = in_ptr[w.row] + w.count * SEGW;
            w.end = min(in_ptr[w.row + 1], w.beg + SEGW);
            break;
        case V_QUAD:
        case V_DEG: w.beg = in_ptr[w.row]; w.end = in_ptr[w.row + w.count]; break;
        default: return;
    }
    work[i] = w;
}

// x0 = 1/n, first contributions and their sum (pagerank.go:103-106 + first :136-137)
template <int GW>
__global__ __launch_bounds__(TPB) void k_pr_begin(PrParams p) {
    const int lane = threadIdx.x & 63;
    const int t = lane % GW;
    const double x0 = p.x0[t];
    double* __restrict__ Tw = p.tab_wr[1];   // the table sweep 0 reads (tab_rd[0]) — see pr_make_params
    double csum = 0.0;
    const size_t n_el = ((size_t)p.sl_nd + p.sl_d) * GW;
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n_el; i += (size_t)gridDim.x * TPB) {
        const uint32_t lrow = (uint32_t)(i / GW);
        const bool real = lrow < p.sl_nd ? lrow < p.cnt_nd : (lrow - p.sl_nd) < p.cnt_d;
        p.x[i] = real ? x0 : 0.0;
        if (lrow < p.sl_nd) {
            double c = 0.0;
            if (real) c = p.d * x0 / (double)p.outdeg[lrow];
            if (p.world == 1 || lrow + 2 < p.sl_nd) Tw[i] = c;   // world>1: last two rows are the tail
            csum += c;
        }
    }
    block_reduce_and_publish<GW>(p, 0.0, csum, Tw, true);
}

// world>1: after the all-gather, combine the per-rank tails (rank order) and apply the stop rule
template <int GW>
__global__ void k_pr_finalize(PrParams p, const double* __restrict__ table, int is_begin) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!is_begin && p.ctl->n_active == 0) return;
    double dl[MAXK], cs[MAXK];
    for (int k = 0; k < MAXK; k++) { dl[k] = 0.0; cs[k] = 0.0; }
    for (int r = 0; r < p.world; r++) {
        const size_t base = ((size_t)r * p.sl_nd + (p.sl_nd - 2)) * GW;
        for (int k = 0; k < GW; k++) {
            cs[k] += table[base + k];
            dl[k] += table[base + GW + k];
        }
    }
    finalize_ctl(p, dl, cs, is_begin != 0);
}

template <int GW>
__global__ void k_pr_read(const double* __restrict__ x, const PrCtl* __restrict__ ctl, const uint32_t* __restrict__ old_id,
                          uint32_t sl_nd, uint32_t cnt_nd, uint32_t pos_nd, uint32_t sl_d, uint32_t cnt_d, uint32_t pos_d,
                          uint64_t id0_nd, uint64_t id0_d, int k_topics, uint64_t out_stride,
                          int by_original_id, uint32_t* __restrict__ ids_out, double* __restrict__ out,
                          const uint32_t* __restrict__ memb, uint32_t ts_mask) {
    const size_t n_rows = (size_t)cnt_nd + cnt_d;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const uint32_t lrow = i < cnt_nd ? (uint32_t)i : sl_nd + (uint32_t)(i - cnt_nd);
    const uint64_t iid = lrow < sl_nd ? id0_nd + lrow : id0_d + (lrow - sl_nd);
    const uint32_t orig = old_id[iid];
    const size_t o = by_original_id ? (size_t)orig : i;
    if (ids_out) ids_out[i] = orig;
    // rows without in-edges are not stored: they all hold ctl->xz
    const bool zero = lrow < sl_nd ? lrow >= pos_nd : (lrow - sl_nd) >= pos_d;
    const uint32_t mb = memb ? memb[lrow] & ts_mask : 0u;     // inside a topic's teleport set: the zero rows' other shared value
    for (int k = 0; k < k_topics; k++)
        out[(size_t)k * out_stride + o] = zero ? (((mb >> k) & 1u) ? ctl->xz_in[k] : ctl->xz[k]) : x[(size_t)lrow * GW + k];
}

// ---- topic-sensitive teleport (opt-in): sets -> per-row membership bits of this rank's rows ------------------------
__global__ void k_pr_memb(const uint64_t* __restrict__ set_ptr, const uint32_t* __restrict__ set_nodes, int k_topics, uint64_t n_nodes,
                          const uint32_t* __restrict__ new_id, uint64_t nd_int, uint32_t sl_nd, uint32_t sl_d, int rank,
                          uint32_t* __restrict__ memb, uint32_t* __restrict__ seen, uint32_t* __restrict__ err) {
    const uint64_t total = set_ptr[k_topics];
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        int k = 0;
        while (k + 1 < k_topics && set_ptr[k + 1] <= i) k++;
        const uint32_t v = set_nodes[i];
        if ((uint64_t)v >= n_nodes) { atomicOr(err, 1u); continue; }
        // |set_k| comes from set_ptr, membership is one bit per node: a node listed twice would get less than its share.
        // Checked over ALL nodes (not only this rank's rows) so that every rank of a sharded graph takes the same decision.
        if (atomicOr(&seen[v], 1u << k) & (1u << k)) { atomicOr(err, 2u); continue; }
        const uint64_t iid = new_id[v];
        uint32_t lrow;
        int owner;
        if (iid < nd_int) { owner = (int)(iid / sl_nd); lrow = (uint32_t)(iid % sl_nd); }
        else { owner = (int)((iid - nd_int) / sl_d); lrow = sl_nd + (uint32_t)((iid - nd_int) % sl_d); }
        if (owner == rank) atomicOr(&memb[lrow], 1u << k);
    }
}
// members among the rows without in-edges (their L1 change is counted, not streamed)
__global__ void k_pr_memb_zero_count(const uint32_t* __restrict__ memb, uint32_t sl_nd, uint32_t cnt_nd, uint32_t pos_nd, uint32_t cnt_d,
                                     uint32_t pos_d, int k_topics, unsigned long long* __restrict__ cnt) {
    const uint32_t z_nd = cnt_nd - pos_nd, n_zero = z_nd + (cnt_d - pos_d);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_zero; i += gridDim.x * blockDim.x) {
        const uint32_t lrow = i < z_nd ? pos_nd + i : sl_nd + pos_d + (i - z_nd);
        const uint32_t mb = memb[lrow];
        for (int k = 0; k < k_topics; k++)
            if ((mb >> k) & 1u) atomicAdd(&cnt[k], 1ull);
    }
}

// ---- diagnostic: the ceiling of the sweep's access pattern (ss_pr_probe) -----------------------------------------
// Gather-only pass over THIS graph's in-edge stream with the sweep's own load shape (one coalesced 64-byte index
// load per lane group, 16 independent whole-row gathers in flight) and nothing else: no row ends, no rank read or
// write, no contribution write, no reductions.  mode 0 = the real index stream, 1 = indices hashed to uniformly
// random rows (no hub reuse), 2 = consecutive rows (a streamed table).  Each lane group keeps one running sum and
// stores it once, so the loads cannot be dropped.
// POL (experiments with the cache policy of the gathers): 0 default, 1 all non-temporal, 2 all sc1 (agent-scope atomic
// load), 3 rows below `hot` default / others non-temporal, 4 rows below `hot` default / others sc1
template <int GW, int POL>
__global__ __launch_bounds__(TPB) void k_pr_probe(const double* __restrict__ T, const uint32_t* __restrict__ in_src, size_t n_edges,
                                                  uint32_t n_rows, int mode, double* __restrict__ sink, uint32_t hot) {
    if constexpr (GW >= 8) {
        constexpr int NSLOT = 64 / GW;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int t = lane % GW, slot = lane / GW, gbase = lane - t;
        const size_t groups = (size_t)gridDim.x * WAVES * NSLOT;
        const size_t gi = ((size_t)blockIdx.x * WAVES + wave) * NSLOT + slot;
        // contiguous span of 16-edge chunks per lane group
        const size_t n_chunks = n_edges / CH;
        const size_t per = (n_chunks + groups - 1) / groups;
        size_t c = min(gi * per, n_chunks);
        const size_t c_hi = min(c + per, n_chunks);
        double acc = 0.0;
        for (; c < c_hi; c++) {
            constexpr int R = CH / GW;
            uint32_t src[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const size_t e = c * CH + (size_t)(r * GW + t);
                uint32_t v = NT_LOAD(&in_src[e]) & SRC_MASK;
                if (mode == 1) v = (uint32_t)(((uint64_t)(v * 2654435761u + (uint32_t)e * 40503u) * n_rows) >> 32);
                if (mode == 2) v = (uint32_t)(e % n_rows);
                src[r] = v;
            }
            double v[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) {
                const uint32_t sj = (uint32_t)__shfl((int)src[j / GW], gbase + (j % GW), 64);
                const double* a = &T[(size_t)sj * GW + t];
                if constexpr (POL == 0) v[j] = *a;
                else if constexpr (POL == 1) v[j] = __builtin_nontemporal_load(a);
                else if constexpr (POL == 2) v[j] = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if constexpr (POL == 3) v[j] = sj < hot ? *a : __builtin_nontemporal_load(a);
                else v[j] = sj < hot ? *a : __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int j = 0; j < CH; j++) acc += v[j];
        }
        sink[((size_t)blockIdx.x * TPB + threadIdx.x)] = acc;
    }
}

}  // namespace

// ------------------------------------------------------------------------------

struct ss_pr {
    ss_graph* g = nullptr;
    int gw = 1;            // lane-group width = padded topic count
    ss::DevBuf<float> wire_send, wire_recv;   // option "pr.wire_f32": the contribution slice as float32 on the wire
    bool nwave = false;    // K <= 2 on the wave-item kernel k_pr_sweep_n (gw = K; the work items are those of the 8-wide sweep)
    int persist_mode = 1;  // 1: write-through hand-offs, 2: release / acquire fences around the wait
    bool persist = false;  // ... with ss_pr_step's sweeps inside ONE launch (k_pr_multi_n): small graphs, whose sweep is mostly fixed cost
    int k = 1;
    PrParams prm{};
    unsigned nblocks = 0;
    ss::DevBuf<double> x, tab0, tab1, send, partials, segpart, x0;
    ss::DevBuf<uint32_t> rowticket;
    ss::DevBuf<uint32_t> memb;          // topic-sensitive teleport (optional)
    ss::DevBuf<double> tin, nz_in;
    ss::DevBuf<WorkItem> work;
    ss::DevBuf<uint32_t> woff;          // k_pr_sweep: per-wave class offsets into work
    ss::DevBuf<PrCtl> ctl;
    bool begun = false;
    bool need_finalize = false;   // world>1: a begin/step is waiting for its exchange + finalize
    bool finalize_is_begin = false;
};

namespace {

int pick_gw(int k, uint64_t table_rows, bool force_narrow, bool narrow_wave) {
    // K = 3, 4 run the wave-item sweep padded to 8 topics (measured on the 10M/50M R-MAT at K=4: 0.77 ms against 0.94 ms
    // for the 4-wide block-item kernel).  K <= 2: the wave-item kernel for one or two topics, k_pr_sweep_n, unpadded (round 4).
    // Before it (option "pr.narrow_wave" = 0): padded to 8 when the padded table stays cache-resident (<= 64 MB of 64-byte
    // rows: the padding costs no HBM traffic then; 2^20 nodes / 5M edges, K=1: 0.074 ms), else the block-item kernel k_pr_step
    // (10M/50M, K=1: 0.55 ms against 0.79 ms padded).  "pr.force_narrow": always k_pr_step (tests reach it on small graphs).
    if (k >= 3 && k <= 8) return 8;
    if (k <= 2 && (narrow_wave || force_narrow)) return k;
    if (k <= 2 && table_rows * 64 <= (64ull << 20)) return 8;
    if (k <= 2) return k;
    return 16;
}

void build_work(const ss_graph* g, int gw, bool lane_rows, int64_t item_turns_default, std::vector<WorkItem>& items, uint32_t& nsegs, uint32_t& nmulti,
                uint32_t& seg_edges, uint32_t& pos_nd, uint32_t& pos_d, uint32_t (&vbeg)[7]) {
    const uint32_t NSLOT = 64 / gw;
    // gw < 8: rows above T_SEG in-edges get block(s) of their own, then wave-per-row / group-per-row classes.
    // gw >= 8: emit_v below.
    const uint32_t T_SEG = 32 * NSLOT;
    const uint32_t T_WAVE = 2 * NSLOT;
    seg_edges = 128 * NSLOT;
    nsegs = 0;
    nmulti = 0;
    // (scratch kept per thread across calls: freshly reserved vectors of a few MB cost more in page faults — 0.7 ms at 10M nodes — than
    //  the items cost to make)
    static thread_local std::vector<WorkItem> seg, rwg, wav, grp, zer;

    // gw >= 8: wave-owned items of k_pr_sweep.  deg is sorted descending.
    static thread_local std::vector<WorkItem> vseg, vroww, vquad, vdeg[3], vzero;
    for (auto* v : {&seg, &rwg, &wav, &grp, &zer, &vseg, &vroww, &vquad, &vdeg[0], &vdeg[1], &vdeg[2], &vzero}) v->clear();
    auto emit_v = [&](const ss_graph::SortedDegrees& deg, uint32_t row0, bool non_dangling, uint32_t& n_pos) {
        const uint32_t cnt = (uint32_t)deg.size();
        const uint32_t T_MULTI = 4096, T_DEG = 8;
        const uint32_t T_QUAD = (uint32_t)g->ctx->opt("pr.t_quad", lane_rows ? 256 : 128);   // (k_pr_sweep_n keeps 256)
        // turns per V_DEG item (a V_QUAD item: twice that): small graphs want finer items — with ~20 turns per wave in all, an
        // item of 16 leaves the deal nothing to balance ("pr.item_turns"; default from the graph's size, see ss_pr_create)
        const uint32_t item_turns = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(64, g->ctx->opt("pr.item_turns", item_turns_default)));
        // (rows and limits only move forward: two cursors over the degree runs instead of a bisection per item — the bisections were
        //  most of the 0.8 ms this took at config 4)
        size_t j_at = 0, j_lim = 0;
        const size_t n_run = deg.val.size();
        auto deg_at = [&](uint32_t r) -> uint32_t {              // in-degree of row r (r < cnt, never smaller than the last call's)
            while (deg.start[j_at + 1] <= r) j_at++;
            return deg.val[j_at];
        };
        auto end_above = [&](uint32_t lim) -> uint32_t {         // first row with in-degree <= lim (lim never larger than the last call's)
            while (j_lim < n_run && deg.val[j_lim] > lim) j_lim++;
            return deg.start[j_lim];
        };
        uint32_t r = 0;
        for (; r < cnt && deg_at(r) > T_MULTI; r++) {
            const uint32_t ns = (deg_at(r) + SEGW - 1) / SEGW;
            const uint32_t tix = nmulti++;
            for (uint32_t s = 0; s < ns; s++) vseg.push_back({V_SEG, row0 + r, s, ns, nsegs, tix});
            nsegs += ns;
        }
        {
            const uint32_t end = std::min(cnt, end_above(T_QUAD));
            for (; r < end; r++) vroww.push_back({V_ROWW, row0 + r, 0, 0, 0, 0});
        }
        // one row per lane group and turn; an item's rows all take nch = ceil(longest / 16) turns, at most gw row groups
        // and about 32 turns per item
        while (r < cnt && deg_at(r) > T_DEG) {
            const uint32_t nch = (deg_at(r) + CH - 1) / CH;
            const uint32_t max_groups = std::min<uint32_t>((uint32_t)gw, std::max<uint32_t>(1u, 2u * item_turns / nch));
            // rows of the same turn count nch: in-degree > (nch - 1) * CH (and > T_DEG)
            const uint32_t lim = std::max<uint32_t>(T_DEG, (nch - 1) * CH);
            const uint32_t end = end_above(lim);
            const uint32_t same = end > r ? end - r : 0u;
            const uint32_t rows = std::min<uint32_t>(same, max_groups * NSLOT);
            vquad.push_back({V_QUAD, row0 + r, rows, nch, 0, 0});
            r += rows;
        }
        // exact-degree runs
        while (r < cnt && deg_at(r) > 0) {
            const uint32_t D = deg_at(r);
            // deg is sorted descending and run-length encoded: the rows with exactly D in-edges end with r's run
            const uint32_t run = deg.start[j_at + 1] - r;
            const uint32_t R = D <= 2 ? 8 : D <= 4 ? 4 : 2;
            // (k_pr_sweep_n gives every LANE a row: whole waves of 64 rows per item there)
            const uint32_t per_item = lane_rows ? 64u * item_turns : NSLOT * R * item_turns;
            auto& out = vdeg[R == 2 ? 0 : R == 4 ? 1 : 2];
            for (uint32_t o = 0; o < run; o += per_item) out.push_back({V_DEG, row0 + r + o, std::min(per_item, run - o), D, 0, 0});
            r += run;
        }
        n_pos = r;
        const uint32_t ZERO_ROWS = 16 * NSLOT;
        if (non_dangling)
            for (uint32_t o = r; o < cnt; o += ZERO_ROWS) vzero.push_back({V_ZERO, row0 + o, std::min<uint32_t>(ZERO_ROWS, cnt - o), 0, 0, 0});
    };
    auto emit = [&](const ss_graph::SortedDegrees& deg, uint32_t row0, bool non_dangling, uint32_t& n_pos) {
        if (gw >= 8) { emit_v(deg, row0, non_dangling, n_pos); return; }
        const uint32_t cnt = (uint32_t)deg.size();
        // deg is sorted descending: class boundaries
        const uint32_t a = deg.first_at_most(T_SEG);
        const uint32_t c = std::max(a, deg.first_at_most(0));
        for (uint32_t r = 0; r < a; r++) {
            const uint32_t ns = (deg[r] + seg_edges - 1) / seg_edges;
            const uint32_t tix = ns > 1 ? nmulti++ : 0;
            for (uint32_t s = 0; s < ns; s++) seg.push_back({W_SEG, row0 + r, s, ns, nsegs, tix});
            nsegs += ns;
        }
        {
            const uint32_t b = std::min(c, std::max(a, deg.first_at_most(T_WAVE)));
            for (uint32_t r = a; r < b; r += WAVES) wav.push_back({W_WAVE, row0 + r, std::min<uint32_t>(WAVES, b - r), 0, 0, 0});
            const uint32_t GROUP_ROWS = WAVES * NSLOT * 4;   // 4 rows per lane group per block
            for (uint32_t r = b; r < c; r += GROUP_ROWS) grp.push_back({W_GROUP, row0 + r, std::min<uint32_t>(GROUP_ROWS, c - r), 0, 0, 0});
        }
        n_pos = c;
        // rows without in-edges all share one rank (zero_row_rank): only the non-dangling ones have work
        // (their next contribution); 8 elements per thread
        const uint32_t ZERO_ROWS = (TPB * 8) / gw;
        if (non_dangling)
            for (uint32_t r = c; r < cnt; r += ZERO_ROWS) zer.push_back({W_ZERO, row0 + r, std::min<uint32_t>(ZERO_ROWS, cnt - r), 0, 0, 0});
    };
    emit(g->h_indeg_nd, 0, true, pos_nd);
    const size_t vquad_split = vquad.size();            // the non-dangling rows' groups (falling length), then the dangling rows'
    emit(g->h_indeg_d, g->sl_nd, false, pos_d);
    items.clear();
    items.reserve(seg.size() + rwg.size() + wav.size() + grp.size() + zer.size());
    // heavy work first
    items.insert(items.end(), seg.begin(), seg.end());
    items.insert(items.end(), rwg.begin(), rwg.end());
    items.insert(items.end(), wav.begin(), wav.end());
    items.insert(items.end(), grp.begin(), grp.end());
    items.insert(items.end(), zer.begin(), zer.end());
    // k_pr_sweep: longest first (pieces of the hubs, whole long rows, then the row groups by falling length)
    vbeg[0] = (uint32_t)items.size();
    items.insert(items.end(), vseg.begin(), vseg.end());
    items.insert(items.end(), vroww.begin(), vroww.end());
    vbeg[1] = (uint32_t)items.size();
    // row groups by falling length (the dangling class was appended after the non-dangling one)
    // (each of the two classes is in falling length already: one stable merge, not a sort)
    std::inplace_merge(vquad.begin(), vquad.begin() + (ptrdiff_t)vquad_split, vquad.end(),
                       [](const WorkItem& a, const WorkItem& b) { return a.nseg > b.nseg; });
    items.insert(items.end(), vquad.begin(), vquad.end());
    for (int k = 0; k < 3; k++) {
        vbeg[2 + k] = (uint32_t)items.size();
        items.insert(items.end(), vdeg[k].begin(), vdeg[k].end());
    }
    vbeg[5] = (uint32_t)items.size();
    items.insert(items.end(), vzero.begin(), vzero.end());
    vbeg[6] = (uint32_t)items.size();
}

template <int GW>
void launch_step(ss_pr* pr, hipStream_t st) {
    if constexpr (GW <= 2) {
        if (pr->nwave) {
            if (pr->prm.memb) hipLaunchKernelGGL((k_pr_sweep_n<GW, true>), dim3(pr->nblocks), dim3(TPB), 0, st, pr->prm);
            else hipLaunchKernelGGL((k_pr_sweep_n<GW, false>), dim3(pr->nblocks), dim3(TPB), 0, st, pr->prm);
            return;
        }
    }
    if constexpr (GW >= 8) {
        if (pr->prm.memb) hipLaunchKernelGGL((k_pr_sweep<GW, true>), dim3(pr->nblocks), dim3(TPB), 0, st, pr->prm);
        else hipLaunchKernelGGL((k_pr_sweep<GW, false>), dim3(pr->nblocks), dim3(TPB), 0, st, pr->prm);
    }
    else hipLaunchKernelGGL(k_pr_step<GW>, dim3(pr->nblocks), dim3(TPB), 0, st, pr->prm);
}
template <int GW>
void launch_multi(ss_pr* pr, hipStream_t st, int n_steps) {
    if constexpr (GW <= 2) {
        if (pr->persist_mode == 2) hipLaunchKernelGGL((k_pr_multi_n<GW, 2>), dim3(pr->nblocks), dim3(TPB), 0, st, pr->prm, n_steps);
        else hipLaunchKernelGGL((k_pr_multi_n<GW, 1>), dim3(pr->nblocks), dim3(TPB), 0, st, pr->prm, n_steps);
    }
}
template <int GW>
void sweep_occupancy(int* blocks_per_cu) {
    if constexpr (GW >= 8) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k_pr_sweep<GW, false>, TPB, 0);
    else *blocks_per_cu = 8;
}
template <int GW>
void launch_begin(ss_pr* pr, hipStream_t st, unsigned nb) {
    hipLaunchKernelGGL(k_pr_begin<GW>, dim3(nb), dim3(TPB), 0, st, pr->prm);
}
template <int GW>
void launch_finalize(ss_pr* pr, hipStream_t st, int is_begin) {
    hipLaunchKernelGGL(k_pr_finalize<GW>, dim3(1), dim3(64), 0, st, pr->prm, (const double*)pr->tab0.p, is_begin);
}
// The same by ORIGINAL id on an unsharded graph (ss_pr_read: rank_out[k][v]): one thread per original node v.  k_pr_read above
// walks the rows in internal order and scatters 8-byte values into K topic planes at random original ids (every store a partial
// line: 3.0 ms for the 1.28 GB of config 4, 3.05 GB fetched); here the WRITES are the coalesced side — for every topic the 64
// lanes of a wave store 64 consecutive doubles — and the reads gather whole rows (GW doubles, one 128-byte line at GW = 16).
template <int GW>
__global__ __launch_bounds__(TPB) void k_pr_read_orig(const double* __restrict__ x, const PrCtl* __restrict__ ctl, const uint32_t* __restrict__ new_id,
                                                      uint64_t n, uint32_t sl_nd, uint32_t pos_nd, uint32_t pos_d, int k_topics,
                                                      double* __restrict__ out, const uint32_t* __restrict__ memb, uint32_t ts_mask) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint32_t lrow = new_id[v];                  // world == 1: internal id == local row
    const bool zero = lrow < sl_nd ? lrow >= pos_nd : (lrow - sl_nd) >= pos_d;
    double r[GW];
    if (!zero) {
        if constexpr (GW >= 2) {
            const double2* const row = reinterpret_cast<const double2*>(x + (size_t)lrow * GW);
#pragma unroll
            for (int j = 0; j < GW / 2; j++) { const double2 t = row[j]; r[2 * j] = t.x; r[2 * j + 1] = t.y; }
        } else {
            r[0] = x[lrow];
        }
    } else {
        const uint32_t mb = memb ? memb[lrow] & ts_mask : 0u;
#pragma unroll
        for (int k = 0; k < GW; k++) r[k] = ((mb >> k) & 1u) ? ctl->xz_in[k] : ctl->xz[k];
    }
#pragma unroll
    for (int k = 0; k < GW; k++)
        if (k < k_topics) out[(size_t)k * n + v] = r[k];
}


// ---- two-vector form: per-topic L1 change, stop rule, write-out ("pr.affine"; see AffCtl) ---------------------------------------
// x holds (p, q) of the rows WITH in-edges; the edge-less rows share one (p, q) per class position (ctl->xz).  A topic's rank
// of a row is (p*u + q) / (r*u + s).
constexpr int AFF_KC = 16;        // topics a thread accumulates per pass over the rows
constexpr unsigned AFF_NB = 512;  // blocks of k_aff_delta (their partial sums are added in a fixed order)
__global__ __launch_bounds__(TPB) void k_aff_delta(const double2* __restrict__ xp, const double2* __restrict__ xn, const AffCtl* __restrict__ a,
                                                   uint32_t sl_nd, uint32_t pos_nd, uint32_t pos_d, double* __restrict__ partials) {
    __shared__ double red[WAVES][AFF_KC];
    if (a->n_active == 0) return;                                     // every topic has stopped: the enqueued iterations are no-ops
    const int k_real = a->k_real;
    const double r_old = a->r_prev, r_new = a->r_x, s_old = a->s_prev, s_new = a->s_x;
    const uint32_t n_rows = pos_nd + pos_d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k0 = 0; k0 < k_real; k0 += AFF_KC) {
        double u[AFF_KC], d_old[AFF_KC], d_new[AFF_KC], acc[AFF_KC];
        bool any = false;
#pragma unroll
        for (int j = 0; j < AFF_KC; j++) {
            const bool on = k0 + j < k_real && a->active[k0 + j] != 0;
            any = any || on;
            u[j] = on ? a->u[k0 + j] : 0.0;
            d_old[j] = on ? r_old * u[j] + s_old : 1.0;
            d_new[j] = on ? r_new * u[j] + s_new : 1.0;
            acc[j] = 0.0;
        }
        if (any) {
            for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < n_rows; i += gridDim.x * TPB) {
                const uint32_t lrow = i < pos_nd ? i : sl_nd + (i - pos_nd);
                const double2 o = xp[lrow], n = xn[lrow];
#pragma unroll
                for (int j = 0; j < AFF_KC; j++) acc[j] += fabs((n.x * u[j] + n.y) / d_new[j] - (o.x * u[j] + o.y) / d_old[j]);   // pagerank.go:118
            }
        }
#pragma unroll
        for (int j = 0; j < AFF_KC; j++) {
            double v = acc[j];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) red[wave][j] = v;
        }
        __syncthreads();
        if (threadIdx.x < AFF_KC && k0 + (int)threadIdx.x < k_real) {
            double v = 0.0;
            for (int w = 0; w < WAVES; w++) v += red[w][threadIdx.x];
            partials[(size_t)blockIdx.x * AFF_MAXK + k0 + threadIdx.x] = v;
        }
        __syncthreads();
    }
}
// one block: the topics' L1 changes (block partials in block order + the edge-less rows, which all hold one value), pagerank.go:93
// and the max_iter cut, per topic
// `stride`: doubles between two blocks' (ranks') rows of sums — AFF_MAXK for a partials / gather buffer, sl_nd * 2 when the ranks' sums are
// read where they arrived: in the spare tail rows of the all-gathered contribution table
__global__ __launch_bounds__(AFF_MAXK) void k_aff_ctl(AffCtl* __restrict__ a, PrCtl* __restrict__ ctl, const double* __restrict__ partials, unsigned nb,
                                                      double n_zero, double eps, int max_iter, size_t stride) {
    __shared__ int s_na, s_nj;
    __shared__ double s_part[16][AFF_KC];
    __shared__ double s_dl[AFF_MAXK];
    const int k = threadIdx.x;
    if (a->n_active == 0) {
        if (k == 0) a->n_just = 0;
        return;
    }
    if (k == 0) { s_na = 0; s_nj = 0; }
    const int k_real = a->k_real;
    // the blocks' partial sums, 16 topics at a time: thread (part, j) adds blocks part, part + 16, ... of topic j in block order, thread
    // j then adds the 16 parts in order — fixed order, and nobody adds 256 numbers alone
    for (int k0 = 0; k0 < k_real; k0 += AFF_KC) {
        const int j = k & (AFF_KC - 1), part = k >> 4;
        double v = 0.0;
        if (k0 + j < k_real)
            for (unsigned b = (unsigned)part; b < nb; b += 16) v += partials[(size_t)b * stride + k0 + j];
        s_part[part][j] = v;
        __syncthreads();
        if (k < AFF_KC) {
            double t = 0.0;
            for (int q = 0; q < 16; q++) t += s_part[q][k];
            s_dl[k0 + k] = t;
        }
        __syncthreads();
    }
    const int it = a->it + 1;
    if (k < k_real) {
        a->just[k] = 0;
        if (a->active[k]) {
            double dl = s_dl[k];
            const double u = a->u[k];
            const double z_new = (ctl->xz[0] * u + ctl->xz[1]) / (a->r_x * u + a->s_x);
            const double z_old = (a->xz_prev[0] * u + a->xz_prev[1]) / (a->r_prev * u + a->s_prev);
            dl += n_zero * fabs(z_new - z_old);
            a->delta[k] = dl;
            a->iters[k] = it;
            bool cont = dl > eps;                                       // pagerank.go:93
            if (max_iter > 0 && it >= max_iter) cont = false;
            if (cont) atomicAdd(&s_na, 1);
            else { a->active[k] = 0; a->just[k] = 1; atomicAdd(&s_nj, 1); }
        }
    }
    __syncthreads();
    if (k == 0) {
        a->it = it;
        a->n_active = s_na;
        a->n_just = s_nj;
        if (s_na == 0) ctl->n_active = 0;                             // the sweeps that are already enqueued return at once
    }
}
// ranks of the topics that have just stopped, by ORIGINAL id (rank_out[k][v]): one thread per node, its row's (p, q) once
__global__ __launch_bounds__(TPB) void k_aff_emit(const double2* __restrict__ x, const PrCtl* __restrict__ ctl, const AffCtl* __restrict__ a,
                                                  const uint32_t* __restrict__ new_id, uint64_t n, uint32_t sl_nd, uint32_t pos_nd, uint32_t pos_d,
                                                  double* __restrict__ out) {
    if (a->n_just == 0) return;
    const double r = a->r_x, sx = a->s_x;
    const int k_real = a->k_real;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t lrow = new_id[v];
        const bool zero = lrow < sl_nd ? lrow >= pos_nd : (lrow - sl_nd) >= pos_d;
        double2 pq;
        if (zero) pq = make_double2(ctl->xz[0], ctl->xz[1]);
        else pq = x[lrow];
        for (int k = 0; k < k_real; k++)
            if (a->just[k]) {
                const double u = a->u[k];
                out[(size_t)k * n + v] = (pq.x * u + pq.y) / (r * u + sx);
            }
    }
}

// sharded two-vector form: this rank's per-topic sums (its blocks' partials in a fixed order + its own edge-less rows); the ranks'
// sums are all-gathered and k_aff_ctl adds them in rank order (nb = world, n_zero = 0), so every rank decides alike
__global__ __launch_bounds__(AFF_MAXK) void k_aff_local(const AffCtl* __restrict__ a, const PrCtl* __restrict__ ctl, const double* __restrict__ partials, unsigned nb,
                                                        double n_zero, double* __restrict__ loc) {
    __shared__ double s_part[16][AFF_KC];
    const int k = threadIdx.x;
    const int k_real = a->k_real;
    if (a->n_active == 0) return;
    for (int k0 = 0; k0 < k_real; k0 += AFF_KC) {
        const int j = k & (AFF_KC - 1), part = k >> 4;
        double v = 0.0;
        if (k0 + j < k_real)
            for (unsigned b = (unsigned)part; b < nb; b += 16) v += partials[(size_t)b * AFF_MAXK + k0 + j];
        s_part[part][j] = v;
        __syncthreads();
        if (k < AFF_KC && k0 + k < k_real) {
            double t = 0.0;
            for (int q = 0; q < 16; q++) t += s_part[q][k];
            const int kk = k0 + k;
            if (a->active[kk]) {
                const double u = a->u[kk];
                const double z_new = (ctl->xz[0] * u + ctl->xz[1]) / (a->r_x * u + a->s_x);
                const double z_old = (a->xz_prev[0] * u + a->xz_prev[1]) / (a->r_prev * u + a->s_prev);
                t += n_zero * fabs(z_new - z_old);
            }
            loc[kk] = t;
        }
        __syncthreads();
    }
}
// ... and the ranks of the topics that have just stopped, for this rank's rows in local order (out[k][row], as ss_pr_read_local)
__global__ __launch_bounds__(TPB) void k_aff_emit_local(const double2* __restrict__ x, const PrCtl* __restrict__ ctl, const AffCtl* __restrict__ a,
                                                        uint32_t sl_nd, uint32_t cnt_nd, uint32_t cnt_d, uint32_t pos_nd, uint32_t pos_d,
                                                        double* __restrict__ out, int lag) {
    if (a->n_just == 0) return;
    // lag: the decision arrives one exchange late (its sums rode on the next iteration's all-gather): `x` are the vectors BEFORE the last
    // sweep, and their r, s and edge-less rows are the control block's "previous" ones
    const double r = lag ? a->r_prev : a->r_x, sx = lag ? a->s_prev : a->s_x;
    const double xz0 = lag ? a->xz_prev[0] : ctl->xz[0], xz1 = lag ? a->xz_prev[1] : ctl->xz[1];
    const int k_real = a->k_real;
    const uint32_t n_rows = cnt_nd + cnt_d;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += gridDim.x * blockDim.x) {
        const uint32_t lrow = i < cnt_nd ? i : sl_nd + (i - cnt_nd);
        const bool zero = lrow < sl_nd ? lrow >= pos_nd : (lrow - sl_nd) >= pos_d;
        double2 pq;
        if (zero) pq = make_double2(xz0, xz1);
        else pq = x[lrow];
        for (int k = 0; k < k_real; k++)
            if (a->just[k]) {
                const double u = a->u[k];
                out[(size_t)k * n_rows + i] = (pq.x * u + pq.y) / (r * u + sx);
            }
    }
}

template <int GW>
void launch_read(ss_pr* pr, hipStream_t st, int by_orig, uint64_t stride, uint32_t* ids, double* out) {
    const ss_graph* g = pr->g;
    const size_t n_rows = (size_t)g->cnt_nd + g->cnt_d;
    if (!n_rows) return;
    if (by_orig && g->world == 1 && !ids && stride == g->n) {
        hipLaunchKernelGGL(k_pr_read_orig<GW>, dim3(ss::div_up(g->n, TPB)), dim3(TPB), 0, st, (const double*)pr->x.p, (const PrCtl*)pr->ctl.p,
                           (const uint32_t*)g->new_id.p, g->n, g->sl_nd, pr->prm.pos_nd, pr->prm.pos_d, pr->k, out, pr->prm.memb, pr->prm.ts_mask);
        return;
    }
    hipLaunchKernelGGL(k_pr_read<GW>, dim3(ss::div_up(n_rows, TPB)), dim3(TPB), 0, st, (const double*)pr->x.p,
                       (const PrCtl*)pr->ctl.p, (const uint32_t*)g->old_id.p, g->sl_nd, g->cnt_nd, pr->prm.pos_nd, g->sl_d, g->cnt_d,
                       pr->prm.pos_d,
                       (uint64_t)g->rank * g->sl_nd, g->nd_int + (uint64_t)g->rank * g->sl_d, pr->k, stride, by_orig, ids, out,
                       pr->prm.memb, pr->prm.ts_mask);
}

// ---- float32 on the wire (option "pr.wire_f32", opt-in; bench.py reports it under `decompositions` only) --------------------
// The doc-range-sharded sweep is bound by the bytes a rank receives per sweep over its point-to-point xGMI links (VERDICT r3:
// 52 MB per link at K = 16 whatever the world size).  With this option a rank's contribution slice travels as float32 — half the
// bytes — and is widened again on arrival; the two tail rows of the slice (the rank's partial sums: normaliser and L1 change,
// which every rank must combine identically and which decide the stop rule) travel as (hi, lo) float pairs, i.e. to ~2^-48.
// Every rank decodes the SAME gathered floats, its own slice included, so the ranks stay in lockstep.  The ranks differ from the
// float64 exchange by the rounding of the contributions (2^-24 relative each, averaging out in the sums): inside the 1e-6 parity
// gate, not the reference's float64 arithmetic — which is why it is not the default.  Arithmetic on the device stays float64.
__global__ void k_wire_pack(const double* __restrict__ send, uint32_t n_body, uint32_t n_tail, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_body) out[i] = (float)send[i];
    else if (i < n_body + n_tail) {
        const double d = send[i];
        const float hi = (float)d;
        out[n_body + 2 * (i - n_body)] = hi;
        out[n_body + 2 * (i - n_body) + 1] = (float)(d - (double)hi);
    }
}
// in: [world][n_body + 2 n_tail] floats -> table [world][n_body + n_tail] doubles
__global__ void k_wire_unpack(const float* __restrict__ in, uint32_t world, uint32_t n_body, uint32_t n_tail, double* __restrict__ table) {
    const uint64_t per_out = (uint64_t)n_body + n_tail, per_in = (uint64_t)n_body + 2ull * n_tail;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_out * world) return;
    const uint64_t r = i / per_out, j = i % per_out;
    const float* src = in + r * per_in;
    table[i] = j < n_body ? (double)src[j] : (double)src[n_body + 2 * (j - n_body)] + (double)src[n_body + 2 * (j - n_body) + 1];
}
// floats a rank sends: the slice's body rows + its tail rows (the two sums rows and the TAIL_SUM_ROWS in front of them) as pairs
constexpr uint32_t WIRE_TAIL_ROWS = 2u + TAIL_SUM_ROWS;
size_t wire_floats(const ss_pr* pr) { return (size_t)pr->g->sl_nd * pr->gw + (size_t)WIRE_TAIL_ROWS * (size_t)pr->gw; }
hipError_t wire_alloc(ss_pr* pr) {
    if (pr->wire_send.p) return hipSuccess;
    hipError_t e = pr->wire_send.alloc(wire_floats(pr));
    if (e == hipSuccess) e = pr->wire_recv.alloc(wire_floats(pr) * (size_t)pr->g->world);
    return e;
}
void wire_pack(ss_pr* pr, hipStream_t st) {
    const uint32_t n_tail = WIRE_TAIL_ROWS * (uint32_t)pr->gw, n_body = pr->g->sl_nd * (uint32_t)pr->gw - n_tail;
    hipLaunchKernelGGL(k_wire_pack, dim3(ss::div_up((size_t)n_body + n_tail, TPB)), dim3(TPB), 0, st, (const double*)pr->send.p, n_body, n_tail, pr->wire_send.p);
}
void wire_unpack(ss_pr* pr, hipStream_t st) {
    const uint32_t n_tail = WIRE_TAIL_ROWS * (uint32_t)pr->gw, n_body = pr->g->sl_nd * (uint32_t)pr->gw - n_tail;
    hipLaunchKernelGGL(k_wire_unpack, dim3(ss::div_up(((size_t)n_body + n_tail) * pr->g->world, TPB)), dim3(TPB), 0, st, (const float*)pr->wire_recv.p,
                       (uint32_t)pr->g->world, n_body, n_tail, pr->tab0.p);
}

#define SS_GW_DISPATCH(gw, fn, ...)          \
    switch (gw) {                            \
        case 1: fn<1>(__VA_ARGS__); break;   \
        case 2: fn<2>(__VA_ARGS__); break;   \
        case 8: fn<8>(__VA_ARGS__); break;   \
        default: fn<16>(__VA_ARGS__); break; \
    }

}  // namespace

extern "C" {

int32_t ss_pr_create(ss_graph* g, double damping, double eps, int32_t max_iter, int32_t k_topics,
                     const int32_t* n_topic, ss_pr** out) {
    if (!g) return SS_ERR_INVALID;
    ss_ctx* ctx = g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!out) return ctx->fail(SS_ERR_INVALID, "ss_pr_create: out is NULL");
    *out = nullptr;
    if (k_topics < 1 || !n_topic) return ctx->fail(SS_ERR_INVALID, "ss_pr_create: k_topics < 1 or n_topic NULL");
    if (k_topics > MAXK) return ctx->fail(SS_ERR_UNSUPPORTED, "ss_pr_create: k_topics %d > %d per state (ss_pagerank_run splits larger K)", k_topics, MAXK);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    ss_pr* pr = new (std::nothrow) ss_pr();
    if (!pr) return ctx->fail(SS_ERR_OOM, "ss_pr_create: host OOM");
    std::unique_ptr<ss_pr> guard(pr);
    pr->g = g;
    pr->k = k_topics;
    const bool force_narrow = ctx->opt("pr.force_narrow", 0) != 0;
    pr->gw = pick_gw(k_topics, g->nd_int, force_narrow, ctx->opt("pr.narrow_wave", 1) != 0);
    pr->nwave = pr->gw <= 2 && !force_narrow && ctx->opt("pr.narrow_wave", 1) != 0;
    const int GW = pr->gw;
    const bool vitems = GW >= 8 || pr->nwave;          // wave-owned items (k_pr_sweep / k_pr_sweep_n); otherwise k_pr_step's block items
    const int GI = pr->nwave ? 8 : GW;                 // lane-group width the ITEMS are cut for
    const size_t n_local = g->n_local();

    const bool trace = ctx->opt("pr.trace", 0) != 0;
    auto t_now = [] { return std::chrono::steady_clock::now(); };
    auto t_ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    // the large tables first: the device zeroes them (gigabytes at config 4) while the host builds and deals the work items below
    if (((uint64_t)g->nd_int + 1) * GW * 8 >= (1ull << 32))
        return ctx->fail(SS_ERR_UNSUPPORTED, "ss_pr_create: contribution table of %llu rows x %d topics exceeds 4 GiB (shard the graph over more ranks)",
                         (unsigned long long)g->nd_int, GW);
    SS_HIP(ctx, pr->x.alloc_streaming(n_local * GW));
    // + the all-zero row k_pr_sweep's unused slots gather from (never written: the exchange and the sweeps stop at nd_int)
    SS_HIP(ctx, pr->tab0.alloc(((size_t)g->nd_int + 1) * GW));
    SS_HIP(ctx, hipMemsetAsync(pr->tab0.p, 0, std::max<size_t>(pr->tab0.bytes(), 8), st));
    if (g->world == 1) {
        SS_HIP(ctx, pr->tab1.alloc(((size_t)g->nd_int + 1) * GW));
        SS_HIP(ctx, hipMemsetAsync(pr->tab1.p, 0, std::max<size_t>(pr->tab1.bytes(), 8), st));
    } else {
        SS_HIP(ctx, pr->send.alloc((size_t)g->sl_nd * GW));
        SS_HIP(ctx, hipMemsetAsync(pr->send.p, 0, std::max<size_t>(pr->send.bytes(), 8), st));
    }
    const auto tc0 = t_now();
    static thread_local std::vector<WorkItem> items, dealt;       // (scratch kept per thread across calls, see build_work)
    static thread_local std::vector<double> cost;
    static thread_local std::vector<uint32_t> owner;
    uint32_t nsegs = 0, nmulti = 0, seg_edges = 0, pos_nd = 0, pos_d = 0;
    uint32_t vbeg[7] = {0};
    // item granularity (measured, sweep ms at 2 / 4 / 8 / 16 / 32 turns per V_DEG item): 2^20 nodes, 5M edges, K=1: 0.078 / 0.077 / 0.096 /
    // 0.102 / 0.158; 10M nodes, 50M edges, K=16: 0.973 / 0.968 / 0.968 / 0.988 / 1.013 — a small graph gives every wave ~20 turns in all,
    // and the deal can only balance what the items let it; the large one pays for more items in the deal itself (host time)
    build_work(g, GI, pr->nwave, pr->nwave ? (n_local <= ((size_t)4 << 20) ? 1 : 4) : (n_local <= ((size_t)4 << 20) ? 4 : 8), items, nsegs, nmulti, seg_edges, pos_nd, pos_d, vbeg);
    const auto tc1 = t_now();
    if (items.empty()) items.push_back({W_ZERO, 0, 0, 0, 0, 0});
    // persistent grid, each block (gw < 8) or wave (gw >= 8) walks the work table round-robin: gw < 8: 8 blocks per CU at
    // most; gw >= 8: exactly the waves the chip holds at once
    int per_cu = 8;
    {
        // (the occupancy query is a runtime call of ~0.3 ms: asked once per kernel width and process)
        static std::mutex occ_mu;
        static int occ_cache[17] = {0};
        std::lock_guard<std::mutex> lk_occ(occ_mu);
        const int slot = pr->nwave ? 2 + GW : GW;      // (3, 4: the narrow wave-item kernels)
        if (!occ_cache[slot]) {
            if (pr->nwave) {
                if (GW == 1) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pr_sweep_n<1, false>, TPB, 0);
                else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pr_sweep_n<2, false>, TPB, 0);
            } else {
                SS_GW_DISPATCH(GW, sweep_occupancy, &per_cu);
            }
            occ_cache[slot] = per_cu < 1 ? 1 : per_cu;
        }
        per_cu = occ_cache[slot];
    }
    const auto tc1a = t_now();
    per_cu = (int)std::max<int64_t>(1, ctx->opt("pr.blocks_per_cu", per_cu));
    // Several sweeps per launch (k_pr_multi_n), OPT-IN ("pr.persistent" = 1: write-through hand-offs, 2: release / acquire fences): one
    // rank, the reference's uniform teleport, K <= 2 on the wave-item kernel.  The blocks wait for each other between two sweeps, so ALL
    // of them must be resident: half of what the occupancy query admits per CU, at most "pr.persistent_blocks" (default 4).
    // Measured and therefore off by default (round 5, config 2: 2^20 nodes / 5M edges, K = 1): 0.054 ms per sweep with one launch per
    // sweep against 0.113 (write-through) / 0.152 (fences) inside one launch at 4 blocks per CU, 0.075 / 0.094 at 2, 0.078 / 0.082 at 1 —
    // the wait costs ~25 us per 256 resident blocks, far more than the 33 us an empty launch of this sweep costs in all; 10M / 50M:
    // 0.38 against 0.51 ms.  Results are bit-identical in every mode (test_sweeps_inside_one_launch_are_bit_identical).
    {
        const int64_t want = ctx->opt("pr.persistent", 0);
        pr->persist_mode = want == 2 ? 2 : 1;
        pr->persist = pr->nwave && g->world == 1 && !ctx->opt("pr.affine", 0) && want > 0;
        if (pr->persist) {
            // what the runtime admits of k_pr_multi_n itself, less two: the query answers one block per CU too many for kernels with
            // 97-112 SGPRs (MI355X_MICROARCH.md, residency), and a block that is not resident would be waited for in vain
            static std::mutex occ2_mu;
            static int occ_multi[3] = {0, 0, 0};
            std::lock_guard<std::mutex> lk2(occ2_mu);
            if (!occ_multi[GW <= 2 ? GW : 0]) {
                int o = 0;
                if (GW == 1) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_pr_multi_n<1, 1>, TPB, 0);
                else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_pr_multi_n<2, 1>, TPB, 0);
                occ_multi[GW <= 2 ? GW : 0] = std::max(o, 1);
            }
            const int admit = std::max(1, occ_multi[GW <= 2 ? GW : 0] - 2);
            per_cu = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(admit, std::max(1, per_cu / 2)), ctx->opt("pr.persistent_blocks", 4)));
        }
    }
    // How many waves: k_pr_sweep gives every wave at least one item; k_pr_sweep_n at least "pr.items_per_wave" (default 4: 0.0537 / 0.0531 / 0.0522 / 0.0483 / 0.0523 ms at 1 / 2 / 3 / 4 / 6) — on a small
    // graph what a sweep costs is mostly per WAVE (launch, control-block and offset reads, the hand-in of the partial sums), and a wave
    // with a single pass of 64 rows is all overhead.  Config 2 (2^20 nodes / 5M edges, K = 1), ms per sweep by resident blocks per CU:
    // 8: 0.0533, 6: 0.0513, 4: 0.0485, 3: 0.0461, 2: 0.0468.  [Built, measured, removed: the same waves in 1024-thread blocks (a quarter
    // of the arrivals at the hand-in): 0.067 against 0.054 — sixteen waves wait for their slowest at every workgroup barrier.]
    const size_t ipw = pr->nwave ? (size_t)std::max<int64_t>(1, ctx->opt("pr.items_per_wave", 4)) : 1;
    pr->nblocks = vitems ? (unsigned)std::min<size_t>(std::max<size_t>(1, ss::div_up(items.size(), (size_t)WAVES * ipw)), (size_t)ctx->cu_count * per_cu)
                          : (unsigned)std::min<size_t>(items.size(), (size_t)ctx->cu_count * 8);
    static thread_local std::vector<uint32_t> woff, cnt;
    woff.clear();
    if (vitems) {
        // k_pr_sweep: the items' turn counts (from the sorted in-degrees the graph keeps on the host), then the items dealt to the
        // grid's waves; the edge ranges are filled in on the device (k_pr_item_ranges)
        const uint32_t NS = 64 / GI;
        // (rows rise inside a class's items: a cursor per degree table, a bisection only when a row steps back)
        struct DegCursor {
            const ss_graph::SortedDegrees* d;
            size_t j = 0;
            uint32_t at(uint32_t r) {
                if (r >= d->size()) return 0u;
                if (d->start[j] > r) j = d->run_of(r);
                while (d->start[j + 1] <= r) j++;
                return d->val[j];
            }
        } cur_nd{&g->h_indeg_nd}, cur_d{&g->h_indeg_d};
        auto deg_of = [&](uint32_t lrow) -> uint32_t { return lrow < g->sl_nd ? cur_nd.at(lrow) : cur_d.at(lrow - g->sl_nd); };
        cost.assign(items.size(), 0.0);
        const auto td0 = t_now();
        for (size_t i = 0; i < items.size(); i++) {
            const WorkItem& w = items[i];
            double turns = 1.0;
            switch (w.kind) {
                case V_ROWW: turns = ss::div_up(deg_of(w.row), NS * CH); break;
                case V_SEG: turns = ss::div_up(std::min<uint32_t>(SEGW, deg_of(w.row) - w.count * SEGW), NS * CH) + 2.0; break;
                case V_QUAD: turns = (double)ss::div_up(w.count, NS) * w.nseg; break;
                case V_DEG: {
                    if (pr->nwave) { turns = (double)ss::div_up(w.count, 64u) * (0.6 + 0.2 * (w.nseg <= 2 ? 2 : w.nseg <= 4 ? 4 : 8)); break; }   // a pass of 64 rows: 2, 4 or 8 gathers per lane + a row each
                    const uint32_t R = w.nseg <= 2 ? 8 : w.nseg <= 4 ? 4 : 2;
                    turns = (double)ss::div_up(w.count, NS * R) * (R == 8 ? 2.5 : R == 4 ? 1.7 : 1.3);   // a turn finishes R rows per lane group
                    break;
                }
                default: turns = 0.5; break;
            }
            cost[i] = turns + 1.0;                                   // + the item's own overhead
        }
        const auto td0a = t_now();
        // Longest-processing-time deal: items in table order (classes by falling item length), each to the wave with the
        // least work so far — every wave ends up with the same number of turns (+- one item), whatever the degree mix.
        const uint32_t nw = pr->nblocks * WAVES;
        // The items come in falling cost inside each class.  Deal them nw at a time: the waves ordered by their load so far, the
        // chunk's items in table order (costliest first inside a class) to the least loaded waves first — the longest-processing-
        // time rule applied per chunk, one sort of nw loads per chunk instead of a heap operation per item (3 ms -> 0.4 ms at
        // 60k items / 3072 waves, same balance: every wave ends within one item of the mean).
        owner.assign(items.size(), 0u);
        {
            static thread_local std::vector<double> load;
            static thread_local std::vector<uint32_t> by_load, idx;
            static thread_local std::vector<std::pair<double, uint32_t>> key;
            load.assign(nw, 0.0);
            by_load.resize(nw);
            for (uint32_t w = 0; w < nw; w++) by_load[w] = w;
            // Many chunks (a large graph): the loads are not sorted between chunks, every other chunk is dealt in reverse — costs fall
            // smoothly inside a class, so the snake ends as level as the sorted deal (config 4: sweep 0.960 against 0.963 ms) and the
            // deal takes 0.3 ms of host time instead of 2.8.  Few chunks: least-loaded-first as before (config 2 on k_pr_sweep<8>: 0.075
            // against 0.077 ms); k_pr_sweep_n's finer items sweep the same either way and the update is 0.1 ms shorter with the snake.
            const bool snake = ctx->opt("pr.deal_snake", items.size() >= (size_t)8 * nw || pr->nwave ? 1 : 0) != 0;
            // All items by falling cost first ("pr.deal_global": 1 = one order over all classes, 2 = class by class in table order): a
            // counting sort on the cost in sixteenths of a turn, stable (table order inside a bucket), O(items).  The chunks below
            // then come sorted.  [Sorting each chunk took 0.2 of
            // config 2's 0.4 ms here: inside a class the costs fall, but every run of equal rows ends in a short item, so a chunk
            // is dozens of falling runs, not one.]
            static thread_local std::vector<uint32_t> order, bucket;
            // (`tools/pr_deal.py`, sweep ms at 10M / 50M: K = 16 chunks 0.959, global 0.963, class-major 0.955; K = 1 chunks 0.383, global
            //  0.379, class-major 0.389 — k_pr_sweep and k_pr_sweep_n<2> take class-major, k_pr_sweep_n<1> the one global order)
            // (K = 2 on k_pr_sweep_n at 10M / 50M: class-major 0.471, global 0.481, chunks 0.483)
            const int64_t deal_mode = ctx->opt("pr.deal_global", pr->nwave && pr->gw == 1 ? 1 : 2);
            const bool global_order = deal_mode != 0;
            if (global_order) {
                constexpr uint32_t NB = 1u << 14;
                // (mode 2: class-major — the table's class order kept, falling cost inside a class)
                const auto key = [&](size_t i) { return NB - 1 - (uint32_t)std::min<double>(cost[i] * 16.0, (double)(NB - 1)); };
                bucket.assign(NB + 1, 0u);
                order.resize(items.size());
                size_t c0 = 0;
                for (int kcls = 0; kcls < (deal_mode == 2 ? 6 : 1); kcls++) {
                    const size_t c1 = deal_mode == 2 ? (kcls < 5 ? std::min<size_t>(vbeg[kcls + 1], items.size()) : items.size()) : items.size();
                    if (kcls) std::fill(bucket.begin(), bucket.end(), 0u);
                    for (size_t i = c0; i < c1; i++) bucket[key(i) + 1]++;
                    for (uint32_t b = 0; b < NB; b++) bucket[b + 1] += bucket[b];
                    for (size_t i = c0; i < c1; i++) order[c0 + bucket[key(i)]++] = (uint32_t)i;
                    c0 = c1;
                }
            }
            // (Equal modelled loads do not end together: the hardware issues oldest-first, so of a CU's four resident blocks the one
            //  that arrived first is out of items after 623 us at config 4 and the last one after 906 — -DSS_PR_WAVETIME,
            //  tools/pr_wavetime.py.  Shares weighted by those speeds were tried and dropped: the late blocks end where they ended
            //  before, the early ones later, the sweep 0.985-1.0 ms instead of 0.955 — the sweep is bound by the memory system's
            //  throughput, and who finishes first is the scheduler's business.)
#ifdef SS_PR_WAVETIME
            // experiment (SS_PR_HEAP="123,106,92,85"): exact weighted longest-first — every item to the wave whose load / share is least
            bool heap_done = false;
            if (const char* ws = getenv("SS_PR_HEAP")) {
                double wg[8] = {100, 100, 100, 100, 100, 100, 100, 100};
                sscanf(ws, "%lf,%lf,%lf,%lf", &wg[0], &wg[1], &wg[2], &wg[3]);
                const uint32_t wpr = (uint32_t)std::max(ctx->cu_count, 1) * WAVES;
                typedef std::pair<double, uint32_t> E;
                std::priority_queue<E, std::vector<E>, std::greater<E>> pq;
                for (uint32_t w = 0; w < nw; w++) pq.push({0.0, w});
                for (size_t j = 0; j < items.size(); j++) {
                    const uint32_t it = global_order ? order[j] : (uint32_t)j;
                    const E top = pq.top(); pq.pop();
                    owner[it] = top.second;
                    load[top.second] += cost[it];
                    pq.push({load[top.second] / wg[std::min<uint32_t>(7, top.second / wpr)], top.second});
                }
                heap_done = true;
            }
            for (size_t i0 = 0; i0 < items.size() && !heap_done; i0 += nw) {
#else
            for (size_t i0 = 0; i0 < items.size(); i0 += nw) {
#endif
                const size_t n_chunk = std::min<size_t>(nw, items.size() - i0);
                if (i0 && snake) {
                    std::reverse(by_load.begin(), by_load.end());
                } else if (i0) {
                    // (load, wave) pairs sorted by value: several times faster than a comparator that reads load[] through the ids
                    key.resize(nw);
#ifdef SS_PR_WAVETIME
                    // experiment: loads normalised by the share of the wave's block round (SS_PR_WEIGHTS="123,106,92,85")
                    if (const char* ws = getenv("SS_PR_WEIGHTS")) {
                        double wg[8] = {100, 100, 100, 100, 100, 100, 100, 100};
                        sscanf(ws, "%lf,%lf,%lf,%lf", &wg[0], &wg[1], &wg[2], &wg[3]);
                        const uint32_t wpr = (uint32_t)std::max(ctx->cu_count, 1) * WAVES;
                        for (uint32_t w = 0; w < nw; w++) key[w] = {load[w] / wg[std::min<uint32_t>(7, w / wpr)], w};
                    } else
#endif
                    for (uint32_t w = 0; w < nw; w++) key[w] = {load[w], w};
                    std::sort(key.begin(), key.end());
                    for (uint32_t w = 0; w < nw; w++) by_load[w] = key[w].second;
                }
                // the chunk's costliest item to the least loaded wave: order the chunk by falling cost (it already is, except
                // where it crosses a class boundary)
                // (a few falling runs: merged pairwise, O(chunk) per boundary — a full stable_sort of the chunk through cost[] took
                //  0.15 ms a chunk, most of config 2's deal)
                idx.resize(n_chunk);
                for (size_t j = 0; j < n_chunk; j++) idx[j] = global_order ? order[i0 + j] : (uint32_t)(i0 + j);
                const auto falling = [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; };
                size_t run_end = 0, n_merge = 0;
                for (size_t j = 1; j <= n_chunk && !global_order; j++) {
                    if (j < n_chunk && !(cost[i0 + j] > cost[i0 + j - 1])) continue;     // still falling (or level)
                    if (run_end) {
                        if (++n_merge > 8) { std::stable_sort(idx.begin(), idx.end(), falling); break; }
                        std::inplace_merge(idx.begin(), idx.begin() + (ptrdiff_t)run_end, idx.begin() + (ptrdiff_t)j, falling);
                    }
                    run_end = j;
                }
                for (size_t j = 0; j < n_chunk; j++) {
                    owner[idx[j]] = by_load[j];
                    load[by_load[j]] += cost[idx[j]];
                }
            }
        }
#ifdef SS_PR_WAVETIME
        if (FILE* f = fopen("gpurun_out/pr_load.csv", "w")) {
            static thread_local std::vector<double> wl, wcls;
            wl.assign(nw, 0.0); wcls.assign((size_t)nw * 6, 0.0);
            for (size_t i = 0; i < items.size(); i++) {
                int kc = 0; while (kc < 5 && i >= vbeg[kc + 1]) kc++;
                wl[owner[i]] += cost[i]; wcls[(size_t)owner[i] * 6 + kc] += cost[i];
            }
            fprintf(f, "wave,load,c0,c1,c2,c3,c4,c5\n");
            for (uint32_t w = 0; w < nw; w++) fprintf(f, "%u,%.2f,%.2f,%.2f,%.2f,%.2f,%.2f,%.2f\n", w, wl[w], wcls[(size_t)w * 6], wcls[(size_t)w * 6 + 1], wcls[(size_t)w * 6 + 2], wcls[(size_t)w * 6 + 3], wcls[(size_t)w * 6 + 4], wcls[(size_t)w * 6 + 5]);
            fclose(f);
        }
#endif
        const auto td1 = t_now();
        if (trace) fprintf(stderr, "[pr trace]   deal: occupancy query %.2f ms, costs %.3f ms, owners %.3f ms\n", t_ms(tc1, tc1a), t_ms(td0, td0a), t_ms(td0a, td1));
        // table order inside a wave's list = item order = class order: count per (wave, class), offsets, place
        // (the items are in class order: owner[i] * 8 + class, computed once per class range)
        woff.assign((size_t)nw * 8, 0);
        cnt.assign((size_t)nw * 8, 0);
        for (int k = 0; k < 6; k++) {
            const size_t i1 = k < 5 ? std::min<size_t>(vbeg[k + 1], items.size()) : items.size();
            for (size_t i = std::min<size_t>(vbeg[k], i1); i < i1; i++) { owner[i] = owner[i] * 8 + (uint32_t)k; cnt[owner[i]]++; }
        }
        uint32_t run_off = 0;
        for (uint32_t w = 0; w < nw; w++) {
            for (int k = 0; k < 8; k++) {
                woff[(size_t)w * 8 + k] = run_off;
                run_off += cnt[(size_t)w * 8 + k];
                cnt[(size_t)w * 8 + k] = woff[(size_t)w * 8 + k];       // becomes the write cursor of (wave, class)
            }
        }
        dealt.resize(items.size());
        for (size_t i = 0; i < items.size(); i++) dealt[cnt[owner[i]]++] = items[i];
        dealt.push_back({V_ZERO, 0, 0, 0, 0, 0, 0, 0});              // the pipelines read two items ahead
        dealt.push_back({V_ZERO, 0, 0, 0, 0, 0, 0, 0});
        items.swap(dealt);
        if (trace) fprintf(stderr, "[pr trace]   deal: placement %.3f ms\n", t_ms(td1, t_now()));
    }

    // the graph's build temporaries (ss_graph::late_free): its last kernels ran under the host work above
    g->settle();
    const auto tc2 = t_now();
    const unsigned begin_blocks = std::max(1u, std::min(2048u, ss::div_up(n_local * GW, TPB)));
    SS_HIP(ctx, pr->partials.alloc(((size_t)std::max(pr->nblocks, begin_blocks) + 8) * 2 * GW));   // block rows + 8 group rows
    SS_HIP(ctx, pr->segpart.alloc((size_t)std::max(nsegs, 1u) * GW));
    SS_HIP(ctx, pr->rowticket.alloc(std::max(nmulti, 1u)));
    SS_HIP(ctx, hipMemsetAsync(pr->rowticket.p, 0, pr->rowticket.bytes(), st));
    SS_HIP(ctx, pr->woff.alloc(std::max<size_t>(woff.size(), 8)));
    if (!woff.empty()) SS_HIP(ctx, hipMemcpyAsync(pr->woff.p, woff.data(), woff.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    SS_HIP(ctx, pr->work.alloc(items.size()));
    SS_HIP(ctx, hipMemcpyAsync(pr->work.p, items.data(), items.size() * sizeof(WorkItem), hipMemcpyHostToDevice, st));
    if (vitems)
        hipLaunchKernelGGL(k_pr_item_ranges, dim3(ss::div_up(items.size(), TPB)), dim3(TPB), 0, st, pr->work.p, (uint32_t)items.size(), (const uint32_t*)g->in_ptr.p);
    SS_HIP(ctx, pr->ctl.alloc(1));
    SS_HIP(ctx, hipMemsetAsync(pr->ctl.p, 0, sizeof(PrCtl), st));
    double h_x0[MAXK];
    for (int k = 0; k < MAXK; k++) h_x0[k] = k < k_topics ? 1.0 / (double)n_topic[k] : 0.0;   // pagerank.go:104
    SS_HIP(ctx, pr->x0.alloc(MAXK));
    SS_HIP(ctx, hipMemcpyAsync(pr->x0.p, h_x0, sizeof(h_x0), hipMemcpyHostToDevice, st));
    SS_HIP(ctx, hipStreamSynchronize(st));   // items / h_x0 are stack/host temporaries
    if (trace) fprintf(stderr, "[pr trace] ss_pr_create: build_work %.2f ms (%zu items), edge ranges + deal %.2f ms, alloc + upload %.2f ms\n", t_ms(tc0, tc1), items.size(), t_ms(tc1, tc2), t_ms(tc2, t_now()));
    if (trace) fprintf(stderr, "[pr trace] where the state lives: x %p  tab0 %p  tab1 %p  in_src %p  in_ptr %p  outdeg %p  work %p  woff %p\n", (void*)pr->x.p, (void*)pr->tab0.p,
                       (void*)pr->tab1.p, (void*)g->in_src.p, (void*)g->in_ptr.p, (void*)g->outdeg.p, (void*)pr->work.p, (void*)pr->woff.p);

    PrParams& p = pr->prm;
    p.in_ptr = g->in_ptr.p;
    p.in_src = g->in_src.p;
    p.outdeg = g->outdeg.p;
    p.x = pr->x.p;
    if (g->world == 1) {
        // sweep s reads tab[s&1], writes tab[(s&1)^1]; begin writes tab_wr[1] = tab0
        p.tab_rd[0] = pr->tab0.p; p.tab_wr[0] = pr->tab1.p;
        p.tab_rd[1] = pr->tab1.p; p.tab_wr[1] = pr->tab0.p;
    } else {
        p.tab_rd[0] = p.tab_rd[1] = pr->tab0.p;
        p.tab_wr[0] = p.tab_wr[1] = pr->send.p;
    }
    p.work = pr->work.p;
    p.partials = pr->partials.p;
    p.segpart = pr->segpart.p;
    p.rowticket = pr->rowticket.p;
    p.ctl = pr->ctl.p;
    p.x0 = pr->x0.p;
    p.d = damping;
    p.teleport = 1.0 - damping;                       // pagerank.go:90
    p.eps = eps;
    p.tele_n = p.teleport * (double)g->n;             // pagerank.go:112
    p.max_iter = max_iter;
    p.k_topics = k_topics;
    p.world = g->world;
    p.sl_nd = g->sl_nd;
    p.cnt_nd = g->cnt_nd;
    p.sl_d = g->sl_d;
    p.cnt_d = g->cnt_d;
    p.seg_edges = seg_edges;
    p.n_items = (uint32_t)items.size();
    p.pos_nd = pos_nd;
    p.pos_d = pos_d;
    p.zrow = (uint32_t)g->nd_int;
    p.woff = pr->woff.p;
    p.stagger_div = ctx->opt("pr.stagger", 0) != 0 ? (uint32_t)std::max(ctx->cu_count, 1) : 0u;
    p.stagger_code = ctx->opt("pr.stagger", 0) >= 10 ? (uint32_t)(ctx->opt("pr.stagger", 0) - 10) : 0u;
    {
        // "pr.class_order": six decimal digits, position by position (012345 = long rows, mid rows, the three short-row classes,
        // edge-less rows); anything that is not a permutation of 0..5 falls back to that order
        int64_t code = ctx->opt("pr.class_order", 235401);
        uint32_t packed = 0, seen = 0;
        for (int pos = 5; pos >= 0; pos--) {
            const uint32_t c = (uint32_t)(code % 10);
            code /= 10;
            packed |= (c & 7u) << (3 * pos);
            if (c < 6) seen |= 1u << c;
        }
        p.class_order = seen == 0x3Fu ? packed : (0u | 1u << 3 | 2u << 6 | 3u << 9 | 4u << 12 | 5u << 15);
        // "pr.n_class_order": four digits for k_pr_sweep_n's phases (0 = long rows, 1 = mid rows, 2 = rows of <= 8 in-edges, 3 = edge-less rows)
        // (all 24 orders, round 5: 2^20 nodes / 5M edges 0.0472 ms for 2-3-1-0 against 0.0492 for 0-1-2-3, the worst; at 10M / 50M the
        //  numbering order is within 0.2 % of the best and short-rows-first among the worst: 0.3796 against 0.379 / 0.389)
        int64_t c4 = ctx->opt("pr.n_class_order", (size_t)g->n_local() <= ((size_t)4 << 20) ? 2310 : 123);
        uint32_t p4 = 0, s4 = 0;
        for (int pos = 3; pos >= 0; pos--) {
            const uint32_t c = (uint32_t)(c4 % 10);
            c4 /= 10;
            p4 |= (c & 3u) << (2 * pos);
            if (c < 4) s4 |= 1u << c;
        }
        p.n_order = s4 == 0xFu ? p4 : (0u | 1u << 2 | 2u << 4 | 3u << 6);
    }
#ifdef SS_PR_EXP_KINDMASK
    p.kind_mask = getenv("SS_PR_KIND_MASK") ? (uint32_t)strtoul(getenv("SS_PR_KIND_MASK"), nullptr, 0) : 0xFFFFFFFFu;
    {
        size_t cnt[16] = {0};
        for (auto& it : items) cnt[it.kind & 15]++;
        fprintf(stderr, "[pr] items: seg %zu wave %zu group %zu zero %zu | vseg %zu vroww %zu vquad %zu vdeg %zu vzero %zu; blocks %u\n", cnt[0], cnt[1], cnt[2], cnt[3],
                cnt[8], cnt[9], cnt[10], cnt[11], cnt[12], pr->nblocks);
    }
#endif
    g->users++;
    *out = guard.release();
    return SS_OK;
}

int32_t ss_pr_set_teleport(ss_pr* pr, const uint64_t* set_ptr, const uint32_t* set_nodes) {
    if (!pr) return SS_ERR_INVALID;
    ss_ctx* ctx = pr->g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    if (pr->begun) return ctx->fail(SS_ERR_STATE, "ss_pr_set_teleport: call before ss_pr_begin");
    hipStream_t st = ctx->stream;
    PrParams& p = pr->prm;
    if (!set_ptr) {                                  // back to the reference's uniform teleport
        p.memb = nullptr; p.tin = nullptr; p.nz_in = nullptr; p.ts_mask = 0;
        return SS_OK;
    }
    const ss_graph* g = pr->g;
    const int K = pr->k;
    std::vector<uint64_t> h_ptr(K + 1);
    SS_HIP(ctx, ss::copy_in(ctx->stream, h_ptr.data(), set_ptr, (K + 1) * sizeof(uint64_t)));
    if (h_ptr[0] != 0) return ctx->fail(SS_ERR_INVALID, "ss_pr_set_teleport: set_ptr[0] != 0");
    for (int k = 0; k < K; k++)
        if (h_ptr[k + 1] < h_ptr[k]) return ctx->fail(SS_ERR_INVALID, "ss_pr_set_teleport: set_ptr not non-decreasing");
    const uint64_t total = h_ptr[K];
    if (total && !set_nodes) return ctx->fail(SS_ERR_INVALID, "ss_pr_set_teleport: set_nodes is NULL");
    double h_tin[MAXK];
    uint32_t mask = 0;
    for (int k = 0; k < MAXK; k++) {
        h_tin[k] = 0.0;
        if (k < K && h_ptr[k + 1] > h_ptr[k]) {
            mask |= 1u << k;
            h_tin[k] = p.teleport * (double)g->n / (double)(h_ptr[k + 1] - h_ptr[k]);     // the set shares the mass (1-d)*N
        }
    }
    const size_t n_local = g->n_local();
    ss::DevBuf<uint64_t> d_ptr;
    ss::DevBuf<uint32_t> d_nodes, d_err, d_seen;
    ss::DevBuf<unsigned long long> d_cnt;
    SS_HIP(ctx, d_ptr.alloc(K + 1));
    SS_HIP(ctx, d_seen.alloc(g->n));
    SS_HIP(ctx, hipMemsetAsync(d_seen.p, 0, std::max<size_t>(d_seen.bytes(), 4), st));
    SS_HIP(ctx, d_nodes.alloc(total));
    SS_HIP(ctx, d_err.alloc(1));
    SS_HIP(ctx, d_cnt.alloc(MAXK));
    SS_HIP(ctx, pr->memb.alloc(n_local));
    SS_HIP(ctx, pr->tin.alloc(MAXK));
    SS_HIP(ctx, pr->nz_in.alloc(MAXK));
    SS_HIP(ctx, hipMemcpyAsync(d_ptr.p, h_ptr.data(), (K + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    if (total) SS_HIP(ctx, hipMemcpyAsync(d_nodes.p, set_nodes, total * sizeof(uint32_t), hipMemcpyDefault, st));
    SS_HIP(ctx, hipMemsetAsync(pr->memb.p, 0, std::max<size_t>(pr->memb.bytes(), 4), st));
    SS_HIP(ctx, hipMemsetAsync(d_err.p, 0, sizeof(uint32_t), st));
    SS_HIP(ctx, hipMemsetAsync(d_cnt.p, 0, MAXK * sizeof(unsigned long long), st));
    if (total)
        hipLaunchKernelGGL(k_pr_memb, dim3(std::min<unsigned>(ss::div_up(total, TPB), 4096u)), dim3(TPB), 0, st, (const uint64_t*)d_ptr.p,
                           (const uint32_t*)d_nodes.p, K, g->n, (const uint32_t*)g->new_id.p, g->nd_int, std::max(g->sl_nd, 1u), std::max(g->sl_d, 1u),
                           g->rank, pr->memb.p, d_seen.p, d_err.p);
    const uint32_t n_zero = (g->cnt_nd - p.pos_nd) + (g->cnt_d - p.pos_d);
    if (n_zero)
        hipLaunchKernelGGL(k_pr_memb_zero_count, dim3(std::min<unsigned>(ss::div_up(n_zero, TPB), 4096u)), dim3(TPB), 0, st,
                           (const uint32_t*)pr->memb.p, g->sl_nd, g->cnt_nd, p.pos_nd, g->cnt_d, p.pos_d, K, d_cnt.p);
    SS_HIP(ctx, hipGetLastError());
    uint32_t h_err = 0;
    unsigned long long h_cnt[MAXK];
    SS_HIP(ctx, ss::fetch(ctx, st, &h_err, d_err.p, sizeof(h_err), h_cnt, d_cnt.p, sizeof(h_cnt)));
    if (h_err & 1u) return ctx->fail(SS_ERR_INVALID, "ss_pr_set_teleport: a teleport set holds a node id >= n_nodes");
    if (h_err & 2u) return ctx->fail(SS_ERR_INVALID, "ss_pr_set_teleport: a teleport set lists a node twice (the ids of a set must be distinct)");
    double h_nz[MAXK];
    for (int k = 0; k < MAXK; k++) h_nz[k] = (double)h_cnt[k];
    SS_HIP(ctx, hipMemcpyAsync(pr->tin.p, h_tin, sizeof(h_tin), hipMemcpyHostToDevice, st));
    SS_HIP(ctx, hipMemcpyAsync(pr->nz_in.p, h_nz, sizeof(h_nz), hipMemcpyHostToDevice, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    p.memb = pr->memb.p;
    p.tin = pr->tin.p;
    p.nz_in = pr->nz_in.p;
    p.ts_mask = mask;
    return SS_OK;
}

int32_t ss_pr_destroy(ss_pr* pr) {
#ifdef SS_PR_WAVETIME
    if (pr && pr->gw >= 8) {
        (void)hipDeviceSynchronize();
        const uint32_t nwv = std::min<uint32_t>(65536u, pr->nblocks * WAVES);
        std::vector<unsigned long long> h((size_t)nwv * 2);
        if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_pr_wt), h.size() * sizeof(unsigned long long)) == hipSuccess && nwv) {
            unsigned long long t0 = ~0ull;
            for (uint32_t w = 0; w < nwv; w++) t0 = std::min(t0, h[2 * w]);
            std::vector<double> st(nwv), en(nwv);
            for (uint32_t w = 0; w < nwv; w++) { st[w] = (double)(h[2 * w] - t0) / 100.0; en[w] = (double)(h[2 * w + 1] - t0) / 100.0; }
            std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
            if (FILE* f = fopen("gpurun_out/pr_wt.csv", "w")) {
                fprintf(f, "wave,start_us,end_us\n");
                for (uint32_t w = 0; w < nwv; w++) fprintf(f, "%u,%.2f,%.2f\n", w, (double)(h[2 * w] - t0) / 100.0, (double)(h[2 * w + 1] - t0) / 100.0);
                fclose(f);
            }
            fprintf(stderr, "[pr wavetime] %u waves: start us median %.1f max %.1f | out of items us min %.1f p10 %.1f median %.1f p90 %.1f p99 %.1f max %.1f\n", nwv,
                    st[nwv / 2], st[nwv - 1], en[0], en[nwv / 10], en[nwv / 2], en[nwv * 9 / 10], en[(size_t)nwv * 99 / 100], en[nwv - 1]);
        }
    }
#endif

    if (!pr) return SS_ERR_INVALID;
    ss_ctx* ctx = pr->g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    if (!ss::device_wedged(ctx->device)) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->comm_stream);
    }
    pr->g->users--;
    delete pr;
    return SS_OK;
}

int32_t ss_pr_begin(ss_pr* pr) {
    if (!pr) return SS_ERR_INVALID;
    ss_ctx* ctx = pr->g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    if (pr->need_finalize) return ctx->fail(SS_ERR_STATE, "ss_pr_begin: pending exchange/finalize");
    const size_t n_el = (size_t)pr->g->n_local() * pr->gw;
    const unsigned nb = std::max(1u, std::min(2048u, ss::div_up(n_el, TPB)));
    SS_HIP(ctx, hipMemsetAsync(&pr->ctl.p->ticket, 0, sizeof(uint32_t), ctx->stream));
    SS_GW_DISPATCH(pr->gw, launch_begin, pr, ctx->stream, nb);
    SS_HIP(ctx, hipGetLastError());
    pr->begun = true;
    if (pr->g->world > 1) {
        pr->need_finalize = true;
        pr->finalize_is_begin = true;
    }
    return SS_OK;
}

int32_t ss_pr_step(ss_pr* pr, int32_t n_steps) {
    if (!pr) return SS_ERR_INVALID;
    ss_ctx* ctx = pr->g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    if (!pr->begun) return ctx->fail(SS_ERR_STATE, "ss_pr_step: ss_pr_begin not called");
    if (pr->need_finalize) return ctx->fail(SS_ERR_STATE, "ss_pr_step: pending exchange/finalize");
    if (n_steps < 1) return ctx->fail(SS_ERR_INVALID, "ss_pr_step: n_steps < 1");
    if (pr->g->world > 1 && n_steps != 1) return ctx->fail(SS_ERR_INVALID, "ss_pr_step: world>1 needs an exchange after every step");
    SS_HIP(ctx, hipEventRecord(ctx->ev[0][0], ctx->stream));
    if (pr->persist && !pr->prm.memb && !pr->prm.aff) {
        SS_GW_DISPATCH(pr->gw, launch_multi, pr, ctx->stream, (int)n_steps);   // the sweeps wait for each other inside the launch
    } else {
        for (int i = 0; i < n_steps; i++) SS_GW_DISPATCH(pr->gw, launch_step, pr, ctx->stream);
    }
    SS_HIP(ctx, hipEventRecord(ctx->ev[0][1], ctx->stream));
    ctx->ev_valid[0] = true;
    SS_HIP(ctx, hipGetLastError());
    if (pr->g->world > 1) {
        pr->need_finalize = true;
        pr->finalize_is_begin = false;
    }
    return SS_OK;
}

int32_t ss_pr_finalize(ss_pr* pr) {
    if (!pr) return SS_ERR_INVALID;
    ss_ctx* ctx = pr->g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    if (pr->g->world == 1) return ctx->fail(SS_ERR_STATE, "ss_pr_finalize: world==1 folds the finalize into the sweep");
    if (!pr->need_finalize) return ctx->fail(SS_ERR_STATE, "ss_pr_finalize: nothing to finalize");
    SS_GW_DISPATCH(pr->gw, launch_finalize, pr, ctx->stream, pr->finalize_is_begin ? 1 : 0);
    SS_HIP(ctx, hipGetLastError());
    pr->need_finalize = false;
    return SS_OK;
}

int32_t ss_pr_exchange_buffers(ss_pr* pr, void** send_dev, uint64_t* send_bytes, void** recv_dev, uint64_t* recv_bytes) {
    if (!pr) return SS_ERR_INVALID;
    ss_ctx* ctx = pr->g->ctx;
    if (pr->g->world == 1) return ctx->fail(SS_ERR_STATE, "ss_pr_exchange_buffers: world==1 has no exchange");
    if (send_dev) *send_dev = pr->send.p;
    if (send_bytes) *send_bytes = pr->send.bytes();
    if (recv_dev) *recv_dev = pr->tab0.p;
    if (recv_bytes) *recv_bytes = (uint64_t)pr->g->nd_int * pr->gw * sizeof(double);   // without the table's zero row
    return SS_OK;
}

int32_t ss_pr_exchange(ss_pr* pr, int32_t allreduce) {
    if (!pr) return SS_ERR_INVALID;
    ss_ctx* ctx = pr->g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    const ss_graph* g = pr->g;
    if (g->world == 1) return ctx->fail(SS_ERR_STATE, "ss_pr_exchange: world==1 has no exchange");
    if (!pr->need_finalize) return ctx->fail(SS_ERR_STATE, "ss_pr_exchange: nothing to exchange (call after ss_pr_begin / ss_pr_step)");
    if (!ctx->comm || ctx->comm_world != g->world || ctx->comm_rank != g->rank)
        return ctx->fail(SS_ERR_STATE, "ss_pr_exchange: the context's communicator (rank %d of %d) does not match the graph's shard (rank %d of %d)",
                         ctx->comm ? ctx->comm_rank : -1, ctx->comm ? ctx->comm_world : 0, g->rank, g->world);
    const size_t slice = (size_t)g->sl_nd * pr->gw;            // doubles per rank
    if (!allreduce && ctx->opt("pr.wire_f32", 0) != 0) {       // opt-in: float32 on the wire (see k_wire_pack)
        SS_HIP(ctx, wire_alloc(pr));
        wire_pack(pr, ctx->stream);
        SS_TRY(ss::comm_allgather(ctx, pr->wire_send.p, pr->wire_recv.p, wire_floats(pr) * sizeof(float)));
        wire_unpack(pr, ctx->stream);
        return SS_OK;
    }
    if (!allreduce) return ss::comm_allgather(ctx, pr->send.p, pr->tab0.p, slice * sizeof(double));
    // north-star form: own slice inside a zeroed full-size table, tables summed
    SS_HIP(ctx, hipMemsetAsync(pr->tab0.p, 0, pr->tab0.bytes(), ctx->stream));
    SS_HIP(ctx, hipMemcpyAsync(pr->tab0.p + (size_t)g->rank * slice, pr->send.p, slice * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return ss::comm_allreduce_f64(ctx, pr->tab0.p, pr->tab0.p, slice * (size_t)g->world);
}

}  // extern "C"

namespace {

// ---- the sharded power iteration as ONE pipeline inside the library ---------------------------------
// The K topic vectors are split into B topic blocks (separate states over the same shard); block b's exchange runs on the
// context's second stream while block b+1 is finalised and swept on the first:
//     compute:  [fin 0][sweep 0]      [fin 1][sweep 1]      [fin 0][sweep 0] ...
//     comm:                 [exchange 0 ..........][exchange 1 ..........]
// HIP events order the two streams; the host enqueues and looks at the device-side stop rule every BATCH sweeps.  Topics are
// independent power iterations (pagerank.go:54-63), so a block's results are bit for bit those of running its topics alone.
// The exchange is a functor: RCCL for one shard per process (ss_pagerank_run_sharded), device-to-device copies for the
// shards of a single-process group (ss_pagerank_run_group: tests, and one process driving several shards on one device).
struct ShardBlocks { std::vector<ss_pr*> blk; };

template <typename Exchange>
int32_t run_pipelined(ss_ctx* ctx, std::vector<ShardBlocks>& sh, int32_t max_iter, Exchange&& exchange, int32_t* iters_out,
                      const std::vector<int>& blk_k0) {
    const int B = (int)sh[0].blk.size(), S = (int)sh.size();
    hipStream_t cs = ctx->stream, xs = ctx->comm_stream;
    std::vector<hipEvent_t> ev_step(B), ev_xchg(B);
    for (int b = 0; b < B; b++) {
        SS_HIP(ctx, hipEventCreateWithFlags(&ev_step[b], hipEventDisableTiming));
        SS_HIP(ctx, hipEventCreateWithFlags(&ev_xchg[b], hipEventDisableTiming));
    }
    auto cleanup = [&] {
        if (!ss::device_wedged(ctx->device)) {                       // (after a timed-out collective neither stream will ever drain)
            (void)hipStreamSynchronize(xs);
            (void)hipStreamSynchronize(cs);
        }
        for (int b = 0; b < B; b++) { (void)hipEventDestroy(ev_step[b]); (void)hipEventDestroy(ev_xchg[b]); }
    };
    int32_t rc = SS_OK;
    auto start_exchange = [&](int b) -> int32_t {
        SS_HIP(ctx, hipEventRecord(ev_step[b], cs));
        SS_HIP(ctx, hipStreamWaitEvent(xs, ev_step[b], 0));
        SS_TRY(exchange(b, xs));
        SS_HIP(ctx, hipEventRecord(ev_xchg[b], xs));
        return SS_OK;
    };
    // prime: every block begun, its first exchange in flight
    for (int b = 0; b < B && rc == SS_OK; b++) {
        for (int s = 0; s < S; s++) {
            ss_pr* pr = sh[s].blk[b];
            const size_t n_el = (size_t)pr->g->n_local() * pr->gw;
            const unsigned nb = std::max(1u, std::min(2048u, ss::div_up(n_el, TPB)));
            SS_HIP(ctx, hipMemsetAsync(&pr->ctl.p->ticket, 0, sizeof(uint32_t), cs));
            SS_GW_DISPATCH(pr->gw, launch_begin, pr, cs, nb);
            pr->begun = true;
        }
        rc = start_exchange(b);
    }
    bool first = true;
    const int BATCH = 8;
    int32_t sweeps_done = 0;
    for (;;) {
        if (rc != SS_OK) break;
        int todo = BATCH;
        if (max_iter > 0) todo = std::min(BATCH, std::max(1, max_iter - sweeps_done));
        for (int i = 0; i < todo && rc == SS_OK; i++) {
            for (int b = 0; b < B && rc == SS_OK; b++) {
                if (hipStreamWaitEvent(cs, ev_xchg[b], 0) != hipSuccess) { rc = ctx->fail(SS_ERR_HIP, "hipStreamWaitEvent"); break; }
                for (int s = 0; s < S; s++) SS_GW_DISPATCH(sh[s].blk[b]->gw, launch_finalize, sh[s].blk[b], cs, first ? 1 : 0);
                for (int s = 0; s < S; s++) SS_GW_DISPATCH(sh[s].blk[b]->gw, launch_step, sh[s].blk[b], cs);
                rc = start_exchange(b);
            }
            first = false;
            sweeps_done++;
        }
        if (rc != SS_OK) break;
        if (hipGetLastError() != hipSuccess) { rc = ctx->fail(SS_ERR_HIP, "sharded sweep: launch failed"); break; }
        // the stop rule lives on the device and is evaluated identically on every rank (same gathered sums, same order);
        // what the host reads lags the in-flight sweep by one, and launches after convergence are no-ops on all ranks alike
        int32_t n_active = 0;
        for (int b = 0; b < B && rc == SS_OK; b++) {
            ctx->pin_used = 0;
            PrCtl* const hp = ctx->pin<PrCtl>();
            if (hipMemcpyAsync(hp, sh[0].blk[b]->ctl.p, sizeof(PrCtl), hipMemcpyDeviceToHost, cs) != hipSuccess) { rc = ctx->fail(SS_ERR_HIP, "sharded sweep: status read failed"); break; }
            if ((rc = ss::sync_bounded(ctx, cs, "sharded sweep (waiting for the exchange)")) != SS_OK) break;
            n_active += hp->n_active;
        }
        if (n_active == 0) break;
    }
    // drain: apply the exchanges still in flight (no-ops after convergence)
    for (int b = 0; b < B && rc == SS_OK; b++) {
        if (hipStreamWaitEvent(cs, ev_xchg[b], 0) != hipSuccess) { rc = ctx->fail(SS_ERR_HIP, "hipStreamWaitEvent"); break; }
        for (int s = 0; s < S; s++) {
            SS_GW_DISPATCH(sh[s].blk[b]->gw, launch_finalize, sh[s].blk[b], cs, first ? 1 : 0);
            sh[s].blk[b]->need_finalize = false;
        }
    }
    if (rc == SS_OK && iters_out) {
        for (int b = 0; b < B && rc == SS_OK; b++) {
            ctx->pin_used = 0;
            PrCtl* const hp = ctx->pin<PrCtl>();
            if (hipMemcpyAsync(hp, sh[0].blk[b]->ctl.p, sizeof(PrCtl), hipMemcpyDeviceToHost, cs) != hipSuccess) { rc = ctx->fail(SS_ERR_HIP, "sharded sweep: status read failed"); break; }
            if ((rc = ss::sync_bounded(ctx, cs, "sharded sweep (last exchange)")) != SS_OK) break;
            for (int k = 0; k < sh[0].blk[b]->k; k++) iters_out[blk_k0[b] + k] = hp->iters[k];
        }
    }
    cleanup();
    return rc;
}

// topic blocks of a K-topic run: option "pr.topic_blocks" (default 2 above 8 topics: one block's exchange then hides behind
// the other block's sweep — and both blocks still fill an 8-wide table; splitting 8 topics or fewer would pad every block to
// the 8-wide kernel and double the bytes on the wire), never more blocks than topics
int topic_blocks_for(ss_ctx* ctx, int k_topics) {
    int B = (int)ctx->opt("pr.topic_blocks", k_topics > 8 ? 2 : 1);
    return std::max(1, std::min(B, k_topics));
}

// ---- the two-vector form on a sharded graph (option "pr.affine"): a TWO-column exchange per iteration, whatever K is -------------
// One K = 2 state per shard; per iteration: sweep (local rows) -> all-gather of the 2-wide contribution slices -> k_pr_finalize
// (affine branch: r, s, the column teleports; identical on every rank) -> the topics' L1 changes over the local rows -> all-gather
// of the ranks' K sums -> stop rule (k_aff_ctl adds them in rank order) -> write-out of the topics that have just stopped.
struct AffShard {
    ss::DevBuf<AffCtl> aff;
    ss::DevBuf<double> x_alt, partials, loc, gath, out;
    double n_zero = 0.0;
};
int32_t aff_attach(ss_pr* pr, AffShard& A, int32_t k_topics, const int32_t* n_topic, int world) {
    ss_ctx* ctx = pr->g->ctx;
    hipStream_t st = ctx->stream;
    if (!pr->nwave || pr->gw != 2) return ctx->fail(SS_ERR_UNSUPPORTED, "pr.affine: needs the wave-item K = 2 sweep (pr.narrow_wave / pr.force_narrow at their defaults)");
    const ss_graph* g = pr->g;
    const size_t n_rows = (size_t)g->cnt_nd + g->cnt_d;
    SS_HIP(ctx, A.aff.alloc(1));
    SS_HIP(ctx, A.x_alt.alloc((size_t)g->n_local() * 2));
    SS_HIP(ctx, A.partials.alloc((size_t)AFF_NB * AFF_MAXK));
    SS_HIP(ctx, A.loc.alloc(AFF_MAXK));
    SS_HIP(ctx, A.gath.alloc((size_t)world * AFF_MAXK));
    SS_HIP(ctx, A.out.alloc(std::max<size_t>(1, (size_t)k_topics * n_rows)));
    SS_HIP(ctx, hipMemsetAsync(A.loc.p, 0, A.loc.bytes(), st));
    std::vector<AffCtl> h(1);
    std::memset(h.data(), 0, sizeof(AffCtl));
    for (int k = 0; k < k_topics; k++) { h[0].u[k] = 1.0 / (double)n_topic[k]; h[0].active[k] = 1; }   // pagerank.go:104
    h[0].k_real = k_topics;
    h[0].n_active = k_topics;
    SS_HIP(ctx, hipMemcpyAsync(A.aff.p, h.data(), sizeof(AffCtl), hipMemcpyHostToDevice, st));
    const double x0[MAXK] = {1.0, 0.0};
    SS_HIP(ctx, hipMemcpyAsync(pr->x0.p, x0, sizeof(x0), hipMemcpyHostToDevice, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    pr->prm.aff = A.aff.p;
    pr->prm.tele_col = pr->ctl.p->tele;
    pr->prm.x_alt = A.x_alt.p;
    A.n_zero = (double)((g->cnt_nd - pr->prm.pos_nd) + (g->cnt_d - pr->prm.pos_d));
    return SS_OK;
}
template <typename Exchange, typename ExchangeSums>
int32_t run_affine_sharded(ss_ctx* ctx, std::vector<ShardBlocks>& sh, std::vector<AffShard>& A, int world, double eps, int32_t max_iter, int32_t k_topics,
                           Exchange&& exchange, ExchangeSums&& exchange_sums, int32_t* iters_out) {
    const int S = (int)sh.size();
    hipStream_t st = ctx->stream;
    for (int s = 0; s < S; s++) {
        ss_pr* pr = sh[s].blk[0];
        const size_t n_el = (size_t)pr->g->n_local() * pr->gw;
        const unsigned nb = std::max(1u, std::min(2048u, ss::div_up(n_el, TPB)));
        SS_HIP(ctx, hipMemsetAsync(&pr->ctl.p->ticket, 0, sizeof(uint32_t), st));
        launch_begin<2>(pr, st, nb);
        pr->begun = true;
    }
    SS_TRY(exchange(0, st));
    for (int s = 0; s < S; s++) launch_finalize<2>(sh[s].blk[0], st, 1);
    int32_t n_active = k_topics, it = 0;
    const int BATCH = 4;
    // ONE collective per iteration (option "pr.affine_lag", default 1): the ranks' per-topic L1 sums of iteration i are written into the
    // spare tail rows of the rank's contribution slice (graph.hpp: TAIL_SUM_ROWS) and travel with iteration i + 1's all-gather, so the
    // stop decisions of iteration i are taken one exchange later — from the same numbers, added in the same rank order — and the ranks
    // of a topic that stops are written from the vectors of the iteration it stopped in, which the alternating pair still holds.
    // A second, 2 KB all-gather per iteration was pure latency on a ~0.25 ms iteration; it survives only as the flush
    // after the last sweep of a max_iter run.  0: the round-4 protocol (sums in a collective of their own, decisions at once).
    // (the spare rows hold 2 * TAIL_SUM_ROWS = 64 sums = SS_MAX_TOPICS; these two entry points take up to AFF_MAXK topics in this form, and
    //  beyond 64 the sums keep their own collective)
    const bool lag = ctx->opt("pr.affine_lag", 1) != 0 && k_topics <= (int32_t)(2 * TAIL_SUM_ROWS);
    auto sums_row = [&](ss_pr* pr, double* base) { return base + ((size_t)pr->g->sl_nd - 2 - TAIL_SUM_ROWS) * 2; };
    auto delta_and_local = [&](int s, double* loc) {
        ss_pr* pr = sh[s].blk[0];
        const ss_graph* g = pr->g;
        const double2* const x_old = reinterpret_cast<const double2*>((it & 1) ? A[s].x_alt.p : pr->x.p);
        const double2* const x_new = reinterpret_cast<const double2*>((it & 1) ? pr->x.p : A[s].x_alt.p);
        hipLaunchKernelGGL(k_aff_delta, dim3(AFF_NB), dim3(TPB), 0, st, x_old, x_new, (const AffCtl*)A[s].aff.p, g->sl_nd, pr->prm.pos_nd, pr->prm.pos_d,
                           A[s].partials.p);
        hipLaunchKernelGGL(k_aff_local, dim3(1), dim3(AFF_MAXK), 0, st, (const AffCtl*)A[s].aff.p, (const PrCtl*)pr->ctl.p, (const double*)A[s].partials.p,
                           AFF_NB, A[s].n_zero, loc);
    };
    auto ctl_and_emit = [&](int s, const double* sums, size_t stride, int emit_lag) {
        ss_pr* pr = sh[s].blk[0];
        const ss_graph* g = pr->g;
        // (emit_lag: the vectors before the last sweep — iteration it - 1 — else the ones it has just written)
        const bool odd = (it & 1) != 0;
        const double2* const x_src = reinterpret_cast<const double2*>(emit_lag ? (odd ? A[s].x_alt.p : pr->x.p) : (odd ? pr->x.p : A[s].x_alt.p));
        hipLaunchKernelGGL(k_aff_ctl, dim3(1), dim3(AFF_MAXK), 0, st, A[s].aff.p, pr->ctl.p, sums, (unsigned)world, 0.0, eps, max_iter, stride);
        const unsigned nbe = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(4096, ss::div_up((uint64_t)g->cnt_nd + g->cnt_d, TPB)));
        hipLaunchKernelGGL(k_aff_emit_local, dim3(nbe), dim3(TPB), 0, st, x_src, (const PrCtl*)pr->ctl.p, (const AffCtl*)A[s].aff.p, g->sl_nd, g->cnt_nd,
                           g->cnt_d, pr->prm.pos_nd, pr->prm.pos_d, A[s].out.p, emit_lag);
    };
    while (n_active > 0) {
        for (int b = 0; b < BATCH; b++) {
            for (int s = 0; s < S; s++) launch_step<2>(sh[s].blk[0], st);
            SS_TRY(exchange(0, st));
            for (int s = 0; s < S; s++) launch_finalize<2>(sh[s].blk[0], st, 0);
            if (lag) {
                for (int s = 0; s < S; s++) {
                    ss_pr* pr = sh[s].blk[0];
                    // iteration it - 1: its sums have just arrived with this iteration's table (every rank's slice, rank order)
                    if (it > 0) ctl_and_emit(s, sums_row(pr, pr->tab0.p), (size_t)pr->g->sl_nd * 2, 1);
                    delta_and_local(s, sums_row(pr, pr->send.p));              // iteration it: into the slice the NEXT all-gather sends
                }
            } else {
                for (int s = 0; s < S; s++) delta_and_local(s, A[s].loc.p);
                SS_TRY(exchange_sums(st));
                for (int s = 0; s < S; s++) ctl_and_emit(s, A[s].gath.p, (size_t)AFF_MAXK, 0);
            }
            it++;
            if (max_iter > 0 && it >= max_iter) break;
        }
        SS_HIP(ctx, hipGetLastError());
        ctx->pin_used = 0;
        int32_t* const hn = ctx->pin<int32_t>();
        SS_HIP(ctx, hipMemcpyAsync(hn, &A[0].aff.p->n_active, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        SS_TRY(ss::sync_bounded(ctx, st, "sharded two-vector sweep (waiting for the exchange)"));
        n_active = *hn;
        if (max_iter > 0 && it >= max_iter) break;
    }
    if (lag && max_iter > 0 && it >= max_iter && n_active > 0) {
        // the last sweep's sums have no further all-gather to ride on: one small collective of their own, then the decisions (every
        // topic still running stops here: it >= max_iter) and the ranks from the vectors that sweep has written
        it--;                                                          // (the lambdas take the parity of the sweep they describe)
        for (int s = 0; s < S; s++) {
            ss_pr* pr = sh[s].blk[0];
            hipLaunchKernelGGL(k_aff_local, dim3(1), dim3(AFF_MAXK), 0, st, (const AffCtl*)A[s].aff.p, (const PrCtl*)pr->ctl.p, (const double*)A[s].partials.p,
                               AFF_NB, A[s].n_zero, A[s].loc.p);
        }
        SS_TRY(exchange_sums(st));
        for (int s = 0; s < S; s++) ctl_and_emit(s, A[s].gath.p, (size_t)AFF_MAXK, 0);
        it++;
        SS_HIP(ctx, hipGetLastError());
    }
    for (int s = 0; s < S; s++) sh[s].blk[0]->need_finalize = false;
    if (iters_out) {
        std::vector<AffCtl> h(1);
        SS_HIP(ctx, hipMemcpyAsync(h.data(), A[0].aff.p, sizeof(AffCtl), hipMemcpyDeviceToHost, st));
        SS_TRY(ss::sync_bounded(ctx, st, "sharded two-vector sweep (status)"));
        for (int k = 0; k < k_topics; k++) iters_out[k] = h[0].iters[k];
    }
    return SS_OK;
}
// the ids of a shard's rows in local order (what ss_pr_read_local returns beside the ranks)
int32_t local_ids(ss_pr* pr, uint32_t* ids_out) {
    const size_t n_rows = (size_t)pr->g->cnt_nd + pr->g->cnt_d;
    std::vector<double> scratch(std::max<size_t>(1, n_rows * (size_t)pr->k));
    return ss_pr_read_local(pr, ids_out, scratch.data());
}

}  // namespace

extern "C" {

int32_t ss_pagerank_run_sharded(ss_graph* g, double damping, double eps, int32_t max_iter, int32_t k_topics,
                                const int32_t* n_topic, int32_t allreduce, uint32_t* ids_out, double* rank_out, int32_t* iters_out) {
    if (!g) return SS_ERR_INVALID;
    ss_ctx* ctx = g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (g->world < 2) return ctx->fail(SS_ERR_STATE, "ss_pagerank_run_sharded: needs a sharded graph (world > 1); use ss_pagerank_run");
    const bool affine = ctx->opt("pr.affine", 0) != 0 && !allreduce;      // the two-vector form: a 2-column exchange whatever K is
    const int k_max = affine ? AFF_MAXK : MAXK;
    if (k_topics < 1 || k_topics > k_max || !n_topic || !rank_out)
        return ctx->fail(SS_ERR_INVALID, "ss_pagerank_run_sharded: k_topics must be 1..%d, n_topic/rank_out not NULL", k_max);
    if (!(eps >= 0.0) && max_iter <= 0 && !(eps != eps))
        return ctx->fail(SS_ERR_INVALID, "ss_pagerank_run_sharded: eps < 0 (never converges) needs max_iter > 0");
    if (!ctx->comm || ctx->comm_world != g->world || ctx->comm_rank != g->rank)
        return ctx->fail(SS_ERR_STATE, "ss_pagerank_run_sharded: the context's communicator (rank %d of %d) does not match the graph's shard (rank %d of %d)",
                         ctx->comm ? ctx->comm_rank : -1, ctx->comm ? ctx->comm_world : 0, g->rank, g->world);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    const int B = affine ? 1 : topic_blocks_for(ctx, k_topics);
    std::vector<ShardBlocks> sh(1);
    std::vector<AffShard> aff(affine ? 1 : 0);
    std::vector<int> k0(B + 1);
    for (int b = 0; b <= B; b++) k0[b] = (int)((int64_t)k_topics * b / B);
    int32_t rc = SS_OK;
    const int32_t two[2] = {1, 1};
    for (int b = 0; b < B && rc == SS_OK; b++) {
        ss_pr* pr = nullptr;
        rc = affine ? ss_pr_create(g, damping, -1.0, 0, 2, two, &pr) : ss_pr_create(g, damping, eps, max_iter, k0[b + 1] - k0[b], n_topic + k0[b], &pr);
        if (rc == SS_OK) sh[0].blk.push_back(pr);
    }
    if (affine && rc == SS_OK)
        for (int k = 0; k < k_topics && rc == SS_OK; k++)
            if (n_topic[k] < 1) rc = ctx->fail(SS_ERR_INVALID, "ss_pagerank_run_sharded: n_topic[%d] < 1", k);
    if (affine && rc == SS_OK) rc = aff_attach(sh[0].blk[0], aff[0], k_topics, n_topic, g->world);
    const bool wire_f32 = ctx->opt("pr.wire_f32", 0) != 0;
    if (rc == SS_OK) {
        auto exchange = [&](int b, hipStream_t xs) -> int32_t {
            ss_pr* pr = sh[0].blk[b];
            const size_t slice = (size_t)g->sl_nd * pr->gw;            // doubles per rank
            if (!allreduce && wire_f32) {
                SS_HIP(ctx, wire_alloc(pr));
                wire_pack(pr, xs);
                SS_TRY(ss::comm_allgather_on(ctx, pr->wire_send.p, pr->wire_recv.p, wire_floats(pr) * sizeof(float), xs));
                wire_unpack(pr, xs);
                return SS_OK;
            }
            if (!allreduce) return ss::comm_allgather_on(ctx, pr->send.p, pr->tab0.p, slice * sizeof(double), xs);
            // north-star form: own slice inside a zeroed full-size table, tables summed
            SS_HIP(ctx, hipMemsetAsync(pr->tab0.p, 0, pr->tab0.bytes(), xs));
            SS_HIP(ctx, hipMemcpyAsync(pr->tab0.p + (size_t)g->rank * slice, pr->send.p, slice * sizeof(double), hipMemcpyDeviceToDevice, xs));
            return ss::comm_allreduce_f64_on(ctx, pr->tab0.p, pr->tab0.p, slice * (size_t)g->world, xs);
        };
        if (affine) {
            auto exchange_sums = [&](hipStream_t xs) -> int32_t {
                return ss::comm_allgather_on(ctx, aff[0].loc.p, aff[0].gath.p, (size_t)AFF_MAXK * sizeof(double), xs);
            };
            rc = run_affine_sharded(ctx, sh, aff, g->world, eps, max_iter, k_topics, exchange, exchange_sums, iters_out);
        } else {
            rc = run_pipelined(ctx, sh, max_iter, exchange, iters_out, k0);
        }
    }
    // this rank's rows: ids once, ranks block by block (topic-major: rank_out[k][rows])
    const size_t n_rows = (size_t)g->cnt_nd + g->cnt_d;
    if (affine) {
        if (rc == SS_OK && ids_out) rc = local_ids(sh[0].blk[0], ids_out);
        if (rc == SS_OK && n_rows) {
            if (hipMemcpyAsync(rank_out, aff[0].out.p, (size_t)k_topics * n_rows * sizeof(double), hipMemcpyDefault, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                rc = ctx->fail(SS_ERR_HIP, "ss_pagerank_run_sharded: copy of the ranks failed");
        }
    } else
    for (int b = 0; b < (int)sh[0].blk.size() && rc == SS_OK; b++)
        rc = ss_pr_read_local(sh[0].blk[b], b == 0 ? ids_out : nullptr, rank_out + (size_t)k0[b] * n_rows);
    for (ss_pr* pr : sh[0].blk) ss_pr_destroy(pr);
    return rc;
}

int32_t ss_pagerank_run_group(ss_graph* const* shards, int32_t world, double damping, double eps, int32_t max_iter, int32_t k_topics,
                              const int32_t* n_topic, double* rank_out, int32_t* iters_out) {
    if (!shards || world < 2 || !shards[0]) return SS_ERR_INVALID;
    ss_ctx* ctx = shards[0]->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    const bool affine = ctx->opt("pr.affine", 0) != 0;
    const int k_max = affine ? AFF_MAXK : MAXK;
    if (k_topics < 1 || k_topics > k_max || !n_topic || !rank_out)
        return ctx->fail(SS_ERR_INVALID, "ss_pagerank_run_group: k_topics must be 1..%d, n_topic/rank_out not NULL", k_max);
    if (!(eps >= 0.0) && max_iter <= 0 && !(eps != eps))
        return ctx->fail(SS_ERR_INVALID, "ss_pagerank_run_group: eps < 0 (never converges) needs max_iter > 0");
    for (int s = 0; s < world; s++)
        if (!shards[s] || shards[s]->ctx != ctx || shards[s]->world != world || shards[s]->rank != s || shards[s]->n != shards[0]->n)
            return ctx->fail(SS_ERR_INVALID, "ss_pagerank_run_group: shards[%d] is not shard %d of %d of the same graph on this context", s, s, world);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    const int B = affine ? 1 : topic_blocks_for(ctx, k_topics);
    std::vector<ShardBlocks> sh(world);
    std::vector<AffShard> aff(affine ? world : 0);
    std::vector<int> k0(B + 1);
    for (int b = 0; b <= B; b++) k0[b] = (int)((int64_t)k_topics * b / B);
    int32_t rc = SS_OK;
    const int32_t two[2] = {1, 1};
    for (int k = 0; k < k_topics && affine && rc == SS_OK; k++)
        if (n_topic[k] < 1) rc = ctx->fail(SS_ERR_INVALID, "ss_pagerank_run_group: n_topic[%d] < 1", k);
    for (int s = 0; s < world && rc == SS_OK; s++)
        for (int b = 0; b < B && rc == SS_OK; b++) {
            ss_pr* pr = nullptr;
            rc = affine ? ss_pr_create(shards[s], damping, -1.0, 0, 2, two, &pr)
                        : ss_pr_create(shards[s], damping, eps, max_iter, k0[b + 1] - k0[b], n_topic + k0[b], &pr);
            if (rc == SS_OK) sh[s].blk.push_back(pr);
            if (rc == SS_OK && affine) rc = aff_attach(pr, aff[s], k_topics, n_topic, world);
        }
    if (rc == SS_OK) {
        // the all-gather, by hand: every shard's slice into every shard's table, rank order
        const bool wire_f32 = ctx->opt("pr.wire_f32", 0) != 0;
        auto exchange = [&](int b, hipStream_t xs) -> int32_t {
            if (wire_f32) {                                         // the float32 wire format, played by device-to-device copies
                for (int src = 0; src < world; src++) {
                    SS_HIP(ctx, wire_alloc(sh[src].blk[b]));
                    wire_pack(sh[src].blk[b], xs);
                }
                for (int dst = 0; dst < world; dst++) {
                    ss_pr* to = sh[dst].blk[b];
                    for (int src = 0; src < world; src++) {
                        ss_pr* from = sh[src].blk[b];
                        SS_HIP(ctx, hipMemcpyAsync(to->wire_recv.p + (size_t)src * wire_floats(from), from->wire_send.p, wire_floats(from) * sizeof(float),
                                                   hipMemcpyDeviceToDevice, xs));
                    }
                    wire_unpack(to, xs);
                }
                return SS_OK;
            }
            for (int dst = 0; dst < world; dst++)
                for (int src = 0; src < world; src++) {
                    ss_pr* from = sh[src].blk[b];
                    ss_pr* to = sh[dst].blk[b];
                    const size_t slice = (size_t)shards[src]->sl_nd * from->gw;
                    SS_HIP(ctx, hipMemcpyAsync(to->tab0.p + (size_t)src * slice, from->send.p, slice * sizeof(double), hipMemcpyDeviceToDevice, xs));
                }
            return SS_OK;
        };
        if (affine) {
            auto exchange_sums = [&](hipStream_t xs) -> int32_t {     // the K sums of every shard into every shard's table, rank order
                for (int dst = 0; dst < world; dst++)
                    for (int src = 0; src < world; src++)
                        SS_HIP(ctx, hipMemcpyAsync(aff[dst].gath.p + (size_t)src * AFF_MAXK, aff[src].loc.p, (size_t)AFF_MAXK * sizeof(double),
                                                   hipMemcpyDeviceToDevice, xs));
                return SS_OK;
            };
            rc = run_affine_sharded(ctx, sh, aff, world, eps, max_iter, k_topics, exchange, exchange_sums, iters_out);
        } else {
            rc = run_pipelined(ctx, sh, max_iter, exchange, iters_out, k0);
        }
    }
    // assemble [K][N] by original id
    const uint64_t n = shards[0]->n;
    for (int s = 0; s < world && rc == SS_OK && affine; s++) {
        const size_t n_rows = (size_t)shards[s]->cnt_nd + shards[s]->cnt_d;
        std::vector<uint32_t> ids(n_rows);
        std::vector<double> part((size_t)k_topics * n_rows);
        rc = local_ids(sh[s].blk[0], ids.data());
        if (rc == SS_OK && n_rows) {
            if (hipMemcpyAsync(part.data(), aff[s].out.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                rc = ctx->fail(SS_ERR_HIP, "ss_pagerank_run_group: copy of the ranks failed");
        }
        for (int k = 0; k < k_topics && rc == SS_OK; k++)
            for (size_t i = 0; i < n_rows; i++) rank_out[(size_t)k * n + ids[i]] = part[(size_t)k * n_rows + i];
    }
    for (int s = 0; s < world && rc == SS_OK && !affine; s++) {
        const size_t n_rows = (size_t)shards[s]->cnt_nd + shards[s]->cnt_d;
        std::vector<uint32_t> ids(n_rows);
        for (int b = 0; b < B && rc == SS_OK; b++) {
            const int kb = k0[b + 1] - k0[b];
            std::vector<double> part((size_t)kb * n_rows);
            rc = ss_pr_read_local(sh[s].blk[b], ids.data(), part.data());
            if (rc != SS_OK) break;
            for (int k = 0; k < kb; k++)
                for (size_t i = 0; i < n_rows; i++) rank_out[(size_t)(k0[b] + k) * n + ids[i]] = part[(size_t)k * n_rows + i];
        }
    }
    for (auto& s : sh)
        for (ss_pr* pr : s.blk) ss_pr_destroy(pr);
    return rc;
}

int32_t ss_pr_status(ss_pr* pr, int32_t* iters_out, int32_t* n_active, int32_t* sweeps, double* last_delta_out,
                     double* last_total_out) {
    if (!pr) return SS_ERR_INVALID;
    ss_ctx* ctx = pr->g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    ctx->pin_used = 0;
    PrCtl* const hp = ctx->pin<PrCtl>();               // pinned: a read-back into pageable memory pins the page per call (ss_ctx::h_pin)
    SS_HIP(ctx, hipMemcpyAsync(hp, pr->ctl.p, sizeof(PrCtl), hipMemcpyDeviceToHost, ctx->stream));
    SS_TRY(ss::sync_bounded(ctx, ctx->stream, "ss_pr_status"));
    const PrCtl h = *hp;
    if (h.stuck) return ctx->fail(SS_ERR_STATE, "ss_pr_status: the multi-sweep kernel gave up waiting between two sweeps (its grid of %u blocks was not resident: "
                                  "other kernels held the CUs for seconds); the state is void — set option pr.persistent = 0 and run again", pr->nblocks);
    for (int k = 0; k < pr->k; k++) {
        if (iters_out) iters_out[k] = h.iters[k];
        if (last_delta_out) last_delta_out[k] = h.delta[k];
        if (last_total_out) last_total_out[k] = h.S[k];
    }
    if (n_active) *n_active = h.n_active;
    if (sweeps) *sweeps = h.sweep;
    return SS_OK;
}

int32_t ss_pr_read_local(ss_pr* pr, uint32_t* ids_out, double* rank_out) {
    if (!pr || !rank_out) return SS_ERR_INVALID;
    ss_ctx* ctx = pr->g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    const ss_graph* g = pr->g;
    const size_t n_rows = (size_t)g->cnt_nd + g->cnt_d;
    ss::DevBuf<uint32_t> d_ids;
    ss::DevBuf<double> d_out;
    SS_HIP(ctx, d_ids.alloc(n_rows));
    SS_HIP(ctx, d_out.alloc(n_rows * pr->k));
    SS_GW_DISPATCH(pr->gw, launch_read, pr, ctx->stream, 0, (uint64_t)n_rows, d_ids.p, d_out.p);
    SS_HIP(ctx, hipGetLastError());
    if (ids_out && n_rows) SS_HIP(ctx, hipMemcpyAsync(ids_out, d_ids.p, n_rows * sizeof(uint32_t), hipMemcpyDefault, ctx->stream));
    if (n_rows) SS_HIP(ctx, hipMemcpyAsync(rank_out, d_out.p, n_rows * pr->k * sizeof(double), hipMemcpyDefault, ctx->stream));
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

int32_t ss_pr_read(ss_pr* pr, double* rank_out) {
    if (!pr || !rank_out) return SS_ERR_INVALID;
    ss_ctx* ctx = pr->g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    const ss_graph* g = pr->g;
    if (g->world != 1) return ctx->fail(SS_ERR_STATE, "ss_pr_read: world>1, use ss_pr_read_local");
    // ranks wanted in device memory: written there directly (no K*N*8-byte staging buffer and copy)
    hipPointerAttribute_t pa{};
    const bool dev_out = hipPointerGetAttributes(&pa, rank_out) == hipSuccess && pa.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    ss::DevBuf<double> d_out;
    if (!dev_out) SS_HIP(ctx, d_out.alloc((size_t)g->n * pr->k));
    SS_GW_DISPATCH(pr->gw, launch_read, pr, ctx->stream, 1, (uint64_t)g->n, (uint32_t*)nullptr, dev_out ? rank_out : d_out.p);
    SS_HIP(ctx, hipGetLastError());
    if (!dev_out) SS_HIP(ctx, hipMemcpyAsync(rank_out, d_out.p, (size_t)g->n * pr->k * sizeof(double), hipMemcpyDefault, ctx->stream));
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SS_OK;
}

int32_t ss_pr_probe(ss_pr* pr, int32_t mode, int32_t n_reps, float* ms_out) {
    if (!pr || !ms_out) return SS_ERR_INVALID;
    ss_ctx* ctx = pr->g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    SS_HIP(ctx, hipSetDevice(ctx->device));
    if (pr->gw < 8) return ctx->fail(SS_ERR_UNSUPPORTED, "ss_pr_probe: the probe mirrors the chunked gather of the K >= 5 kernels");
    if (mode < 0 || (mode & 7) > 2 || (mode >> 3) > 4 || n_reps < 1) return ctx->fail(SS_ERR_INVALID, "ss_pr_probe: mode 0..2 (+ 8 * gather cache policy 0..4), n_reps >= 1");
    const int pol = mode >> 3;
    mode &= 7;
    const uint32_t hot = (uint32_t)ctx->opt("pr.probe_hot", 24576);
    const ss_graph* g = pr->g;
    const unsigned nb = (unsigned)ctx->cu_count * 8;
    ss::DevBuf<double> sink;
    SS_HIP(ctx, sink.alloc((size_t)nb * TPB));
    hipEvent_t e0, e1;
    SS_HIP(ctx, hipEventCreate(&e0));
    SS_HIP(ctx, hipEventCreate(&e1));
    hipStream_t st = ctx->stream;
    const double* T = pr->tab0.p;
    for (int r = 0; r < n_reps + 1; r++) {
        if (r == 1) SS_HIP(ctx, hipEventRecord(e0, st));          // first launch = warm-up
#define SS_PROBE_LAUNCH(GWV, POLV) hipLaunchKernelGGL((k_pr_probe<GWV, POLV>), dim3(nb), dim3(TPB), 0, st, T, (const uint32_t*)g->in_src.p, (size_t)g->e_local, (uint32_t)g->nd_int, mode, sink.p, hot)
        if (pr->gw == 8) SS_PROBE_LAUNCH(8, 0);
        else switch (pol) {
            case 0: SS_PROBE_LAUNCH(16, 0); break;
            case 1: SS_PROBE_LAUNCH(16, 1); break;
            case 2: SS_PROBE_LAUNCH(16, 2); break;
            case 3: SS_PROBE_LAUNCH(16, 3); break;
            default: SS_PROBE_LAUNCH(16, 4); break;
        }
    }
    SS_HIP(ctx, hipEventRecord(e1, st));
    SS_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    SS_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    SS_HIP(ctx, hipGetLastError());
    *ms_out = ms / (float)n_reps;
    return SS_OK;
}

// ss_pagerank_run in the two-vector form (option "pr.affine" = 1; world 1, the reference's uniform teleport): every topic of the
// reference's recurrence from TWO vectors (AffCtl).  One iteration = a copy of the stored vectors, one K = 2 sweep (k_pr_sweep_n<2>),
// one streaming pass for the topics' L1 changes, the stop rule, and the write-out of the topics that have just stopped.
static int32_t run_affine(ss_graph* g, double damping, double eps, int32_t max_iter, int32_t k_topics, const int32_t* n_topic,
                          double* rank_out, int32_t* iters_out) {
    ss_ctx* ctx = g->ctx;
    hipStream_t st = ctx->stream;
    if (k_topics > AFF_MAXK) return ctx->fail(SS_ERR_UNSUPPORTED, "ss_pagerank_run (pr.affine): at most %d topics", AFF_MAXK);
    for (int k = 0; k < k_topics; k++)
        if (n_topic[k] < 1) return ctx->fail(SS_ERR_INVALID, "ss_pagerank_run: n_topic[%d] < 1", k);
    const int32_t two[2] = {1, 1};
    ss_pr* pr = nullptr;
    SS_TRY(ss_pr_create(g, damping, -1.0, 0, 2, two, &pr));
    struct Guard { ss_pr* p; ~Guard() { ss_pr_destroy(p); } } guard{pr};
    if (!pr->nwave || pr->gw != 2) return ctx->fail(SS_ERR_UNSUPPORTED, "ss_pagerank_run (pr.affine): needs the wave-item K = 2 sweep (pr.narrow_wave / pr.force_narrow at their defaults)");
    const uint32_t n_local = g->n_local();
    ss::DevBuf<AffCtl> aff;
    ss::DevBuf<double> x_prev, partials, d_out;
    SS_HIP(ctx, aff.alloc(1));
    SS_HIP(ctx, x_prev.alloc((size_t)n_local * 2));
    SS_HIP(ctx, partials.alloc((size_t)AFF_NB * AFF_MAXK));
    std::vector<AffCtl> h(1);
    std::memset(h.data(), 0, sizeof(AffCtl));
    for (int k = 0; k < k_topics; k++) { h[0].u[k] = 1.0 / (double)n_topic[k]; h[0].active[k] = 1; }   // pagerank.go:104
    h[0].k_real = k_topics;
    h[0].n_active = k_topics;
    SS_HIP(ctx, hipMemcpyAsync(aff.p, h.data(), sizeof(AffCtl), hipMemcpyHostToDevice, st));
    const double x0[MAXK] = {1.0, 0.0};                            // p = 1, q = 0: the start vector is u * 1
    SS_HIP(ctx, hipMemcpyAsync(pr->x0.p, x0, sizeof(x0), hipMemcpyHostToDevice, st));
    SS_HIP(ctx, hipStreamSynchronize(st));                         // (h, x0: host temporaries)
    pr->prm.aff = aff.p;
    pr->prm.tele_col = pr->ctl.p->tele;
    pr->prm.x_alt = x_prev.p;
    // results: straight into the caller's array when it lives on the device
    hipPointerAttribute_t at{};
    const bool dev_out = hipPointerGetAttributes(&at, rank_out) == hipSuccess && at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    double* out = rank_out;
    if (!dev_out) {
        SS_HIP(ctx, d_out.alloc((size_t)k_topics * g->n));
        out = d_out.p;
    }
    SS_TRY(ss_pr_begin(pr));
    const double n_zero = (double)((g->cnt_nd - pr->prm.pos_nd) + (g->cnt_d - pr->prm.pos_d));
    int32_t n_active = k_topics, it = 0;
    const int BATCH = 4;
    while (n_active > 0) {
        for (int b = 0; b < BATCH; b++) {
            // sweep number `it` (from 0) reads x / x_alt (even / odd) and writes the other one
            const double2* const x_old = reinterpret_cast<const double2*>((it & 1) ? x_prev.p : pr->x.p);
            const double2* const x_new = reinterpret_cast<const double2*>((it & 1) ? pr->x.p : x_prev.p);
            launch_step<2>(pr, st);
            hipLaunchKernelGGL(k_aff_delta, dim3(AFF_NB), dim3(TPB), 0, st, x_old, x_new, (const AffCtl*)aff.p, g->sl_nd, pr->prm.pos_nd, pr->prm.pos_d, partials.p);
            hipLaunchKernelGGL(k_aff_ctl, dim3(1), dim3(AFF_MAXK), 0, st, aff.p, pr->ctl.p, (const double*)partials.p, AFF_NB, n_zero, eps, max_iter, (size_t)AFF_MAXK);
            hipLaunchKernelGGL(k_aff_emit, dim3((unsigned)std::min<uint64_t>(4096, ss::div_up(g->n, TPB))), dim3(TPB), 0, st, x_new, (const PrCtl*)pr->ctl.p,
                               (const AffCtl*)aff.p, (const uint32_t*)g->new_id.p, g->n, g->sl_nd, pr->prm.pos_nd, pr->prm.pos_d, out);
            it++;
            if (max_iter > 0 && it >= max_iter) break;
        }
        SS_HIP(ctx, hipGetLastError());
        SS_HIP(ctx, ss::fetch(ctx, st, &n_active, &aff.p->n_active, sizeof(int32_t)));
        if (max_iter > 0 && it >= max_iter) break;                 // (the stop rule has closed every topic at max_iter)
    }
    SS_HIP(ctx, hipMemcpyAsync(h.data(), aff.p, sizeof(AffCtl), hipMemcpyDeviceToHost, st));
    if (!dev_out) SS_HIP(ctx, hipMemcpyAsync(rank_out, d_out.p, (size_t)k_topics * g->n * sizeof(double), hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    if (iters_out)
        for (int k = 0; k < k_topics; k++) iters_out[k] = h[0].iters[k];
    return SS_OK;
}

int32_t ss_pagerank_run(ss_graph* g, double damping, double eps, int32_t max_iter, int32_t k_topics,
                        const int32_t* n_topic, double* rank_out, int32_t* iters_out) {
    if (!g) return SS_ERR_INVALID;
    ss_ctx* ctx = g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (g->world != 1) return ctx->fail(SS_ERR_STATE, "ss_pagerank_run: needs a (rank 0, world 1) graph");
    if (k_topics < 1 || k_topics > SS_MAX_TOPICS || !n_topic || !rank_out)
        return ctx->fail(SS_ERR_INVALID, "ss_pagerank_run: bad k_topics / NULL argument");
    if (!(eps >= 0.0) && max_iter <= 0 && !(eps != eps))
        return ctx->fail(SS_ERR_INVALID, "ss_pagerank_run: eps < 0 (never converges) needs max_iter > 0");
    if (ctx->opt("pr.affine", 0) != 0) return run_affine(g, damping, eps, max_iter, k_topics, n_topic, rank_out, iters_out);
    // topics are independent power iterations (pagerank.go:54-63): run them MAXK at a time
    for (int k0 = 0; k0 < k_topics; k0 += MAXK) {
        const int kk = std::min(MAXK, k_topics - k0);
        ss_pr* pr = nullptr;
        SS_TRY(ss_pr_create(g, damping, eps, max_iter, kk, n_topic + k0, &pr));
        int32_t rc = ss_pr_begin(pr);
        int32_t n_active = kk, sweeps = 0;
        // the stop rule lives on the device; the host only looks every BATCH sweeps
        // (launches after convergence are no-ops)
        const int BATCH = 8;
        while (rc == SS_OK && n_active > 0) {
            int todo = BATCH;
            if (max_iter > 0) todo = std::min(BATCH, std::max(1, max_iter - sweeps));
            rc = ss_pr_step(pr, todo);
            if (rc == SS_OK) rc = ss_pr_status(pr, iters_out ? iters_out + k0 : nullptr, &n_active, &sweeps, nullptr, nullptr);
        }
        if (rc == SS_OK) rc = ss_pr_read(pr, rank_out + (size_t)k0 * g->n);
        ss_pr_destroy(pr);
        if (rc != SS_OK) return rc;
    }
    return SS_OK;
}

}  // extern "C"
