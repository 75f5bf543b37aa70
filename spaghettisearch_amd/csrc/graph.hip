// graph.hip — build the PageRank graph layout on the device (see graph.hpp).
// Setup code: runs once per UpdateTopicSensitivePagerank call.  Sorting and
// scanning use rocPRIM device primitives; the hot loop lives in pagerank.hip.
#include "graph.hpp"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>

namespace {

constexpr int TPB = 256;

// out-degree per node, count of non-dangling nodes.  A grid of at most 2048 blocks strides over the nodes and adds its count once:
// atomics on ONE word retire ~80M a second — one per wave took 1.9 ms at 10M nodes, one per 256-node block still 0.47 ms.
__global__ __launch_bounds__(TPB) void k_outdeg(const uint64_t* __restrict__ out_ptr, uint64_t n, uint32_t* __restrict__ outdeg,
                                                unsigned long long* __restrict__ n_nd, uint32_t* __restrict__ err) {
    __shared__ unsigned s_cnt[TPB / 64];
    unsigned nd = 0;
    bool bad = false;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t a = out_ptr[v], b = out_ptr[v + 1];
        if (b < a || b - a > 0xFFFFFFFFull) { bad = true; b = a; }
        outdeg[v] = (uint32_t)(b - a);
        nd += b > a;
    }
    if (bad) atomicOr(err, 1u);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nd += __shfl_xor(nd, d);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = nd;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned c = 0;
        for (int w = 0; w < TPB / 64; w++) c += s_cnt[w];
        if (c) atomicAdd(n_nd, (unsigned long long)c);
    }
}

// range check of out_dst (the in-degrees come out of the edge sort: see build)
__global__ void k_check_dst(const uint32_t* __restrict__ out_dst, uint64_t e, uint64_t n, uint32_t* __restrict__ err) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (; i < e; i += stride) bad = bad || out_dst[i] >= n;
    if (bad) atomicOr(err, 2u);
}
// indeg[uniq[j]] = cnt[j] for the runs of the destination-sorted edge keys (indeg is zero elsewhere)
__global__ void k_scatter_runs(const uint32_t* __restrict__ uniq, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ n_runs,
                               uint32_t* __restrict__ indeg) {
    const uint32_t n = *n_runs;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) indeg[uniq[j]] = cnt[j];
}

// sort key per node: class (dangling last), in-degree descending; the sort is stable and the values start as the ids in
// ascending order, so equal keys keep ascending original id
// (the in-degree takes deg_bits bits — no in-degree exceeds the edge count —, the class the bit above: the sort runs over deg_bits + 1)
__global__ void k_row_keys(const uint32_t* __restrict__ outdeg, const uint32_t* __restrict__ indeg, uint64_t n, uint32_t deg_bits,
                           uint32_t* __restrict__ keys, uint32_t* __restrict__ ids) {
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint32_t cls = outdeg[v] == 0 ? 1u : 0u;
    const uint32_t top = (1u << deg_bits) - 1u;
    keys[v] = (cls << deg_bits) | (top - min(indeg[v], top));
    ids[v] = (uint32_t)v;
}

// Everything the host reads back at the end of the build in one block (one copy, one wait): four row offsets, the two run counts,
// and the first `spec` (value, length) runs of each class's in-degrees.  Layout: [4 x u64 | 2 x u32 | pad to 64 B | class 0 values |
// class 0 lengths | class 1 values | class 1 lengths], each run array `spec` words.
__global__ void k_pack_readback(const uint64_t* __restrict__ in_ptr_int, uint64_t p0, uint64_t p1, uint64_t p2, uint64_t p3,
                                const uint32_t* __restrict__ r_n, const uint32_t* __restrict__ v0, const uint32_t* __restrict__ c0,
                                const uint32_t* __restrict__ v1, const uint32_t* __restrict__ c1, uint32_t spec, unsigned char* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        uint64_t* o = reinterpret_cast<uint64_t*>(out);
        o[0] = in_ptr_int[p0]; o[1] = in_ptr_int[p1]; o[2] = in_ptr_int[p2]; o[3] = in_ptr_int[p3];
        uint32_t* q = reinterpret_cast<uint32_t*>(out + 32);
        q[0] = r_n[0]; q[1] = r_n[1];
    }
    uint32_t* runs = reinterpret_cast<uint32_t*>(out + 64);
    const uint32_t n0 = min(r_n[0], spec), n1 = min(r_n[1], spec);
    for (uint32_t j = t; j < spec; j += gridDim.x * blockDim.x) {
        if (j < n0) { runs[j] = v0[j]; runs[spec + j] = c0[j]; }
        if (j < n1) { runs[2 * spec + j] = v1[j]; runs[3 * spec + j] = c1[j]; }
    }
}

// sorted position -> internal id (round-robin deal over ranks inside each class)
__global__ void k_assign_ids(const uint32_t* __restrict__ sorted_ids, uint64_t n, uint64_t n_nd, uint32_t world,
                             uint32_t sl_nd, uint32_t sl_d, const uint32_t* __restrict__ indeg,
                             uint32_t* __restrict__ new_id, uint32_t* __restrict__ old_id,
                             uint32_t* __restrict__ indeg_int) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = sorted_ids[i];
    uint64_t id;
    if (i < n_nd) {
        uint64_t g = i % world, pos = i / world;
        id = g * sl_nd + pos;
    } else {
        uint64_t j = i - n_nd;
        uint64_t g = j % world, pos = j / world;
        id = (uint64_t)world * sl_nd + g * sl_d + pos;
    }
    new_id[v] = (uint32_t)id;
    old_id[id] = v;
    indeg_int[id] = indeg[v];
}

// The rows of EK_CHUNK consecutive CSR positions [base, last] (the block's chunk): thread t gets the rows of its EK_PT consecutive
// positions base + EK_PT * t + k, row = the largest r with ptr[r] <= position.  The rows the chunk spans come from k_chunk_first_rows
// (chunk_row[c] = the row that holds position c * EK_CHUNK); every row that STARTS inside the chunk marks its first position in LDS
// (atomicMax: of several rows starting at one position — empty rows — the last one owns it), and a running maximum over the
// positions hands every position its row.  [A bisection per edge over the spanned rows, ~10 dependent loads from L2 each, took
// 0.52 ms for config 4's 50M edges; two bisections per block by two threads still 0.17 ms.]
constexpr int EK_PT = 8, EK_CHUNK = TPB * EK_PT;
template <typename PtrT>
__global__ void k_chunk_first_rows(const PtrT* __restrict__ ptr, uint64_t n_rows, uint64_t e_total, uint32_t* __restrict__ chunk_row) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint64_t a = ptr[r], b = ptr[r + 1];
    if (b <= a) return;
    for (uint64_t c = (a + EK_CHUNK - 1) / EK_CHUNK; c * EK_CHUNK < b; c++) chunk_row[c] = (uint32_t)r;
    if (b == e_total) chunk_row[(e_total + EK_CHUNK - 1) / EK_CHUNK] = (uint32_t)r;      // the row of the last position: the last chunk's end
}
template <typename PtrT>
__device__ __forceinline__ uint32_t chunk_rows(const PtrT* __restrict__ ptr, const uint32_t* __restrict__ chunk_row, uint64_t base, uint64_t last,
                                               uint32_t* s_row, uint32_t* s_w, uint32_t (&rel)[EK_PT]) {
    {
        uint4* z = reinterpret_cast<uint4*>(s_row) + threadIdx.x * (EK_PT / 4);
#pragma unroll
        for (int k = 0; k < EK_PT / 4; k++) z[k] = make_uint4(0, 0, 0, 0);
    }
    const uint32_t r_lo = chunk_row[blockIdx.x], r_hi = chunk_row[blockIdx.x + 1];
    if (r_hi - r_lo > 4u * EK_CHUNK) {
        // A long stretch of EMPTY rows inside the chunk (the zero-in-degree tail of a class: millions of rows in front of the next
        // class's first edge): walking them to mark nothing kept one block busy for 2 ms.  A bisection per position instead.
#pragma unroll
        for (int k = 0; k < EK_PT; k++) {
            const uint64_t i = base + (uint64_t)threadIdx.x * EK_PT + k;
            uint32_t lo = r_lo, hi = r_hi + 1;                         // ptr[lo] <= i < ptr[hi] (positions past `last`: unused)
            while (hi - lo > 1) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if ((uint64_t)ptr[mid] <= i) lo = mid; else hi = mid;
            }
            rel[k] = lo - r_lo;
        }
        return r_lo;
    }
    __syncthreads();
    // (a row that holds or starts at the NEXT chunk's first position is in the range too: past `last`, skipped)
    for (uint64_t r = (uint64_t)r_lo + 1 + threadIdx.x; r <= r_hi; r += TPB) {
        const uint64_t at = ptr[r];
        if (at <= last) atomicMax(&s_row[at - base], (uint32_t)(r - r_lo));
    }
    __syncthreads();
    {
        const uint4* q = reinterpret_cast<const uint4*>(s_row) + threadIdx.x * (EK_PT / 4);
#pragma unroll
        for (int k = 0; k < EK_PT / 4; k++) {
            const uint4 v = q[k];
            rel[4 * k] = v.x; rel[4 * k + 1] = v.y; rel[4 * k + 2] = v.z; rel[4 * k + 3] = v.w;
        }
    }
#pragma unroll
    for (int k = 1; k < EK_PT; k++) rel[k] = max(rel[k], rel[k - 1]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = rel[EK_PT - 1];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl = max(incl, o);
    }
    if (lane == 63) s_w[wave] = incl;
    uint32_t before = __shfl_up(incl, 1);
    if (lane == 0) before = 0;
    __syncthreads();
    for (int w = 0; w < wave; w++) before = max(before, s_w[w]);
#pragma unroll
    for (int k = 0; k < EK_PT; k++) rel[k] = max(rel[k], before);
    return r_lo;
}

// Edge i of the out-edge CSR -> its parent (original id): the row of position i.
__global__ __launch_bounds__(TPB) void k_edge_parents(const uint64_t* __restrict__ out_ptr, const uint32_t* __restrict__ chunk_row, uint64_t e,
                                                      uint32_t* __restrict__ parent) {
    __shared__ __attribute__((aligned(16))) uint32_t s_row[EK_CHUNK];
    __shared__ uint32_t s_w[TPB / 64];
    const uint64_t base = (uint64_t)blockIdx.x * EK_CHUNK;
    if (base >= e) return;
    const uint64_t last = min(base + EK_CHUNK, e) - 1;
    uint32_t rel[EK_PT];
    const uint64_t r_lo = chunk_rows(out_ptr, chunk_row, base, last, s_row, s_w, rel);
    const uint64_t i0 = base + (uint64_t)threadIdx.x * EK_PT;
    if (i0 + EK_PT - 1 <= last) {
        uint4* o = reinterpret_cast<uint4*>(parent + i0);              // (base is a multiple of EK_CHUNK: 32-byte aligned)
#pragma unroll
        for (int k = 0; k < EK_PT / 4; k++)
            o[k] = make_uint4((uint32_t)r_lo + rel[4 * k], (uint32_t)r_lo + rel[4 * k + 1], (uint32_t)r_lo + rel[4 * k + 2], (uint32_t)r_lo + rel[4 * k + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < EK_PT; k++)
            if (i0 + k <= last) parent[i0 + k] = (uint32_t)r_lo + rel[k];
    }
}
// The in-edge lists of this rank's rows, in internal ids, from the edges sorted by ORIGINAL destination (srcs_by_dst: the
// parents of node v are srcs_by_dst[ptr_orig[v] .. ptr_orig[v + 1]), ascending original id): local row r is original node
// old_id[int_id(r)], its list moves as a piece and every parent is renamed.  One thread per local edge slot; the block finds the
// rows its chunk spans like k_edge_parents.
__global__ __launch_bounds__(TPB) void k_permute_rows(const uint32_t* __restrict__ in_ptr, const uint32_t* __restrict__ chunk_row, uint64_t e_local,
                                                      const uint32_t* __restrict__ old_id, uint32_t sl_nd, uint64_t id0_nd, uint64_t id0_d,
                                                      const uint64_t* __restrict__ ptr_orig, const uint32_t* __restrict__ srcs_by_dst,
                                                      const uint32_t* __restrict__ new_id, uint32_t* __restrict__ in_src) {
    __shared__ __attribute__((aligned(16))) uint32_t s_row[EK_CHUNK];
    __shared__ uint32_t s_w[TPB / 64];
    const uint64_t base = (uint64_t)blockIdx.x * EK_CHUNK;
    if (base >= e_local) return;
    const uint64_t last = min(base + EK_CHUNK, e_local) - 1;
    uint32_t rel[EK_PT];
    const uint32_t r_lo = chunk_rows(in_ptr, chunk_row, base, last, s_row, s_w, rel);
    // the rows go back to LDS and every thread takes slots TPB apart: a wave's loads and stores then fall on consecutive words
    // (with 8 consecutive slots per thread the lanes sat 32 bytes apart, 16 lines per load instead of 2: 2.2 ms instead of 0.8)
    {
        uint4* q = reinterpret_cast<uint4*>(s_row) + threadIdx.x * (EK_PT / 4);
#pragma unroll
        for (int k = 0; k < EK_PT / 4; k++) q[k] = make_uint4(rel[4 * k], rel[4 * k + 1], rel[4 * k + 2], rel[4 * k + 3]);
    }
    __syncthreads();
    uint32_t src[EK_PT];
    uint64_t off[EK_PT];                                               // srcs_by_dst index of slot i = off + i (mod 2^64)
#pragma unroll
    for (int k = 0; k < EK_PT; k++) {
        const uint64_t i = base + (uint64_t)k * TPB + threadIdx.x;
        const uint32_t lo = i <= last ? r_lo + s_row[k * TPB + threadIdx.x] : r_lo;
        const uint64_t iid = lo < sl_nd ? id0_nd + lo : id0_d + (lo - sl_nd);
        src[k] = old_id[iid];                                         // a real row: it has in-edges
        off[k] = (uint64_t)in_ptr[lo];
    }
#pragma unroll
    for (int k = 0; k < EK_PT; k++) off[k] = ptr_orig[src[k]] - off[k];
#pragma unroll
    for (int k = 0; k < EK_PT; k++) {
        const uint64_t i = base + (uint64_t)k * TPB + threadIdx.x;
        src[k] = i <= last ? srcs_by_dst[off[k] + i] : 0u;
    }
#pragma unroll
    for (int k = 0; k < EK_PT; k++) {
        const uint64_t i = base + (uint64_t)k * TPB + threadIdx.x;
        if (i <= last) in_src[i] = new_id[src[k]];
    }
}

__global__ void k_fill_u32(uint32_t* __restrict__ p, uint64_t n, uint32_t v) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// local in_ptr from the internal-id in_ptr (two contiguous id ranges -> one local row space)
__global__ void k_local_ptr(const uint64_t* __restrict__ in_ptr_int, uint32_t sl_nd, uint32_t sl_d,
                            uint64_t id0_nd, uint64_t id0_d, uint64_t off_nd, uint64_t off_d, uint64_t e_nd,
                            uint32_t* __restrict__ in_ptr_local) {
    uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t n_local = (uint64_t)sl_nd + sl_d;
    if (l > n_local) return;
    uint64_t v;
    if (l < sl_nd) v = in_ptr_int[id0_nd + l] - off_nd;
    else v = e_nd + in_ptr_int[id0_d + (l - sl_nd)] - off_d;   // l == n_local: id0_d + sl_d is valid (n_int+1 entries)
    in_ptr_local[l] = (uint32_t)v;
}

// bit 31 of the last in-edge entry of every row marks the row end (pagerank.hip walks rows by it)
__global__ void k_flag_row_ends(const uint32_t* __restrict__ in_ptr, uint64_t n_rows, uint32_t* __restrict__ in_src) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t a = in_ptr[r], b = in_ptr[r + 1];
    if (b > a) in_src[b - 1] |= 0x80000000u;
}

__global__ void k_gather_u32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n,
                             uint32_t* __restrict__ dst) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t j = idx[i];
    dst[i] = j == 0xFFFFFFFFu ? 0u : src[j];
}

// ---- ss_graph_apply_delta: replace the child lists of re-crawled pages in the resident out-edge CSR -----------------------
__global__ void k_delta_mark(const uint32_t* __restrict__ changed, uint64_t n_changed, uint64_t n_new, uint32_t* __restrict__ chg_idx,
                             uint32_t* __restrict__ err) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_changed; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = changed[i];
        if ((uint64_t)v >= n_new) { atomicOr(err, 1u); continue; }
        if (atomicExch(&chg_idx[v], (uint32_t)i) != 0xFFFFFFFFu) atomicOr(err, 2u);
    }
}
__global__ void k_delta_check_children(const uint32_t* __restrict__ c, uint64_t n, uint64_t n_new, uint32_t* __restrict__ err) {
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) bad = bad || (uint64_t)c[i] >= n_new;
    if (bad) atomicOr(err, 4u);
}
// out-degree of every node after the delta (entry n_new = 0: the scan then yields the new out_ptr)
__global__ void k_delta_degrees(const uint64_t* __restrict__ old_ptr, uint64_t n_old, uint64_t n_new, const uint32_t* __restrict__ chg_idx,
                                const uint64_t* __restrict__ nptr, uint64_t* __restrict__ deg) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > n_new) return;
    uint64_t d = 0;
    if (v < n_new) {
        const uint32_t ci = chg_idx[v];
        if (ci != 0xFFFFFFFFu) d = nptr[ci + 1] - nptr[ci];
        else if (v < n_old) d = old_ptr[v + 1] - old_ptr[v];
    }
    deg[v] = d;
}
// edge i of the new CSR: from the delta if its parent changed, from the old CSR otherwise
__global__ __launch_bounds__(TPB) void k_delta_edges(const uint64_t* __restrict__ ptr2, uint64_t n_new, uint64_t e_new, const uint64_t* __restrict__ old_ptr,
                                                     const uint32_t* __restrict__ old_dst, uint64_t n_old, const uint32_t* __restrict__ chg_idx,
                                                     const uint64_t* __restrict__ nptr, const uint32_t* __restrict__ children, uint32_t* __restrict__ dst2) {
    __shared__ uint64_t s_r[2];
    const uint64_t base = (uint64_t)blockIdx.x * EK_CHUNK;
    if (base >= e_new) return;
    const uint64_t last = min(base + EK_CHUNK, e_new) - 1;
    if (threadIdx.x < 2) {
        const uint64_t target = threadIdx.x == 0 ? base : last;
        uint64_t lo = 0, hi = n_new;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (ptr2[mid] <= target) lo = mid; else hi = mid;
        }
        s_r[threadIdx.x] = lo;
    }
    __syncthreads();
    const uint64_t r_lo = s_r[0], r_hi = s_r[1];
    for (int j = 0; j < EK_PT; j++) {
        const uint64_t i = base + (uint64_t)j * TPB + threadIdx.x;
        if (i > last) break;
        uint64_t lo = r_lo, hi = r_hi + 1;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (ptr2[mid] <= i) lo = mid; else hi = mid;
        }
        const uint64_t v = lo, off = i - ptr2[v];
        const uint32_t ci = chg_idx[v];
        dst2[i] = ci != 0xFFFFFFFFu ? children[nptr[ci] + off] : old_dst[old_ptr[v] + off];
    }
}

inline unsigned grid_for(uint64_t n, unsigned cap = 65535u * 16u) {
    uint64_t b = (n + TPB - 1) / TPB;
    if (b < 1) b = 1;
    return (unsigned)std::min<uint64_t>(b, cap);
}

// stable LSD radix sort of (key, value) pairs on the low `end_bit` bits of the key; enqueues only (tmp must outlive the sort)
// (rocprim's default hands up to 1M items to its merge sort — ~26 small kernels, 155 us for config 2's 1M nodes; from 128K items on
//  the radix passes are shorter.  Both are stable: the order is the same.)
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 128 * 1024>;
int32_t sort_pairs_u32(ss_ctx* ctx, uint32_t* k_in, uint32_t* k_out, uint32_t* v_in, uint32_t* v_out, uint64_t n, unsigned end_bit,
                       ss::DevBuf<char>& tmp) {
    if (n == 0) return SS_OK;
    size_t tmp_bytes = 0;
    SS_HIP(ctx, rocprim::radix_sort_pairs<SortConfig>(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)n, 0u, end_bit, ctx->stream));
    SS_HIP(ctx, tmp.alloc(tmp_bytes));
    SS_HIP(ctx, rocprim::radix_sort_pairs<SortConfig>(tmp.p, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)n, 0u, end_bit, ctx->stream));
    return SS_OK;
}
inline unsigned bits_for(uint64_t n) { unsigned b = 1; while (b < 32 && ((uint64_t)1 << b) < n) b++; return b; }

int32_t build(ss_graph* g, const uint64_t* out_ptr_in, const uint32_t* out_dst_in) {
    ss_ctx* ctx = g->ctx;
    hipStream_t st = ctx->stream;
    const bool trace = ctx->opt("pr.trace", 0) != 0;
    auto t_now = [] { return std::chrono::steady_clock::now(); };
    auto t_ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto tg0 = t_now();
    const uint64_t n = g->n, e = g->e;
    const uint32_t W = (uint32_t)g->world;
    g->settle();                                       // (a rebuild: the last build's temporaries go first)

    // the adjacency stays resident (ss_graph_apply_delta works on it); out_ptr_in == nullptr: g->out_ptr / g->out_dst hold it already
    ss::DevBuf<uint32_t> d_outdeg, d_indeg;
    ss::DevBuf<unsigned long long> d_cnt;
    ss::DevBuf<uint32_t> d_err;
    if (out_ptr_in) {
        SS_HIP(ctx, g->out_ptr.alloc(n + 1));
        SS_HIP(ctx, g->out_dst.alloc(e));
        SS_HIP(ctx, hipMemcpyAsync(g->out_ptr.p, out_ptr_in, (n + 1) * sizeof(uint64_t), hipMemcpyDefault, st));
        if (e) SS_HIP(ctx, hipMemcpyAsync(g->out_dst.p, out_dst_in, e * sizeof(uint32_t), hipMemcpyDefault, st));
    }
    const auto tga = t_now();
    const uint64_t* d_out_ptr = g->out_ptr.p;
    const uint32_t* d_out_dst = g->out_dst.p;
    SS_HIP(ctx, d_outdeg.alloc(n));
    SS_HIP(ctx, d_indeg.alloc(n + 1));
    SS_HIP(ctx, d_cnt.alloc(1));
    SS_HIP(ctx, d_err.alloc(1));
    SS_HIP(ctx, hipMemsetAsync(d_indeg.p, 0, (n + 1) * sizeof(uint32_t), st));
    SS_HIP(ctx, hipMemsetAsync(d_cnt.p, 0, sizeof(unsigned long long), st));
    SS_HIP(ctx, hipMemsetAsync(d_err.p, 0, sizeof(uint32_t), st));

    // the first/last offsets must frame out_dst exactly
    // (every read-back lands in the context's pinned scratch: see ss_ctx::h_pin)
    ctx->pin_used = 0;
    uint64_t* const hp_first = ctx->pin<uint64_t>(), * const hp_last = ctx->pin<uint64_t>();
    unsigned long long* const hp_nd = ctx->pin<unsigned long long>();
    uint32_t* const hp_err = ctx->pin<uint32_t>();
    SS_HIP(ctx, hipMemcpyAsync(hp_first, d_out_ptr, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipMemcpyAsync(hp_last, d_out_ptr + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));

    if (n) hipLaunchKernelGGL(k_outdeg, dim3(grid_for(n, 2048)), dim3(TPB), 0, st, d_out_ptr, n, d_outdeg.p, d_cnt.p, d_err.p);
    if (e) hipLaunchKernelGGL(k_check_dst, dim3(grid_for(e, 8192)), dim3(TPB), 0, st, d_out_dst, e, n, d_err.p);
    SS_HIP(ctx, hipMemcpyAsync(hp_nd, d_cnt.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipMemcpyAsync(hp_err, d_err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    const auto tgb = t_now();
    SS_HIP(ctx, hipStreamSynchronize(st));
    const uint64_t h_first = *hp_first, h_last = *hp_last;
    const unsigned long long h_nd = *hp_nd;
    const uint32_t h_err = *hp_err;
    if (trace) fprintf(stderr, "[pr trace]   upload: allocs + copies enqueued %.2f ms, rest enqueued %.2f ms, wait %.2f ms\n", t_ms(tg0, tga), t_ms(tga, tgb), t_ms(tgb, t_now()));
    if (h_first != 0 || h_last != e) return ctx->fail(SS_ERR_INVALID, "ss_graph_create: out_ptr[0]=%llu, out_ptr[n]=%llu, expected 0 and n_edges=%llu",
                                                      (unsigned long long)h_first, (unsigned long long)h_last, (unsigned long long)e);
    if (h_err & 1) return ctx->fail(SS_ERR_INVALID, "ss_graph_create: out_ptr is not non-decreasing");
    if (h_err & 2) return ctx->fail(SS_ERR_INVALID, "ss_graph_create: out_dst holds a node id >= n_nodes");

    g->n_nd = h_nd;
    const uint64_t n_d = n - h_nd;
    // world>1: every rank's all-gather piece ends in two extra (edge-less) rows that carry
    // its partial sums (contribution sum, L1 delta) — see pagerank.hip block_reduce_and_publish —, with TAIL_SUM_ROWS spare rows in
    // front of them (the two-vector form's per-topic sums travel there instead of in a collective of their own)
    g->sl_nd = (uint32_t)((h_nd + W - 1) / W) + (W > 1 ? 2u + TAIL_SUM_ROWS : 0u);
    g->sl_d = (uint32_t)((n_d + W - 1) / W);
    g->nd_int = (uint64_t)W * g->sl_nd;
    g->n_int = (uint64_t)W * ((uint64_t)g->sl_nd + g->sl_d);
    // bit 31 of an in-edge source id is the row-end flag
    if (g->n_int >= 0x7FFFFFFFull) return ctx->fail(SS_ERR_UNSUPPORTED, "ss_graph_create: too many nodes for 31-bit ids");
    // rows of class c dealt round-robin: rank r gets positions r, r+W, ...
    g->cnt_nd = (uint32_t)((h_nd + W - 1 - g->rank) / W);
    g->cnt_d = (uint32_t)((n_d + W - 1 - g->rank) / W);

    const auto tg1 = t_now();
    // ---- edges sorted by (original) destination: the in-degrees are the run lengths ------------------------------------------
    // One stable radix sort of (destination, parent) pairs over the bits the node ids need serves twice: its run lengths ARE the
    // in-degrees (a histogram by 50M atomics on 10M random words took 2.1 ms of the 6.4 ms this function needed at config 4), and
    // the sorted parents are the in-edge lists, which only have to move row by row into the internal order (k_permute_rows).
    // Inside a row the parents keep the order of the input CSR, i.e. ascending original id.
    ss::DevBuf<uint32_t> keys_a, keys_b, vals_a, vals_b;
    ss::DevBuf<char> sort_tmp1, sort_tmp2;
    SS_HIP(ctx, keys_a.alloc(n));                         // node sort keys; before that the run values of the edge sort (at most n runs)
    SS_HIP(ctx, keys_b.alloc(n));
    SS_HIP(ctx, vals_a.alloc(std::max<uint64_t>(n, e)));  // the edges' parents, then the node sort's values
    SS_HIP(ctx, vals_b.alloc(n));
    if (e >= 0xFFFFFFFFull) return ctx->fail(SS_ERR_UNSUPPORTED, "ss_graph_create: more than 2^32 - 2 edges");
    ss::DevBuf<uint32_t> e_dst, e_src;           // the sorted pairs (kept until the rows are permuted)
    ss::DevBuf<uint64_t> ptr_orig;               // [n + 1] exclusive scan of the in-degrees over original ids
    ss::DevBuf<uint32_t> chunk_row;              // [chunks + 1] the row that holds each chunk's first edge (k_chunk_first_rows)
    ss::DevBuf<uint32_t> n_runs;                 // (temporaries of the enqueued primitives live to the end of the function: no wait in between)
    ss::DevBuf<char> rle_tmp, scan_tmp1, scan_tmp2;
    SS_HIP(ctx, ptr_orig.alloc(n + 1));
    if (e) {
        SS_HIP(ctx, e_dst.alloc(e));
        SS_HIP(ctx, e_src.alloc(e));
        SS_HIP(ctx, chunk_row.alloc((size_t)ss::div_up(e, EK_CHUNK) + 1));
        hipLaunchKernelGGL(k_chunk_first_rows<uint64_t>, dim3(ss::div_up(n, TPB)), dim3(TPB), 0, st, (const uint64_t*)d_out_ptr, n, e, chunk_row.p);
        hipLaunchKernelGGL(k_edge_parents, dim3(ss::div_up(e, EK_CHUNK)), dim3(TPB), 0, st, (const uint64_t*)d_out_ptr, (const uint32_t*)chunk_row.p, e, vals_a.p);
        SS_TRY(sort_pairs_u32(ctx, const_cast<uint32_t*>(d_out_dst), e_dst.p, vals_a.p, e_src.p, e, bits_for(n), sort_tmp2));
        SS_HIP(ctx, n_runs.alloc(1));
        size_t tmp_bytes = 0;
        SS_HIP(ctx, rocprim::run_length_encode(nullptr, tmp_bytes, e_dst.p, (size_t)e, keys_a.p, keys_b.p, n_runs.p, st));
        SS_HIP(ctx, rle_tmp.alloc(tmp_bytes));
        SS_HIP(ctx, rocprim::run_length_encode(rle_tmp.p, tmp_bytes, e_dst.p, (size_t)e, keys_a.p, keys_b.p, n_runs.p, st));
        hipLaunchKernelGGL(k_scatter_runs, dim3(grid_for(std::min<uint64_t>(n, e), 4096)), dim3(TPB), 0, st, (const uint32_t*)keys_a.p, (const uint32_t*)keys_b.p,
                           (const uint32_t*)n_runs.p, d_indeg.p);
    }
    {
        size_t tmp_bytes = 0;
        auto in_it = rocprim::make_transform_iterator(d_indeg.p, [] __device__(uint32_t x) { return (uint64_t)x; });
        // (entry n of the scan reads one word past d_indeg: allocate n + 1 and keep the last at zero)
        SS_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp_bytes, in_it, ptr_orig.p, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), st));
        SS_HIP(ctx, scan_tmp1.alloc(tmp_bytes));
        SS_HIP(ctx, rocprim::exclusive_scan(scan_tmp1.p, tmp_bytes, in_it, ptr_orig.p, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), st));
    }
    const auto tg1b = t_now();
    // ---- node order ---------------------------------------------------------
    const uint32_t deg_bits = std::min(31u, bits_for(e + 1));
    if (n) hipLaunchKernelGGL(k_row_keys, dim3(ss::div_up(n, TPB)), dim3(TPB), 0, st, d_outdeg.p, d_indeg.p, n, deg_bits, keys_a.p, vals_a.p);
    SS_TRY(sort_pairs_u32(ctx, keys_a.p, keys_b.p, vals_a.p, vals_b.p, n, deg_bits + 1, sort_tmp1));

    ss::DevBuf<uint32_t> indeg_int;
    SS_HIP(ctx, g->new_id.alloc(n));
    SS_HIP(ctx, g->old_id.alloc(g->n_int));
    SS_HIP(ctx, indeg_int.alloc(g->n_int + 1));
    hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(g->n_int + 1, 4096)), dim3(TPB), 0, st, g->old_id.p, g->n_int, 0xFFFFFFFFu);
    SS_HIP(ctx, hipMemsetAsync(indeg_int.p, 0, (g->n_int + 1) * sizeof(uint32_t), st));
    if (n) hipLaunchKernelGGL(k_assign_ids, dim3(ss::div_up(n, TPB)), dim3(TPB), 0, st, (const uint32_t*)vals_b.p, n, (uint64_t)h_nd, W,
                              g->sl_nd, g->sl_d, d_indeg.p, g->new_id.p, g->old_id.p, indeg_int.p);

    // in_ptr over internal ids (exclusive scan of in-degrees, n_int+1 entries)
    ss::DevBuf<uint64_t> in_ptr_int;
    SS_HIP(ctx, in_ptr_int.alloc(g->n_int + 1));
    {
        size_t tmp_bytes = 0;
        auto in_it = rocprim::make_transform_iterator(indeg_int.p, [] __device__(uint32_t x) { return (uint64_t)x; });
        SS_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp_bytes, in_it, in_ptr_int.p, (uint64_t)0, (size_t)(g->n_int + 1),
                                            rocprim::plus<uint64_t>(), st));
        SS_HIP(ctx, scan_tmp2.alloc(tmp_bytes));
        SS_HIP(ctx, rocprim::exclusive_scan(scan_tmp2.p, tmp_bytes, in_it, in_ptr_int.p, (uint64_t)0, (size_t)(g->n_int + 1),
                                            rocprim::plus<uint64_t>(), st));
    }

    if (trace) SS_HIP(ctx, hipStreamSynchronize(st));
    const auto tg3 = t_now();
    // ---- this rank's rows -----------------------------------------------------
    const uint64_t id0_nd = (uint64_t)g->rank * g->sl_nd;
    const uint64_t id0_d = g->nd_int + (uint64_t)g->rank * g->sl_d;
    // The local in-degrees (sorted descending inside each class slice) go to the host run-length encoded.  Both encodings are
    // enqueued first and come back in ONE wait together with the four row offsets: the run counts and the first RLE_SPEC runs of
    // each class land in a pinned block (a graph has a few thousand distinct in-degrees; more than RLE_SPEC costs a second copy).
    // [The counts, then the runs, then the offsets each had a wait of their own: six round trips of 25-40 us, a third of config 2's
    //  0.9 ms here.]
    constexpr uint32_t RLE_SPEC = 8192;
    const uint32_t cnt[2] = {g->cnt_nd, g->cnt_d};
    const uint64_t id0[2] = {id0_nd, id0_d};
    ss_graph::SortedDegrees* dst[2] = {&g->h_indeg_nd, &g->h_indeg_d};
    ss::DevBuf<uint32_t> r_val[2], r_cnt[2], r_n;
    ss::DevBuf<char> r_tmp[2];
    size_t pin_cap = 0;
    const size_t pin_bytes = 64 + (size_t)4 * RLE_SPEC * sizeof(uint32_t);
    unsigned char* const pin_blk = static_cast<unsigned char*>(ctx->pin_alloc(pin_bytes, &pin_cap));
    if (!pin_blk) return ctx->fail(SS_ERR_OOM, "ss_graph_create: no pinned host memory for the read-backs");
    struct PinGuard { ss_ctx* c; void* p; size_t cap; ~PinGuard() { c->pin_free(p, cap); } } pin_guard{ctx, pin_blk, pin_cap};
    uint64_t* const h_ptr = reinterpret_cast<uint64_t*>(pin_blk);              // [4] row offsets
    uint32_t* const h_nruns = reinterpret_cast<uint32_t*>(pin_blk + 32);       // [2]
    uint32_t* const h_spec = reinterpret_cast<uint32_t*>(pin_blk + 64);        // [class][val | cnt][RLE_SPEC]
    SS_HIP(ctx, r_n.alloc(2));
    SS_HIP(ctx, hipMemsetAsync(r_n.p, 0, 2 * sizeof(uint32_t), st));
    for (int c = 0; c < 2; c++) {
        dst[c]->val.clear();
        dst[c]->start.assign(1, 0u);
        if (!cnt[c]) continue;
        SS_HIP(ctx, r_val[c].alloc(cnt[c]));
        SS_HIP(ctx, r_cnt[c].alloc(cnt[c]));
        size_t tmp_bytes = 0;
        SS_HIP(ctx, rocprim::run_length_encode(nullptr, tmp_bytes, indeg_int.p + id0[c], cnt[c], r_val[c].p, r_cnt[c].p, r_n.p + c, st));
        SS_HIP(ctx, r_tmp[c].alloc(tmp_bytes));
        SS_HIP(ctx, rocprim::run_length_encode(r_tmp[c].p, tmp_bytes, indeg_int.p + id0[c], cnt[c], r_val[c].p, r_cnt[c].p, r_n.p + c, st));
    }
    ss::DevBuf<unsigned char> d_pack;
    SS_HIP(ctx, d_pack.alloc(pin_bytes));
    hipLaunchKernelGGL(k_pack_readback, dim3(ss::div_up(RLE_SPEC, TPB)), dim3(TPB), 0, st, (const uint64_t*)in_ptr_int.p, id0_nd, id0_nd + g->sl_nd,
                       id0_d, id0_d + g->sl_d, (const uint32_t*)r_n.p, (const uint32_t*)r_val[0].p, (const uint32_t*)r_cnt[0].p,
                       (const uint32_t*)r_val[1].p, (const uint32_t*)r_cnt[1].p, RLE_SPEC, d_pack.p);
    SS_HIP(ctx, hipMemcpyAsync(pin_blk, d_pack.p, pin_bytes, hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    const uint64_t e_nd = h_ptr[1] - h_ptr[0], e_d = h_ptr[3] - h_ptr[2];
    g->e_local_nd = e_nd;
    g->e_local = e_nd + e_d;
    if (g->e_local >= 0xFFFFFFFFull) return ctx->fail(SS_ERR_UNSUPPORTED, "ss_graph_create: > 2^32-1 local edges");

    const uint32_t n_local = g->n_local();
    SS_HIP(ctx, g->in_ptr.alloc((size_t)n_local + 1));
    SS_HIP(ctx, g->in_src.alloc_streaming(g->e_local));
    SS_HIP(ctx, g->outdeg.alloc(g->sl_nd));
    hipLaunchKernelGGL(k_local_ptr, dim3(ss::div_up((uint64_t)n_local + 1, TPB)), dim3(TPB), 0, st, in_ptr_int.p, g->sl_nd,
                       g->sl_d, id0_nd, id0_d, h_ptr[0], h_ptr[2], e_nd, g->in_ptr.p);
    ss::DevBuf<uint32_t> chunk_row_in;
    if (g->e_local) {
        SS_HIP(ctx, chunk_row_in.alloc((size_t)ss::div_up(g->e_local, EK_CHUNK) + 1));
        hipLaunchKernelGGL(k_chunk_first_rows<uint32_t>, dim3(ss::div_up(n_local, TPB)), dim3(TPB), 0, st, (const uint32_t*)g->in_ptr.p, (uint64_t)n_local,
                           g->e_local, chunk_row_in.p);
    }
    if (g->e_local)
        hipLaunchKernelGGL(k_permute_rows, dim3(ss::div_up(g->e_local, EK_CHUNK)), dim3(TPB), 0, st, (const uint32_t*)g->in_ptr.p, (const uint32_t*)chunk_row_in.p, g->e_local,
                           (const uint32_t*)g->old_id.p, g->sl_nd, id0_nd, id0_d, (const uint64_t*)ptr_orig.p, (const uint32_t*)e_src.p,
                           (const uint32_t*)g->new_id.p, g->in_src.p);
    if (g->e_local)
        hipLaunchKernelGGL(k_flag_row_ends, dim3(ss::div_up((uint64_t)n_local, TPB)), dim3(TPB), 0, st, (const uint32_t*)g->in_ptr.p,
                           (uint64_t)n_local, g->in_src.p);
    if (g->sl_nd)
        hipLaunchKernelGGL(k_gather_u32, dim3(ss::div_up(g->sl_nd, TPB)), dim3(TPB), 0, st, d_outdeg.p, g->old_id.p + id0_nd,
                           (uint64_t)g->sl_nd, g->outdeg.p);
    SS_HIP(ctx, hipGetLastError());

    // (host side of the read-back, while the device permutes the rows)
    {
        std::vector<uint32_t> h_cnt;
        for (int c = 0; c < 2; c++) {
            if (!cnt[c]) continue;
            const uint32_t n_runs = h_nruns[c];
            if (n_runs > cnt[c]) return ctx->fail(SS_ERR_STATE, "ss_graph_create: internal: %u in-degree runs over %u rows", n_runs, cnt[c]);
            dst[c]->val.resize(n_runs);
            h_cnt.resize(n_runs);
            if (n_runs <= RLE_SPEC) {
                std::memcpy(dst[c]->val.data(), h_spec + (size_t)(2 * c) * RLE_SPEC, n_runs * sizeof(uint32_t));
                std::memcpy(h_cnt.data(), h_spec + (size_t)(2 * c + 1) * RLE_SPEC, n_runs * sizeof(uint32_t));
            } else {
                SS_HIP(ctx, hipMemcpyAsync(dst[c]->val.data(), r_val[c].p, n_runs * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                SS_HIP(ctx, hipMemcpyAsync(h_cnt.data(), r_cnt[c].p, n_runs * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                SS_HIP(ctx, hipStreamSynchronize(st));
            }
            dst[c]->start.resize((size_t)n_runs + 1);
            for (uint32_t j = 0; j < n_runs; j++) dst[c]->start[j + 1] = dst[c]->start[j] + h_cnt[j];
            if (dst[c]->start.back() != cnt[c]) return ctx->fail(SS_ERR_STATE, "ss_graph_create: internal: run-length encoded in-degrees cover %u of %u rows", dst[c]->start.back(), cnt[c]);
        }
    }
    // The kernels above are still running: what they read goes to the graph's late-free list instead of waiting here — the caller's
    // next step is host work (ss_pr_create builds its work items from the degree runs: 0.6 ms at config 2, 2 ms at config 4) and
    // everything that touches the graph's arrays is ordered behind them on the context's stream.  ss_graph::settle() frees them.
    if (ctx->opt("graph.late_free", 1) != 0) {
        g->defer(d_outdeg); g->defer(d_indeg); g->defer(d_cnt); g->defer(d_err);
        g->defer(keys_a); g->defer(keys_b); g->defer(vals_a); g->defer(vals_b); g->defer(sort_tmp1); g->defer(sort_tmp2);
        g->defer(e_dst); g->defer(e_src); g->defer(ptr_orig); g->defer(n_runs); g->defer(rle_tmp); g->defer(scan_tmp1); g->defer(scan_tmp2);
        g->defer(indeg_int); g->defer(in_ptr_int); g->defer(r_n);
        for (int c = 0; c < 2; c++) { g->defer(r_val[c]); g->defer(r_cnt[c]); g->defer(r_tmp[c]); }
        g->defer(d_pack); g->defer(chunk_row); g->defer(chunk_row_in);
    } else {
        SS_HIP(ctx, hipStreamSynchronize(st));
    }
    g->max_indeg = 0;
    if (g->cnt_nd) g->max_indeg = std::max(g->max_indeg, g->h_indeg_nd.val[0]);
    if (g->cnt_d) g->max_indeg = std::max(g->max_indeg, g->h_indeg_d.val[0]);
    SS_HIP(ctx, hipGetLastError());
    if (trace) { uint64_t pm = 0; double pms = 0; ss::pool_stats(&pm, &pms); fprintf(stderr, "[pr trace] pool: %llu hipMalloc so far, %.2f ms in them\n", (unsigned long long)pm, pms); }
    if (trace) fprintf(stderr, "[pr trace] ss_graph_create: upload + out-degrees %.2f ms, edge sort + in-degrees %.2f ms, node order %.2f ms, local rows %.2f ms\n", t_ms(tg0, tg1), t_ms(tg1, tg1b), t_ms(tg1b, tg3), t_ms(tg3, t_now()));
    return SS_OK;
}

}  // namespace

extern "C" {

int32_t ss_graph_create(ss_ctx* ctx, uint64_t n_nodes, uint64_t n_edges, const uint64_t* out_ptr,
                        const uint32_t* out_dst, int32_t rank, int32_t world, ss_graph** out) {
    if (!ctx) return SS_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!out) return ctx->fail(SS_ERR_INVALID, "ss_graph_create: out is NULL");
    *out = nullptr;
    if (!out_ptr || (n_edges && !out_dst)) return ctx->fail(SS_ERR_INVALID, "ss_graph_create: NULL adjacency");
    if (world < 1 || rank < 0 || rank >= world) return ctx->fail(SS_ERR_INVALID, "ss_graph_create: bad rank/world %d/%d", rank, world);
    if (n_nodes == 0) return ctx->fail(SS_ERR_INVALID, "ss_graph_create: empty node set");
    if (n_nodes >= 0xFFFFFFF0ull) return ctx->fail(SS_ERR_UNSUPPORTED, "ss_graph_create: n_nodes exceeds 32-bit ids");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    ss_graph* g = new (std::nothrow) ss_graph();
    if (!g) return ctx->fail(SS_ERR_OOM, "ss_graph_create: host OOM");
    g->ctx = ctx;
    g->n = n_nodes;
    g->e = n_edges;
    g->rank = rank;
    g->world = world;
    int32_t rc = build(g, out_ptr, out_dst);
    if (rc != SS_OK) {
        delete g;
        return rc;
    }
    *out = g;
    return SS_OK;
}

int32_t ss_graph_apply_delta(ss_graph* g, uint64_t n_nodes_new, uint64_t n_changed, const uint32_t* changed,
                             const uint64_t* new_ptr, const uint32_t* new_children) {
    if (!g) return SS_ERR_INVALID;
    ss_ctx* ctx = g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (g->users > 0) return ctx->fail(SS_ERR_STATE, "ss_graph_apply_delta: %d PageRank state(s) still use this graph", g->users);
    if (n_nodes_new < g->n) return ctx->fail(SS_ERR_INVALID, "ss_graph_apply_delta: the node set cannot shrink (%llu < %llu)", (unsigned long long)n_nodes_new, (unsigned long long)g->n);
    if (n_nodes_new >= 0xFFFFFFF0ull) return ctx->fail(SS_ERR_UNSUPPORTED, "ss_graph_apply_delta: n_nodes exceeds 32-bit ids");
    if (n_changed && (!changed || !new_ptr)) return ctx->fail(SS_ERR_INVALID, "ss_graph_apply_delta: NULL array with a non-zero count");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t n_old = g->n, n_new = n_nodes_new;
    std::vector<uint64_t> h_nptr(n_changed + 1, 0);
    if (n_changed) SS_HIP(ctx, ss::copy_in(st, h_nptr.data(), new_ptr, (n_changed + 1) * sizeof(uint64_t)));
    if (h_nptr[0] != 0) return ctx->fail(SS_ERR_INVALID, "ss_graph_apply_delta: new_ptr[0] != 0");
    for (uint64_t i = 0; i < n_changed; i++)
        if (h_nptr[i + 1] < h_nptr[i]) return ctx->fail(SS_ERR_INVALID, "ss_graph_apply_delta: new_ptr is not non-decreasing");
    const uint64_t e_add = h_nptr[n_changed];
    if (e_add && !new_children) return ctx->fail(SS_ERR_INVALID, "ss_graph_apply_delta: new_children is NULL");
    ss::DevBuf<uint32_t> d_changed, d_children, chg_idx, d_err;
    ss::DevBuf<uint64_t> d_nptr, deg, ptr2;
    SS_HIP(ctx, d_changed.alloc(n_changed));
    SS_HIP(ctx, d_children.alloc(e_add));
    SS_HIP(ctx, d_nptr.alloc(n_changed + 1));
    SS_HIP(ctx, chg_idx.alloc(n_new));
    SS_HIP(ctx, d_err.alloc(1));
    SS_HIP(ctx, deg.alloc(n_new + 1));
    SS_HIP(ctx, ptr2.alloc(n_new + 1));
    if (n_changed) SS_HIP(ctx, hipMemcpyAsync(d_changed.p, changed, n_changed * sizeof(uint32_t), hipMemcpyDefault, st));
    if (e_add) SS_HIP(ctx, hipMemcpyAsync(d_children.p, new_children, e_add * sizeof(uint32_t), hipMemcpyDefault, st));
    SS_HIP(ctx, hipMemcpyAsync(d_nptr.p, h_nptr.data(), (n_changed + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    SS_HIP(ctx, hipMemsetAsync(d_err.p, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(n_new, 4096)), dim3(TPB), 0, st, chg_idx.p, n_new, 0xFFFFFFFFu);
    if (n_changed) hipLaunchKernelGGL(k_delta_mark, dim3(grid_for(n_changed, 4096)), dim3(TPB), 0, st, (const uint32_t*)d_changed.p, n_changed, n_new, chg_idx.p, d_err.p);
    if (e_add) hipLaunchKernelGGL(k_delta_check_children, dim3(grid_for(e_add, 4096)), dim3(TPB), 0, st, (const uint32_t*)d_children.p, e_add, n_new, d_err.p);
    hipLaunchKernelGGL(k_delta_degrees, dim3(ss::div_up(n_new + 1, TPB)), dim3(TPB), 0, st, (const uint64_t*)g->out_ptr.p, n_old, n_new,
                       (const uint32_t*)chg_idx.p, (const uint64_t*)d_nptr.p, deg.p);
    {
        size_t tmp_bytes = 0;
        SS_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp_bytes, deg.p, ptr2.p, (uint64_t)0, (size_t)(n_new + 1), rocprim::plus<uint64_t>(), st));
        ss::DevBuf<char> tmp;
        SS_HIP(ctx, tmp.alloc(tmp_bytes));
        SS_HIP(ctx, rocprim::exclusive_scan(tmp.p, tmp_bytes, deg.p, ptr2.p, (uint64_t)0, (size_t)(n_new + 1), rocprim::plus<uint64_t>(), st));
        SS_HIP(ctx, hipStreamSynchronize(st));
    }
    ctx->pin_used = 0;
    uint64_t* const hp_enew = ctx->pin<uint64_t>();
    uint32_t* const hp_err = ctx->pin<uint32_t>();
    SS_HIP(ctx, hipMemcpyAsync(hp_enew, ptr2.p + n_new, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipMemcpyAsync(hp_err, d_err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    const uint64_t e_new = *hp_enew;
    const uint32_t h_err = *hp_err;
    if (h_err & 1) return ctx->fail(SS_ERR_INVALID, "ss_graph_apply_delta: a changed node id is >= n_nodes_new (graph unchanged)");
    if (h_err & 2) return ctx->fail(SS_ERR_INVALID, "ss_graph_apply_delta: the same node is listed twice in `changed` (graph unchanged)");
    if (h_err & 4) return ctx->fail(SS_ERR_INVALID, "ss_graph_apply_delta: a new child id is >= n_nodes_new (graph unchanged)");
    ss::DevBuf<uint32_t> dst2;
    SS_HIP(ctx, dst2.alloc(e_new));
    if (e_new)
        hipLaunchKernelGGL(k_delta_edges, dim3(ss::div_up(e_new, EK_CHUNK)), dim3(TPB), 0, st, (const uint64_t*)ptr2.p, n_new, e_new, (const uint64_t*)g->out_ptr.p,
                           (const uint32_t*)g->out_dst.p, n_old, (const uint32_t*)chg_idx.p, (const uint64_t*)d_nptr.p, (const uint32_t*)d_children.p, dst2.p);
    SS_HIP(ctx, hipGetLastError());
    // the new adjacency replaces the old one, and the layout is built again from it on the device (no upload: the cost is
    // that of ss_graph_create on resident arrays, 7 ms at 10M nodes / 50M edges)
    ss_graph old_layout;                                   // keeps the old layout alive until the new one stands
    std::swap(old_layout.out_ptr, g->out_ptr);
    std::swap(old_layout.out_dst, g->out_dst);
    g->out_ptr = std::move(ptr2);
    g->out_dst = std::move(dst2);
    const uint64_t n_keep = g->n, e_keep = g->e;
    g->n = n_new;
    g->e = e_new;
    const int32_t rc = build(g, nullptr, nullptr);
    if (rc != SS_OK) {                                     // cannot happen after the checks above; keep the graph usable anyway
        std::swap(old_layout.out_ptr, g->out_ptr);
        std::swap(old_layout.out_dst, g->out_dst);
        g->n = n_keep;
        g->e = e_keep;
        (void)build(g, nullptr, nullptr);
    }
    return rc;
}

int32_t ss_graph_get_info(const ss_graph* g, ss_graph_info* info) {
    if (!g || !info) return SS_ERR_INVALID;
    info->n_nodes = g->n;
    info->n_edges = g->e;
    info->n_nondangling = g->n_nd;
    info->n_rows_local = (uint64_t)g->cnt_nd + g->cnt_d;
    info->n_edges_local = g->e_local;
    info->max_indeg = g->max_indeg;
    info->rank = g->rank;
    info->world = g->world;
    return SS_OK;
}

int32_t ss_graph_destroy(ss_graph* g) {
    if (!g) return SS_ERR_INVALID;
    ss_ctx* ctx = g->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    g->settle();
    delete g;
    return SS_OK;
}

}  // extern "C"
