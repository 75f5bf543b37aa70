// graph.hpp — device-resident link graph in the layout the PageRank kernels want.
//
// Reference: ranking/pagerank.go:17-44 builds map[parent][]child and the node
// set (parents U children).  Here the Go shim hands over the same adjacency as
// an out-edge CSR over dense ids and this module rebuilds it, on the device, as
//
//   * an internal node numbering:  [ non-dangling nodes | dangling nodes ]
//       - only non-dangling nodes ever contribute rank (pagerank.go:131-134),
//         so the gathered "contribution table" is compact (n_nd rows) and, with
//         several GPUs, only that part is exchanged;
//       - inside each class rows are sorted by in-degree (descending) and dealt
//         round-robin to the `world` ranks, so every rank's slice is contiguous
//         (one all-gather piece), edge-balanced, and degree-binned
//         (block-per-row / wave-per-row / lane-group-per-row / no-edge rows);
//   * in-edge lists (pull form) of this rank's rows, sources in internal ids,
//     sorted by (row, source): the SpMV needs no atomics and is deterministic.
#pragma once
#include "common.hpp"

// world > 1: rows of a rank's all-gather piece in front of its two tail rows that can carry further partial sums of the rank (the
// two-vector form's per-topic L1 sums ride there: 32 rows of two doubles = SS_MAX_TOPICS sums); zero unless somebody writes them
constexpr uint32_t TAIL_SUM_ROWS = 32;

struct ss_graph {
    ss_ctx* ctx = nullptr;
    uint64_t n = 0, e = 0;
    int rank = 0, world = 1;
    uint64_t n_nd = 0;            // nodes with out-degree > 0
    uint32_t sl_nd = 0, sl_d = 0; // rows per rank slice (padded), per class
    uint32_t cnt_nd = 0, cnt_d = 0; // real rows of this rank, per class
    uint64_t nd_int = 0;          // world * sl_nd: rows of the contribution table
    uint64_t n_int = 0;           // world * (sl_nd + sl_d)
    uint64_t e_local = 0, e_local_nd = 0;
    uint32_t max_indeg = 0;

    // the adjacency as the host handed it over (out-edge CSR over original ids): kept resident so that a re-crawled page's
    // child list can be replaced on the device (ss_graph_apply_delta) without a new upload
    ss::DevBuf<uint64_t> out_ptr;  // [n+1]
    ss::DevBuf<uint32_t> out_dst;  // [e]
    int users = 0;                 // ss_pr states on this graph
    ss::DevBuf<uint32_t> new_id;  // [n]     original id -> internal id
    ss::DevBuf<uint32_t> old_id;  // [n_int] internal id -> original id (0xFFFFFFFF = padding row)
    // local rows: lrow in [0, sl_nd) = non-dangling slice, [sl_nd, sl_nd+sl_d) = dangling slice
    ss::DevBuf<uint32_t> in_ptr;  // [sl_nd + sl_d + 1]
    ss::DevBuf<uint32_t> in_src;  // [e_local] internal ids, all < nd_int
    ss::DevBuf<uint32_t> outdeg;  // [sl_nd] out-degree of the local non-dangling rows
    // The local in-degrees, sorted descending per class, as the HOST sees them for work-table building: run-length encoded on the
    // device (a few thousand distinct values at 10M rows: a few KB over PCIe instead of 40 MB, and a table the host's searches
    // find in its L1 instead of a 40 MB array they miss in).  val[j] = in-degree of rows [start[j], start[j + 1]).
    struct SortedDegrees {
        std::vector<uint32_t> val, start;     // start has val.size() + 1 entries; start.back() = number of rows
        size_t size() const { return start.empty() ? 0 : start.back(); }
        size_t run_of(size_t i) const {        // the run that holds row i (i < size())
            size_t lo = 0, hi = val.size();
            while (hi - lo > 1) {
                const size_t mid = (lo + hi) >> 1;
                if (start[mid] <= i) lo = mid; else hi = mid;
            }
            return lo;
        }
        uint32_t operator[](size_t i) const { return val[run_of(i)]; }
        // number of rows with in-degree > lim = index of the first row whose in-degree is <= lim
        uint32_t first_at_most(uint32_t lim) const {
            size_t lo = 0, hi = val.size();   // first run with val <= lim (val is strictly descending)
            while (lo < hi) {
                const size_t mid = (lo + hi) >> 1;
                if (val[mid] > lim) lo = mid + 1; else hi = mid;
            }
            return start.empty() ? 0u : start[lo];
        }
    };
    SortedDegrees h_indeg_nd, h_indeg_d;

    // Temporaries of the build that its last kernels may still be reading when ss_graph_create returns (option "graph.late_free"):
    // freed by settle() — at the graph's next use that allocates (ss_pr_create after its host work, ss_graph_apply_delta) or when
    // the graph goes.  Everything that reads the graph's arrays runs on the context's stream behind those kernels.
    std::vector<void*> late_free;
    template <typename T>
    void defer(ss::DevBuf<T>& b) { if (b.p) late_free.push_back(b.detach()); }
    void settle() {
        if (late_free.empty()) return;
        ss::pool_free_batch(late_free.data(), late_free.size());
        late_free.clear();
    }
    ~ss_graph() { settle(); }

    uint32_t n_local() const { return sl_nd + sl_d; }
    // internal id of local row
    uint64_t int_id(uint32_t lrow) const {
        return lrow < sl_nd ? (uint64_t)rank * sl_nd + lrow
                            : nd_int + (uint64_t)rank * sl_d + (lrow - sl_nd);
    }
};
