// index.hpp — one device-resident inverted table (inv[0] title / inv[1] body of the reference,
// database/database.go:85-99) as a term-major CSR over dense ids.
//
// HBM layout:  term_ptr u64[T+1] | post_doc u32[P] (strictly ascending inside a term) |
//              post_w f32[P] (normalised tf until ss_tfidf_build, tf*idf after) | mag f64[n_docs]
// Host keeps df per term (u32[T]) so that query planning needs no device round trip.
#pragma once
#include "common.hpp"

struct ss_index {
    ss_ctx* ctx = nullptr;
    uint64_t n_docs = 0, n_terms = 0, n_post = 0;
    ss::DevBuf<uint64_t> term_ptr;
    ss::DevBuf<uint32_t> post_doc;
    ss::DevBuf<float> post_w;
    ss::DevBuf<double> mag;        // sqrt(sum w^2) per doc, valid once `weighted`
    ss::DevBuf<double> mag2;       // sum w^2 per doc (before the square root): what an incremental update adds to / subtracts from
    bool mag2_valid = false;       // mag2 matches the table (set by ss_tfidf_build / ss_index_refresh_magnitudes, not by ss_index_set_weighted)
    ss::DevBuf<uint64_t> pos_ptr;  // [P+1] positional postings (phrase search), optional
    ss::DevBuf<float> pos;         // positions as stored by the reference: float32, -100 = anchor/meta text
    ss::DevBuf<uint64_t> df_global; // [T] whole-corpus document frequencies when this table is one doc-range shard (optional)
    bool has_df_global = false;
    std::vector<uint64_t> h_term_ptr;  // host copy for query planning
    bool weighted = false;
    int users = 0;                 // scorers holding this index
};
