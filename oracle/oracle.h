/*
 * oracle.h — CPU restatement of SpaghettiSearch's ranking hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (Go) has no tests, golden vectors or fixtures
 * for this path and cannot be built in this image (no Go
 * toolchain, un-vendored badger dependency).  This restatement is pinned only
 * by (i) hand-derived known-answer tests written from the Go source
 * (tests/test_oracle_kat.py), (ii) an independently written numpy restatement
 * (oracle/oracle_np.py) that must agree with it, and (iii) exact-rational
 * recomputation of the PageRank recurrence (tests/test_oracle_kat.py).
 *
 * Every function cites the reference file:line it follows
 * (paths relative to the reference root).
 */
#ifndef SPAGHETTI_ORACLE_H
#define SPAGHETTI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One result row, same POD as ss_hit in include/spaghetti_rank.h. */
typedef struct orc_hit {
    uint32_t doc;
    uint32_t _pad;
    double title;    /* cosine-normalised title score  (get_metadata.go:58,64-66) */
    double body;     /* cosine-normalised body score   (get_metadata.go:57,61-63) */
    double pagerank; /* sqd = sum_t topicProbs[t]*PR[doc][t] (get_metadata.go:39-42,68) */
    double final;    /* (0.33*sqd+0.38*title+0.29*body)*100 (get_metadata.go:69) */
} orc_hit;

/* Go's math.Log2 / math.Log (pure-Go amd64 path), restated: FreeBSD e_log.c
 * algorithm as published in the Go standard library (go1.12, src/math/log.go,
 * src/math/log10.go).  Used by term_weighting.go:37. */
double orc_go_log(double x);
double orc_go_log2(double x);
/* Sensitivity of float32(orc_go_log2(total_docs/df)) to last-ulp differences between libm
 * implementations (see oracle.c); out[0..3] = mismatches vs correctly rounded, sensitive inputs,
 * undecidable inputs, max ulp error. */
int orc_log2_sensitivity(double total_docs, uint64_t df_lo, uint64_t df_hi, double margin_ulps,
                         uint64_t out[4], uint64_t* first_bad_df);

/*
 * ranking/pagerank.go:85-145 (updatePagerank + computeRankInherited) for ONE
 * topic, over dense integer ids.  Graph = out-edge CSR (forw[2]: parent ->
 * children; node set = parents U children, pagerank.go:24-44).
 *   n_init   = int(numPages) of the topic (pagerank.go:61,104-105)
 *   eps      = convergenceCriterion; loop runs while lastChange > eps
 *   max_iter = 0: unbounded (reference behaviour); >0: stop after that many
 *              iterations even if not converged (benchmark extension)
 * Outputs: rank[N], *iters = iterations executed, *last_change, *last_total.
 */
int orc_pagerank_topic(uint64_t n_nodes, const uint64_t* out_ptr, const uint32_t* out_dst,
                       double d, double eps, int32_t max_iter, int32_t n_init,
                       double* rank, int32_t* iters, double* last_change, double* last_total);

/* OPT-IN extensions, see oracle.c: a topic's teleport set; computeTopicProbs (main_retrieve.go:106-159)
 * as written (mode 0: all zero, the `var probs float64` quirk) and as intended (mode 1). */
int orc_pagerank_topic_ts(uint64_t n_nodes, const uint64_t* out_ptr, const uint32_t* out_dst,
                          double d, double eps, int32_t max_iter, int32_t n_init,
                          const uint8_t* member, uint64_t n_members,
                          double* rank, int32_t* iters, double* last_change, double* last_total);
int orc_topic_probs(int32_t k_topics, const double* word_count, int32_t n_tok, const uint32_t* tok_ptr,
                    const uint32_t* tok_cat, const double* tok_freq, const uint8_t* tok_missing, int32_t mode, double* probs_out);

/* pagerank.go:54-63: the sequential per-category loop. rank_out is [K][N] (topic-major). */
int orc_pagerank(uint64_t n_nodes, const uint64_t* out_ptr, const uint32_t* out_dst,
                 double d, double eps, int32_t max_iter, int32_t k_topics,
                 const int32_t* n_topic, double* rank_out, int32_t* iters_out);

/* Same arithmetic, but keyed by 32-char hex strings in chained hash maps, the
 * way pagerank.go keys Go maps by md5-hex docHash.  Only used as the
 * "reference-shaped" CPU baseline; must agree with orc_pagerank_topic. */
int orc_pagerank_topic_hashed(uint64_t n_nodes, const uint64_t* out_ptr, const uint32_t* out_dst,
                              double d, double eps, int32_t max_iter, int32_t n_init,
                              double* rank, int32_t* iters);

/* "Strong CPU" baselines: same arithmetic, flat arrays, OpenMP over all host cores.
 * PageRank in pull form over in-edge lists (sums in a different order: agrees to ~1e-15). */
int orc_pagerank_topic_omp(uint64_t n_nodes, const uint64_t* out_ptr, const uint32_t* out_dst,
                           double d, double eps, int32_t max_iter, int32_t n_init,
                           double* rank, int32_t* iters, int32_t* threads_used);
int orc_score_topk_batch_omp(uint64_t n_docs, uint64_t n_terms,
                             const uint64_t* t_ptr, const uint32_t* t_doc, const float* t_w,
                             const uint64_t* b_ptr, const uint32_t* b_doc, const float* b_w,
                             const double* mag_title, const double* mag_body,
                             int32_t k_topics, const double* prior, const double* topic_probs,
                             int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms,
                             const int32_t* query_len, int32_t k, orc_hit* hits, int32_t* n_hits, int32_t* threads_used);

/*
 * ranking/term_weighting.go:10-57 for one inverted table (term-major CSR).
 *   post_w: in = normalised tf (indexer.go:362), out = tf*idf (float32, :42)
 *   total_docs = len(forw[3]) = number of PageRank nodes (:13-17)
 *   mag2[n_docs] += float64(float32(w*w)) (:44) — caller zeroes it
 *   idf_out (nullable) [n_terms]
 * ranking/term_weighting.go:72,97,105: magnitude = sqrt(mag2) -> orc_sqrt_inplace.
 */
int orc_tfidf(uint64_t n_terms, const uint64_t* term_ptr, const uint32_t* post_doc,
              float* post_w, double total_docs, uint64_t n_docs, double* mag2, float* idf_out);
void orc_sqrt_inplace(uint64_t n, double* v);

/*
 * retrieval/main_retrieve.go:50-103 + get_metadata.go:31-69 + util.go:48-54
 * for one query (non-phrase terms).
 *   title / body index: term-major CSR with tf*idf weights
 *   q_terms[n_q_terms]: term ids in query order, duplicates kept
 *       (main_retrieve.go:29-36); id >= n_terms = unknown term (ErrKeyNotFound, :193,:218)
 *   query_len = len(queryTokenised)+len(phraseTokenised) (main_retrieve.go:90)
 *   mag_title/mag_body[n_docs]: forw[4] values, 0 where the key is missing (Q8)
 *   prior [n_docs][k_topics] node-major (forw[3]) and topic_probs[k_topics], both nullable
 *       (nil topicProbs => sqd = 0, main_retrieve.go:88)
 *   extra_docs/extra_title/extra_body (nullable, n_extra): pre-aggregated phrase
 *       contributions merged at main_retrieve.go:73-78; has_title/has_body flags
 *       in extra_flags bit0/bit1.
 *   k: result cut (reference: 50, main_retrieve.go:99-100)
 * Order: FinalRank descending (util.go:49); ties by ascending doc id (a valid
 * linearisation of the reference's arrival-order-dependent tie order); NaN
 * finals last.  Returns number of hits written (<= k) in *n_hits, total
 * candidates in *n_cand.
 */
int orc_score_topk(uint64_t n_docs, uint64_t n_terms,
                   const uint64_t* t_ptr, const uint32_t* t_doc, const float* t_w,
                   const uint64_t* b_ptr, const uint32_t* b_doc, const float* b_w,
                   const double* mag_title, const double* mag_body,
                   int32_t k_topics, const double* prior, const double* topic_probs,
                   const uint32_t* q_terms, int32_t n_q_terms, int32_t query_len,
                   int32_t n_extra, const uint32_t* extra_docs, const float* extra_title,
                   const float* extra_body, const uint8_t* extra_flags,
                   int32_t k, orc_hit* hits, int32_t* n_hits, uint64_t* n_cand);

/* Batch wrapper: q_ptr[n_q+1] into q_terms; query_len[i] nullable (defaults to
 * the term count); topic_probs [n_q][k_topics] nullable. hits [n_q][k]. */
int orc_score_topk_batch(uint64_t n_docs, uint64_t n_terms,
                         const uint64_t* t_ptr, const uint32_t* t_doc, const float* t_w,
                         const uint64_t* b_ptr, const uint32_t* b_doc, const float* b_w,
                         const double* mag_title, const double* mag_body,
                         int32_t k_topics, const double* prior, const double* topic_probs,
                         int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms,
                         const int32_t* query_len, int32_t k, orc_hit* hits, int32_t* n_hits);

/* "Reference-shaped" scoring baseline: main_retrieve.go:61-97 + get_metadata.go:46-69 +
 * util.go:48-54 with the reference's data structures — string-keyed hash maps, appended weight slices,
 * insertion-sort appendSort of Rank_combined-sized rows.  No PageRank blend (topicProbs is nil in the reference,
 * main_retrieve.go:88).  hits[].doc/.final only; equal finals in map order (not comparable with orc_score_topk).
 * orc_magmap = forw[4] as a docHash-keyed map, built once outside the timed region. */
typedef struct orc_magmap orc_magmap;
orc_magmap* orc_magmap_build(uint64_t n_docs, const double* mag_title, const double* mag_body);
void orc_magmap_free(orc_magmap* m);
int orc_score_topk_batch_hashed(const orc_magmap* mags, uint64_t n_terms,
                                const uint64_t* t_ptr, const uint32_t* t_doc, const float* t_w,
                                const uint64_t* b_ptr, const uint32_t* b_doc, const float* b_w,
                                int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms, const int32_t* query_len,
                                int32_t k, int32_t use_threads, orc_hit* hits, int32_t* n_hits, int32_t* threads_used);

/*
 * retrieval/phrase.go:11-170 + util.go:162-203 for one phrase (all quoted
 * phrases of a query are concatenated into ONE phrase, main_retrieve.go:26).
 *   pos_ptr[P+1] into pos[]: positions (float32, parser.go:195-207; -100 = anchor/meta)
 *   of each posting, per table.
 * Output: docs that contain the phrase in body and/or title, with the summed
 * float32 weights (phrase.go:59,69,73,83,90) and flags bit0=title,bit1=body.
 * out arrays must hold min(df of first term) entries... caller passes cap.
 */
int orc_phrase(uint64_t n_terms,
               const uint64_t* t_ptr, const uint32_t* t_doc, const float* t_w,
               const uint64_t* t_pos_ptr, const float* t_pos,
               const uint64_t* b_ptr, const uint32_t* b_doc, const float* b_w,
               const uint64_t* b_pos_ptr, const float* b_pos,
               const uint32_t* phrase_terms, int32_t n_phrase,
               int32_t cap, uint32_t* out_docs, float* out_title, float* out_body,
               uint8_t* out_flags, int32_t* n_out);

#ifdef __cplusplus
}
#endif
#endif
