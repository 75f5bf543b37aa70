/*
 * spaghetti_rank.h — C ABI of the MI355X-native ranking hot path.
 *
 * This is the drop-in boundary for SpaghettiSearch's ranking path.  The
 * reference has no FFI/plugin layer: its boundary is three exported Go
 * functions.  A cgo shim that keeps those three signatures
 * (go/ranking, go/retrieval) binds exactly the entry points
 * declared here; each entry point cites the reference code it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns int32 status, 0 = SS_OK; ss_last_error() gives text.
 *     The Go shim turns any non-zero status into panic(err), matching the
 *     reference's error policy (pagerank.go:20,29,49,57,76,81).
 *   - input pointers may be host OR device memory (copied with
 *     hipMemcpyDefault before the call returns: cgo forbids retaining Go
 *     pointers).  Output pointers likewise, caller-allocated.
 *   - opaque handles are freed by the matching *_destroy.
 *   - one ss_ctx per GPU per process (one process per GPU for multi-GPU).
 *   - compute entry points are thread-safe on a shared handle
 *     (retrieval.Retrieve is called from one goroutine per HTTP request,
 *     cmd/server/server.go:47); calls on one ctx are serialised internally.
 *   - there is NO CPU fallback: without a gfx950 device ss_init fails.
 */
#ifndef SPAGHETTI_RANK_H
#define SPAGHETTI_RANK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4 (round 5): ss_score_topk_submit / ss_score_topk_collect exist; ss_graph_create may return while its last build kernels
 * still run on the context's stream (everything that reads the graph is ordered behind them); "score.pipeline" defaults to 2
 * internal wave streams and ss_last_kernel_ms(1) is the device time of the last scoring call's kernels on the stream that
 * carries its merge.  A binding checks ss_abi_version() == SS_ABI_VERSION at load. */
#define SS_ABI_VERSION 4

enum {
    SS_OK = 0,
    SS_ERR_INVALID = 1,      /* bad argument (null handle, k<=0, ids out of range, ...) */
    SS_ERR_NO_DEVICE = 2,    /* no usable gfx950 device / HIP runtime error at init */
    SS_ERR_HIP = 3,          /* HIP runtime error (text in ss_last_error) */
    SS_ERR_OOM = 4,          /* device or host allocation failed */
    SS_ERR_UNSORTED = 5,     /* a posting list is not strictly ascending by doc id */
    SS_ERR_STATE = 6,        /* call sequence error (e.g. scoring before tfidf build) */
    SS_ERR_UNSUPPORTED = 7,  /* e.g. k > SS_MAX_TOPK, k_topics > SS_MAX_TOPICS */
    SS_ERR_COMM = 8          /* RCCL error (text in ss_last_error) */
};

#define SS_MAX_TOPK 1024     /* largest k accepted by ss_score_topk */
#define SS_MAX_TOPICS 64     /* largest k_topics accepted by every entry point (also by the opt-in two-vector form, "pr.affine") */
#define SS_MAX_QUERY_TERMS 64
#define SS_UNKNOWN_TERM 0xFFFFFFFFu /* term id for "key not found" (main_retrieve.go:193,218) */

typedef struct ss_ctx ss_ctx;
typedef struct ss_graph ss_graph;
typedef struct ss_pr ss_pr;
typedef struct ss_index ss_index;
typedef struct ss_scorer ss_scorer;

/* Result row.  The Go shim maps it into retrieval.Rank_combined
 * (retrieval/util.go:25-36: PageRank, FinalRank) and decorates only the k
 * winners with DocInfo/summary (get_metadata.go:79-235 stays host-side Go). */
typedef struct ss_hit {
    uint32_t doc;      /* dense doc id assigned by the shim (md5-hex docHash <-> id) */
    uint32_t _pad;
    double title;      /* TitleRank after cosine normalisation, get_metadata.go:58,64-66 */
    double body;       /* BodyRank  after cosine normalisation, get_metadata.go:57,61-63 */
    double pagerank;   /* sqd, get_metadata.go:39-42,68 */
    double final;      /* FinalRank, get_metadata.go:69 */
} ss_hit;

typedef struct ss_graph_info {
    uint64_t n_nodes, n_edges;
    uint64_t n_nondangling;      /* nodes with out-degree > 0 */
    uint64_t n_rows_local;       /* destination rows owned by this rank */
    uint64_t n_edges_local;      /* in-edges of those rows */
    uint32_t max_indeg;
    int32_t rank, world;
} ss_graph_info;

/* ---- context ---------------------------------------------------------- */
int32_t ss_abi_version(void);
int32_t ss_init(int32_t device_id, ss_ctx** out);
int32_t ss_shutdown(ss_ctx* ctx);
/* Use the caller's HIP stream (e.g. torch's current stream) for all work of this ctx.
 * NULL = the library's own non-blocking stream (default).  DEVICE pointers passed to any entry
 * point must hold their final contents with respect to the ctx stream: share the producer's
 * stream here, or synchronise the producer first.  Host pointers need nothing. */
int32_t ss_set_stream(ss_ctx* ctx, void* hip_stream);
int32_t ss_synchronize(ss_ctx* ctx);
/* Tuning and diagnostic switches of one context (no reference counterpart; nothing reads the environment).  The
 * defaults are the measured best; tests use the switches to reach kernel variants that the defaults would not pick
 * at test sizes ("pr.force_narrow", "tfidf.bucket_min", "score.exact_all", "score.separate_merge"), experiments to
 * sweep a parameter.  Unknown names are SS_ERR_INVALID; SS_OPTION_DEFAULT restores the default.  An option is read
 * when the object it concerns is created or the call it concerns is made.
 * "score.pipeline" (default 2 internal streams; 0 = every scoring kernel on the context's stream): ss_score_topk calls whose outputs
 * are device buffers run their scoring kernel on an internal stream and only the per-query merge — the kernel that writes the
 * hits — on the context's stream, behind an event: the next batch's scoring starts under this batch's merge and tail.  Batches of
 * the wave kernel since round 4; batches that are all k_score_slices (short lists, many terms, quoted phrases) too unless
 * "score.pipeline_slices" = 0.  Nothing changes for the caller: the hits are complete in the order of the context's stream,
 * and a consumer enqueued between two calls sees the first call's hits. */
#define SS_OPTION_DEFAULT INT64_MIN
int32_t ss_set_option(ss_ctx* ctx, const char* name, int64_t value);
const char* ss_last_error(ss_ctx* ctx); /* ctx may be NULL: last global error */

/* ---- multi-GPU: one context (= one GPU) per rank, collectives inside the library (RCCL over xGMI) -----------
 * The reference has no distributed code; this is what parallelises the TODO at ranking/pagerank.go:52 across GPUs.
 * Rank 0 makes the 128-byte id and hands it to every rank by any channel the host has (file, pipe, environment);
 * every rank then joins with its own context.  ss_comm_init blocks until all `world` ranks have called it.  One
 * process per GPU is the intended shape; a process holding several contexts calls ss_comm_init for each from its
 * own thread.  Collectives run on the context's stream. */
#define SS_COMM_ID_BYTES 128
int32_t ss_comm_unique_id(void* id_out /*[SS_COMM_ID_BYTES]*/);
int32_t ss_comm_init(ss_ctx* ctx, const void* id /*[SS_COMM_ID_BYTES]*/, int32_t rank, int32_t world);
/* 2-D decomposition (topic groups x doc shards): split the ranks into groups by color, renumbered by key inside a group; the
 * context's communicator becomes the group's.  A host with 8 GPUs and 16 topics can then run two groups of four doc shards,
 * each on 8 topics: a rank receives 3/4 of HALF the contribution table per sweep instead of 7/8 of all of it.
 *   ss_comm_split(ctx, rank / 4, rank % 4); ss_graph_create(..., rank % 4, 4); ss_pagerank_run_sharded(g, ..., 8, n_topic + 8 * (rank / 4), ...) */
int32_t ss_comm_split(ss_ctx* ctx, int32_t color, int32_t key);
int32_t ss_comm_destroy(ss_ctx* ctx);
int32_t ss_comm_info(ss_ctx* ctx, int32_t* rank_out /* -1: no communicator */, int32_t* world_out);
/* The exchange steps of the index side: whole-corpus document frequencies = all-reduce(sum) of the
 * shards' list lengths (feed the result to ss_index_set_doc_freq); corpus top-k = all-gather of the shards' hit
 * lists (feed the result to ss_merge_hits).  Host or device buffers; recv holds world * bytes_per_rank, rank order.
 * With host buffers the call returns when the result is there; with device buffers it only enqueues. */
int32_t ss_comm_allreduce_u64(ss_ctx* ctx, uint64_t* buf, uint64_t n);
int32_t ss_comm_allgather(ss_ctx* ctx, const void* send, void* recv, uint64_t bytes_per_rank);

/* ---- link graph: ranking/pagerank.go:17-44 ---------------------------- */
/* Graph as the reference holds it: forw[2] rows parent -> children, flattened
 * to an out-edge CSR over dense ids (node set = parents U children, Q1;
 * frontier pages are nodes with out-degree 0).  The library builds its own
 * HBM layout (in-edge lists, degree-binned row order, non-dangling-first
 * numbering) on the device.  rank/world: this process owns the destination
 * rows of shard `rank` of `world` (doc-range sharding);
 * single GPU = (0, 1).  The input arrays (host or device memory) have been
 * read when the call returns; the last kernels of the layout build may still
 * be running on the context's stream — everything else that touches the
 * graph is ordered behind them there (ss_synchronize to time the build;
 * option "graph.late_free" = 0 makes the call wait itself). */
int32_t ss_graph_create(ss_ctx* ctx, uint64_t n_nodes, uint64_t n_edges,
                        const uint64_t* out_ptr /*[n_nodes+1]*/, const uint32_t* out_dst /*[n_edges]*/,
                        int32_t rank, int32_t world, ss_graph** out);
/* Incremental update of the resident link graph (indexer/indexer.go:302: re-indexing a changed page rewrites
 * its forw[2] row, and setInverted :350-408 may add child pages never seen before).  The child lists of the `changed`
 * parents (distinct node ids) are replaced by new_children[new_ptr[i] .. new_ptr[i+1]); n_nodes_new >= n_nodes admits new
 * nodes (ids n_nodes .. n_nodes_new-1: new children, or new parents when they are in `changed`).  The library keeps the
 * adjacency resident, patches it on the device and rebuilds its layout from it: no upload of the unchanged rows.  The result
 * is the graph ss_graph_create would build from the updated rows (same layout, bit-identical PageRank).  No PageRank state
 * may exist on the graph; on an error the graph is unchanged. */
int32_t ss_graph_apply_delta(ss_graph* g, uint64_t n_nodes_new, uint64_t n_changed, const uint32_t* changed /*[n_changed]*/,
                             const uint64_t* new_ptr /*[n_changed+1]*/, const uint32_t* new_children);
int32_t ss_graph_get_info(const ss_graph* g, ss_graph_info* info);
int32_t ss_graph_destroy(ss_graph* g);

/* ---- PageRank: ranking/pagerank.go:14-145 ------------------------------ */
/* One call = the whole of UpdateTopicSensitivePagerank's compute: all k_topics
 * power iterations (pagerank.go:54-63 runs them sequentially; here they run as
 * one K-wide sweep per iteration), device-resident loop incl. the stop rule
 * (pagerank.go:93,115-119).  n_topic[k] = int(numPages) of category k.
 * eps < 0 never converges (fixed-iteration benchmarking); max_iter = 0 means
 * unbounded.  rank_out [k_topics][n_nodes] topic-major, original ids;
 * iters_out [k_topics].  Requires a (0,1) graph. */
int32_t ss_pagerank_run(ss_graph* g, double damping, double eps, int32_t max_iter,
                        int32_t k_topics, const int32_t* n_topic,
                        double* rank_out, int32_t* iters_out);

/* Step-wise form of the same loop, for multi-GPU hosts (one process per GPU;
 * the host owns the collective, e.g. torch.distributed/RCCL) and for timing.
 *   ss_pr_create   : allocate state for k_topics vectors on this rank's rows
 *   ss_pr_begin    : x0 = 1/n_topic (pagerank.go:103-106) + first contributions
 *   ss_pr_step     : one K-wide sweep over the local rows (computeRankInherited
 *                    :126-145 as a pull SpMV + fused normalise/delta :115-119)
 *   ss_pr_finalize : combine per-rank partial sums, apply the stop rule.
 *                    world==1: folded into begin/step, must not be called.
 *   exchange       : world>1: after begin/step either ss_pr_exchange (RCCL inside the library) or the host
 *                    all-gathers `send` (send_bytes from every rank, rank order) into `recv` itself,
 *                    then ss_pr_finalize.
 * All calls enqueue on the ctx stream and return without waiting, except
 * ss_pr_status / ss_pr_read_*.
 */
int32_t ss_pr_create(ss_graph* g, double damping, double eps, int32_t max_iter,
                     int32_t k_topics, const int32_t* n_topic, ss_pr** out);
int32_t ss_pr_destroy(ss_pr* pr);
/* OPT-IN, beyond what the reference executes: true topic-sensitive PageRank.  The reference advertises
 * Haveliwala's TSPR (README.md:9) but its topics differ only by the start value 1/numPages (pagerank.go:54-63,104): every
 * node teleports with the absolute (1-d) (pagerank.go:117).  With a teleport set per topic — set_ptr[k_topics+1] into
 * set_nodes (DISTINCT original node ids; an empty set keeps that topic on the reference's uniform teleport) — topic k's
 * teleport mass (1-d)*N is spread over its set only: cur[v] = (cur[v] + (v in set_k ? (1-d)*N/|set_k| : 0)) / total,
 * everything else (total, stop rule, start value) as in the reference.  Call after ss_pr_create, before ss_pr_begin;
 * NULL restores the reference behaviour.  Sharded graphs: every rank passes the full sets. */
int32_t ss_pr_set_teleport(ss_pr* pr, const uint64_t* set_ptr /*[k_topics+1]*/, const uint32_t* set_nodes);
int32_t ss_pr_begin(ss_pr* pr);
int32_t ss_pr_step(ss_pr* pr, int32_t n_steps);
int32_t ss_pr_finalize(ss_pr* pr);
int32_t ss_pr_exchange_buffers(ss_pr* pr, void** send_dev, uint64_t* send_bytes,
                               void** recv_dev, uint64_t* recv_bytes);
/* world>1, in-library exchange (needs ss_comm_init with the graph's rank/world): all-gather of the ranks' contribution
 * slices into the full table on the context's stream — the per-iteration collective of the doc-range-sharded sweep.
 * allreduce = 0: all-gather of the non-dangling slices (0.32*N*K*8 bytes received per rank and sweep at the benchmark
 * graph); allreduce = 1: the form the north star names — every rank contributes its slice inside a zeroed full-size
 * table and the tables are summed (all-reduce, ~2x the bytes on the wire, bit-identical result: x + 0 is exact).
 * Call between ss_pr_begin/ss_pr_step and ss_pr_finalize.  Enqueues only. */
int32_t ss_pr_exchange(ss_pr* pr, int32_t allreduce);
/* The whole sharded power iteration of one rank: begin, {sweep, exchange, finalize} until the device-side stop rule
 * (pagerank.go:93) has fired for every topic on every rank (the ranks agree: they finalize the same gathered sums).
 * The K topic vectors run as topic blocks (option "pr.topic_blocks", default 2 above 8 topics) whose exchanges are
 * enqueued on a second stream of the context: block b's collective overlaps block b+1's sweep.
 * Topics are independent, so the results are those of running every block's topics on their own.
 * ids_out [n_rows_local] original node ids of this rank's rows, rank_out [k_topics][n_rows_local], iters_out [k_topics].
 * Every rank writes its own rows of forw[3]; no gather of the result is needed.  k_topics <= 16 per call. */
int32_t ss_pagerank_run_sharded(ss_graph* g, double damping, double eps, int32_t max_iter, int32_t k_topics,
                                const int32_t* n_topic, int32_t allreduce, uint32_t* ids_out, double* rank_out, int32_t* iters_out);
/* The same pipeline for the shards of ONE process: shards[s] = shard s of `world` of the same graph, all on one context
 * (one device); the all-gather is played by device-to-device copies on the context's second stream.  Tests run the
 * topic-blocked, overlapped schedule through this on one GPU (RCCL refuses two ranks on one device), and a host that wants
 * several shards on one device can use it as it is.  rank_out [k_topics][n_nodes] in original ids (host memory). */
int32_t ss_pagerank_run_group(ss_graph* const* shards, int32_t world, double damping, double eps, int32_t max_iter, int32_t k_topics,
                              const int32_t* n_topic, double* rank_out, int32_t* iters_out);
/* Waits for the stream; iters_out[k_topics] = iterations executed per topic,
 * *n_active = topics still iterating, *sweeps = K-wide sweeps executed. */
int32_t ss_pr_status(ss_pr* pr, int32_t* iters_out, int32_t* n_active, int32_t* sweeps,
                     double* last_delta_out /*[k_topics] nullable*/, double* last_total_out /*[k_topics] nullable*/);
/* Local rows of this rank: ids_out[n_rows_local] original node ids,
 * rank_out[k_topics][n_rows_local]. */
int32_t ss_pr_read_local(ss_pr* pr, uint32_t* ids_out, double* rank_out);
/* world==1 only: rank_out[k_topics][n_nodes] in original id order. */
int32_t ss_pr_read(ss_pr* pr, double* rank_out);

/* Diagnostic (bench.py roofline.gather_ceiling_ms; no reference counterpart): average milliseconds of a gather-ONLY
 * pass over this state's in-edge stream and contribution table with the sweep's own load shape — what the access
 * pattern of computeRankInherited (pagerank.go:126-145) costs on this chip with every other part of the sweep
 * removed.  mode 0 = the graph's index stream, 1 = uniformly random rows (no hub reuse), 2 = consecutive rows;
 * + 8 * p selects the cache policy of the gathers (p = 0 default, 1 non-temporal, 2 sc1, 3 / 4 default for the rows
 * below SS_PR_PROBE_HOT and non-temporal / sc1 for the others: tools/pr_probe_pol.py).  States that run the 8- or
 * 16-wide sweep only. */
int32_t ss_pr_probe(ss_pr* pr, int32_t mode, int32_t n_reps, float* ms_out);

/* ---- inverted index + TF-IDF: ranking/term_weighting.go:10-123 --------- */
/* One inverted table (inv[0] title or inv[1] body), term-major CSR over dense
 * term/doc ids; post_tf = listPos[0] (normalised tf, indexer.go:362).  Each
 * term's postings must be strictly ascending by doc id (SS_ERR_UNSORTED). */
int32_t ss_index_create(ss_ctx* ctx, uint64_t n_docs, uint64_t n_terms,
                        const uint64_t* term_ptr /*[n_terms+1]*/, const uint32_t* post_doc,
                        const float* post_tf, ss_index** out);
int32_t ss_index_destroy(ss_index* idx);
/* UpdateTermWeights: idf = float32(log2(total_docs/df)) (:37), w = tf*idf in
 * place (:42), mag[doc] = sqrt(sum float64(float32(w*w))) (:44,:72).
 * total_docs = len(forw[3]) = number of PageRank nodes (:13-17, Q7).
 * w_out [P] / mag_out [n_docs] / idf_out [n_terms] nullable (the shim writes
 * them back to inv[*] / forw[4]).  Not idempotent, like the reference. */
int32_t ss_tfidf_build(ss_index* idx, uint64_t total_docs,
                       float* w_out, double* mag_out, float* idf_out);
/* Doc-range sharding (one shard of the table per GPU): a term's list here is only the
 * slice of its postings that falls into this shard's doc range, but len(docs) in
 * term_weighting.go:37 is the length of the WHOLE list.  df [n_terms] = whole-corpus document
 * frequencies (the host sums the shards' list lengths, one all-reduce); call before
 * ss_tfidf_build.  NULL restores df = local list length.  A doc's postings all live in its own
 * shard, so magnitudes need no exchange. */
int32_t ss_index_set_doc_freq(ss_index* idx, const uint64_t* df /*[n_terms]*/);
/* Load precomputed weights/magnitudes instead (tables already weighted). */
int32_t ss_index_set_weighted(ss_index* idx, const double* mag /*[n_docs]*/);
/* Positional postings for phrase search: listPos[1:] of every posting (parser/parser.go:195-207:
 * float32 positions, -100 for anchor/meta text), pos_ptr[n_postings+1] into pos[]. */
int32_t ss_index_set_positions(ss_index* idx, const uint64_t* pos_ptr, const float* pos);

/* Incremental update of a resident table (indexer/indexer.go:420-641 checkAndUpdate and the re-index
 * that follows it): every posting of del_docs goes (the changed page's old title/body words, :455-531), the single
 * postings (del_term[i], del_doc[i]) go (anchor words of the page's children in inv[0], :533-616; a pair that is not
 * there is ignored like Go's delete on a missing key), the postings (add_term[i], add_doc[i], add_w[i]) arrive (the
 * re-indexed page; a pair must not exist unless this delta deletes it).  The delta is merged into the resident CSR on
 * the device — no re-flatten, no re-upload — and every list is re-validated to be strictly ascending by doc; on any
 * error the table is unchanged.  Weights are taken as given (the reference stores whatever listPos[0] holds).
 * Magnitudes: when the table's squared magnitudes are resident (after ss_tfidf_build or ss_index_refresh_magnitudes), the
 * delta brings the magnitudes of the docs it touches (deleted docs, docs of deleted pairs, docs of new postings) up to date
 * itself: their squares are summed AGAIN from the merged table, in ascending term order — the order in which
 * term_weighting.go:29-46 reaches a doc — so they are what a full pass over the updated table gives, bit for bit, for any
 * weights (idf = log2(N/df) with N the PageRank node count can be 20, 1e-5 or negative: squares dozens of binary orders
 * apart, which a "subtract what left" patch would not survive); a doc left without postings has magnitude exactly 0;
 * ss_index_read_magnitudes reads them back for the forw[4] rows.  After ss_index_set_weighted (magnitudes given from outside) they are NOT touched: call
 * ss_index_refresh_magnitudes.  Positional postings: kept postings keep theirs; ss_index_apply_delta gives the new postings
 * empty position lists, ss_index_apply_delta_pos takes theirs (add_pos_ptr[n_add+1] into add_pos, in the order of the add
 * arrays; parser.go:195-207).  Doc and term ids must exist: grow the table first (ss_index_resize) when a re-indexed page
 * brings new words or new child pages (indexer.go:350-408).  Scorers on this table must be destroyed before and created
 * again after. */
int32_t ss_index_apply_delta(ss_index* idx, uint64_t n_del_docs, const uint32_t* del_docs,
                             uint64_t n_del, const uint32_t* del_term, const uint32_t* del_doc,
                             uint64_t n_add, const uint32_t* add_term, const uint32_t* add_doc, const float* add_w);
int32_t ss_index_apply_delta_pos(ss_index* idx, uint64_t n_del_docs, const uint32_t* del_docs,
                                 uint64_t n_del, const uint32_t* del_term, const uint32_t* del_doc,
                                 uint64_t n_add, const uint32_t* add_term, const uint32_t* add_doc, const float* add_w,
                                 const uint64_t* add_pos_ptr /*[n_add+1] nullable*/, const float* add_pos);
/* Grow the doc and / or term space of a resident table (new docs have no postings and magnitude 0, new terms empty lists). */
int32_t ss_index_resize(ss_index* idx, uint64_t n_docs_new, uint64_t n_terms_new);
/* mag_out[i] = magnitude of docs[i] as it stands (host or device arrays). */
int32_t ss_index_read_magnitudes(ss_index* idx, uint64_t n, const uint32_t* docs, double* mag_out);
/* mag[doc] = sqrt(sum float64(float32(w*w))) over the table's CURRENT weights (term_weighting.go:44,72), without the
 * idf multiplication of ss_tfidf_build.  mag_out [n_docs] nullable. */
int32_t ss_index_refresh_magnitudes(ss_index* idx, double* mag_out);
/* The table as it stands (after updates): sizes, then the arrays (each nullable; host or device). */
int32_t ss_index_get_info(const ss_index* idx, uint64_t* n_docs, uint64_t* n_terms, uint64_t* n_post);
int32_t ss_index_read(ss_index* idx, uint64_t* term_ptr_out /*[n_terms+1]*/, uint32_t* post_doc_out, float* post_w_out);
/* the positional postings as they stand (each nullable): pos_ptr_out [n_post+1], pos_out [pos_ptr[n_post]] */
int32_t ss_index_read_positions(ss_index* idx, uint64_t* pos_ptr_out, float* pos_out);

/* ---- scoring: retrieval/main_retrieve.go:50-103, get_metadata.go:31-69 -- */
int32_t ss_scorer_create(ss_ctx* ctx, ss_index* title, ss_index* body, ss_scorer** out);
int32_t ss_scorer_destroy(ss_scorer* s);
/* forw[3] ranks for the PageRank blend (get_metadata.go:31-42):
 * rank [k_topics][n_docs] topic-major (ss_pagerank_run's layout). k_topics=0 clears. */
int32_t ss_scorer_set_prior(ss_scorer* s, int32_t k_topics, const double* rank);
/* Batch of OR queries.  q_ptr[n_q+1] into q_terms (term ids in query order,
 * duplicates kept, SS_UNKNOWN_TERM for unknown words); query_len[n_q]
 * (len(queryTokenised)+len(phraseTokenised), main_retrieve.go:90; NULL = term
 * count); topic_probs [n_q][k_topics] or NULL (nil map => sqd = 0,
 * main_retrieve.go:88).  hits_out [n_q][k], n_hits_out [n_q] (host or device).
 * Order: FinalRank descending (util.go:48-54), ties ascending doc id, NaN last;
 * reference k = 50 (main_retrieve.go:99-100).
 * The QUERY arrays (q_ptr, q_terms, query_len, topic_probs, and p_ptr / p_terms below) are consumed on the HOST: the
 * slice plan of a batch (which lists, which doc ranges, which kernel) is made on the CPU from the host copy of the
 * tables' term_ptr, as the query words themselves arrive from a host-side tokenizer (main_retrieve.go:17-36).  Pass host
 * arrays; device pointers are accepted and cost one blocking copy back each (~16 KB for 1024 queries).
 * Results in HOST memory: the call returns when they are there.  Results in DEVICE memory (both
 * pointers): the kernels write them directly and the call returns once the work is enqueued on the
 * ctx stream — later work on that stream sees them; ss_synchronize waits for them. */
int32_t ss_score_topk(ss_scorer* s, int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms,
                      const int32_t* query_len, const double* topic_probs, int32_t k,
                      ss_hit* hits_out, int32_t* n_hits_out);

/* The same batch with results in HOST memory and several batches in flight (a server that has the next batch of requests
 * ready while this one runs): ss_score_topk_submit enqueues the batch into device buffers that belong to the ticket, records an event
 * behind its kernels and returns the ticket (never 0) without waiting; ss_score_topk_collect waits for THAT batch's event, copies the
 * rows to the caller (a plain device-to-host copy at collect time; option "score.collect_pinned" = 1 stages it through pinned
 * memory) and writes hits_out [n_q][k],
 * n_hits_out [n_q] (host memory, the n_q and k of the submit).  Up to SS_SCORE_INFLIGHT tickets may be outstanding
 * (SS_ERR_STATE beyond); collect them in any order.  The host's plan for batch i+1 and its copy-out of batch i-1 run under the
 * kernels of batch i: 1024-query batches go host-to-host at the device's batch rate instead of one batch alone plus the copies. */
#define SS_SCORE_INFLIGHT 3
int32_t ss_score_topk_submit(ss_scorer* s, int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms,
                             const uint32_t* p_ptr /* NULL: no quoted phrases (ss_score_topk); else as ss_score_topk_phrase */,
                             const uint32_t* p_terms, const int32_t* query_len, const double* topic_probs, int32_t k,
                             uint64_t* ticket_out);
int32_t ss_score_topk_collect(ss_scorer* s, uint64_t ticket, ss_hit* hits_out, int32_t* n_hits_out);

/* ss_score_topk plus the quoted-phrase part of retrieval.Retrieve (retrieval/phrase.go:11-170,
 * util.go:162-203, merged at main_retrieve.go:73-78).  p_ptr[n_q+1] into p_terms: the tokens of ALL quoted
 * phrases of a query, concatenated into one phrase as the reference does (main_retrieve.go:26); a doc
 * matches if it holds every phrase term (body or title) and, per field, the term positions shifted by
 * the term's index intersect; its float32 weight sums are added to TitleRank/BodyRank.  query_len NULL =
 * len(query tokens)+len(phrase tokens) (main_retrieve.go:90).  At most SS_MAX_PHRASE_TERMS per phrase.
 * Needs ss_index_set_positions on both tables. */
#define SS_MAX_PHRASE_TERMS 16
int32_t ss_score_topk_phrase(ss_scorer* s, int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms,
                             const uint32_t* p_ptr, const uint32_t* p_terms, const int32_t* query_len,
                             const double* topic_probs, int32_t k, ss_hit* hits_out, int32_t* n_hits_out);

/* Doc-range-sharded scoring: every shard scores the same query batch against its own doc range
 * (ss_score_topk, local doc ids) and the host gathers the lists.  ss_merge_hits returns the k best
 * of the union per query in the order of appendSort (util.go:48-54) and the cut of
 * main_retrieve.go:99-103 — identical to scoring the unsharded index.
 * parts [n_parts][n_q][k], n_hits [n_parts][n_q] (the all-gathered outputs of ss_score_topk),
 * doc_base [n_parts] first corpus doc id of each shard (NULL = ids already global);
 * hits_out [n_q][k], n_hits_out [n_q].  All pointers host or device. */
#define SS_MAX_SHARDS 64
int32_t ss_merge_hits(ss_ctx* ctx, int32_t n_q, int32_t n_parts, int32_t k, const ss_hit* parts,
                      const int32_t* n_hits, const uint32_t* doc_base, ss_hit* hits_out, int32_t* n_hits_out);

/* Timing hook for bench.py: milliseconds between the start and end HIP events
 * recorded on the ctx stream around the LAST compute call of the given kind
 * (0 = ss_pr_step batch, 1 = ss_score_topk, 2 = ss_tfidf_build). */
int32_t ss_last_kernel_ms(ss_ctx* ctx, int32_t kind, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif
