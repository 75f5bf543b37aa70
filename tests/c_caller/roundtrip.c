/*
 * roundtrip.c — a plain-C99 caller of the C ABI (include/spaghetti_rank.h), shaped like the code cgo generates for
 * go/spaghetti: host buffers in, status codes out, ss_last_error on failure.  tests/test_gpu_c_caller.py writes the
 * inputs, runs this program on the GPU box and compares its outputs with the oracle.
 *
 *   roundtrip <in.bin> <out.bin>
 * in.bin  : u64 header[10] = {n_nodes, n_edges, k_topics, n_docs, n_terms, Pt, Pb, n_q, n_tok, k}
 *           f64 d, eps | i32 n_topic[k_topics]
 *           u64 out_ptr[n_nodes+1] | u32 out_dst[n_edges]
 *           u64 t_ptr[n_terms+1] | u32 t_doc[Pt] | f32 t_tf[Pt] | u64 b_ptr[n_terms+1] | u32 b_doc[Pb] | f32 b_tf[Pb]
 *           u32 q_ptr[n_q+1] | u32 q_terms[n_tok]
 * out.bin : f64 rank[k_topics][n_nodes] | i32 iters[k_topics] | f32 t_w[Pt] | f64 t_mag[n_docs] | f32 b_w[Pb] |
 *           f64 b_mag[n_docs] | ss_hit hits[n_q][k] | i32 n_hits[n_q]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spaghetti_rank.h"

static ss_ctx* g_ctx = 0;

static void die(const char* what, int rc)
{
    fprintf(stderr, "roundtrip: %s failed: rc=%d (%s)\n", what, rc, ss_last_error(g_ctx));
    exit(2);
}
#define CALL(expr) do { int rc_ = (expr); if (rc_ != SS_OK) die(#expr, rc_); } while (0)

static void* rd(FILE* f, size_t n, size_t sz)
{
    void* p = malloc(n * sz + 1);
    if (!p || fread(p, sz, n, f) != n) { fprintf(stderr, "roundtrip: short read\n"); exit(3); }
    return p;
}
static void wr(FILE* f, const void* p, size_t n, size_t sz)
{
    if (fwrite(p, sz, n, f) != n) { fprintf(stderr, "roundtrip: short write\n"); exit(3); }
}

int main(int argc, char** argv)
{
    if (argc != 3) { fprintf(stderr, "usage: roundtrip in.bin out.bin\n"); return 1; }
    FILE* in = fopen(argv[1], "rb");
    FILE* out = fopen(argv[2], "wb");
    if (!in || !out) { perror("open"); return 1; }
    uint64_t* h = (uint64_t*)rd(in, 10, 8);
    const uint64_t n_nodes = h[0], n_edges = h[1], k_topics = h[2], n_docs = h[3], n_terms = h[4], Pt = h[5], Pb = h[6],
                   n_q = h[7], n_tok = h[8], k = h[9];
    double* de = (double*)rd(in, 2, 8);
    int32_t* n_topic = (int32_t*)rd(in, k_topics, 4);
    uint64_t* out_ptr = (uint64_t*)rd(in, n_nodes + 1, 8);
    uint32_t* out_dst = (uint32_t*)rd(in, n_edges, 4);
    uint64_t* t_ptr = (uint64_t*)rd(in, n_terms + 1, 8);
    uint32_t* t_doc = (uint32_t*)rd(in, Pt, 4);
    float* t_tf = (float*)rd(in, Pt, 4);
    uint64_t* b_ptr = (uint64_t*)rd(in, n_terms + 1, 8);
    uint32_t* b_doc = (uint32_t*)rd(in, Pb, 4);
    float* b_tf = (float*)rd(in, Pb, 4);
    uint32_t* q_ptr = (uint32_t*)rd(in, n_q + 1, 4);
    uint32_t* q_terms = (uint32_t*)rd(in, n_tok, 4);
    fclose(in);

    if (ss_abi_version() != SS_ABI_VERSION) { fprintf(stderr, "roundtrip: ABI mismatch\n"); return 4; }
    CALL(ss_init(0, &g_ctx));

    /* ranking.UpdateTopicSensitivePagerank (pagerank.go:14-83): graph in, K rank vectors out */
    ss_graph* g = 0;
    CALL(ss_graph_create(g_ctx, n_nodes, n_edges, out_ptr, out_dst, 0, 1, &g));
    ss_graph_info gi;
    CALL(ss_graph_get_info(g, &gi));
    if (gi.n_nodes != n_nodes || gi.n_edges != n_edges) { fprintf(stderr, "roundtrip: graph info mismatch\n"); return 5; }
    double* rank = (double*)malloc(sizeof(double) * k_topics * n_nodes);
    int32_t* iters = (int32_t*)malloc(sizeof(int32_t) * k_topics);
    CALL(ss_pagerank_run(g, de[0], de[1], 0, (int32_t)k_topics, n_topic, rank, iters));
    CALL(ss_graph_destroy(g));
    wr(out, rank, k_topics * n_nodes, 8);
    wr(out, iters, k_topics, 4);

    /* ranking.UpdateTermWeights, title then body (start_crawl.go:176-177; total_docs = len(forw[3]) = n_nodes) */
    ss_index *ti = 0, *bi = 0;
    CALL(ss_index_create(g_ctx, n_docs, n_terms, t_ptr, t_doc, t_tf, &ti));
    CALL(ss_index_create(g_ctx, n_docs, n_terms, b_ptr, b_doc, b_tf, &bi));
    float* t_w = (float*)malloc(sizeof(float) * (Pt ? Pt : 1));
    float* b_w = (float*)malloc(sizeof(float) * (Pb ? Pb : 1));
    double* t_mag = (double*)malloc(sizeof(double) * n_docs);
    double* b_mag = (double*)malloc(sizeof(double) * n_docs);
    CALL(ss_tfidf_build(ti, n_nodes, t_w, t_mag, 0));
    CALL(ss_tfidf_build(bi, n_nodes, b_w, b_mag, 0));
    wr(out, t_w, Pt, 4);
    wr(out, t_mag, n_docs, 8);
    wr(out, b_w, Pb, 4);
    wr(out, b_mag, n_docs, 8);

    /* retrieval.Retrieve (main_retrieve.go:15-104) for a batch: OR queries, top-k */
    ss_scorer* sc = 0;
    CALL(ss_scorer_create(g_ctx, ti, bi, &sc));
    ss_hit* hits = (ss_hit*)calloc(n_q * k + 1, sizeof(ss_hit));
    int32_t* n_hits = (int32_t*)calloc(n_q + 1, sizeof(int32_t));
    CALL(ss_score_topk(sc, (int32_t)n_q, q_ptr, q_terms, 0, 0, (int32_t)k, hits, n_hits));
    /* an error must come back as a status with text, never crash: k = 0 */
    if (ss_score_topk(sc, (int32_t)n_q, q_ptr, q_terms, 0, 0, 0, hits, n_hits) != SS_ERR_INVALID || !strlen(ss_last_error(g_ctx))) {
        fprintf(stderr, "roundtrip: k=0 was not rejected\n");
        return 6;
    }
    wr(out, hits, n_q * k, sizeof(ss_hit));
    wr(out, n_hits, n_q, 4);
    CALL(ss_scorer_destroy(sc));
    CALL(ss_index_destroy(ti));
    CALL(ss_index_destroy(bi));
    CALL(ss_shutdown(g_ctx));
    fclose(out);
    printf("roundtrip: ok\n");
    return 0;
}
