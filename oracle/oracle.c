/*
 * oracle.c — CPU restatement of SpaghettiSearch's ranking hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY — see oracle.h.  PARITY UNPINNED (no reference
 * fixtures exist, reference not buildable here); pinned by hand-derived KATs
 * and an independent numpy restatement instead.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: the reference runs on
 * amd64, where the Go compiler never fuses a*b+c, so neither may we).
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* Go's math.Log / math.Log2 (go1.12 src/math/log.go, log10.go), restated.    */
/* term_weighting.go:37 calls math.Log2.                                      */
/* ------------------------------------------------------------------------- */
double orc_go_log(double x)
{
    static const double Ln2Hi = 6.93147180369123816490e-01; /* 3fe62e42 fee00000 */
    static const double Ln2Lo = 1.90821492927058770002e-10; /* 3dea39ef 35793c76 */
    static const double L1 = 6.666666666666735130e-01;
    static const double L2 = 3.999999999940941908e-01;
    static const double L3 = 2.857142874366239149e-01;
    static const double L4 = 2.222219843214978396e-01;
    static const double L5 = 1.818357216161805012e-01;
    static const double L6 = 1.531383769920937332e-01;
    static const double L7 = 1.479819860511658591e-01;
    static const double Sqrt2Half = 0.70710678118654752440084436210484903928483593768847 ;

    if (isnan(x) || (isinf(x) && x > 0)) return x;
    if (x < 0) return NAN;
    if (x == 0) return -INFINITY;

    int ki;
    double f1 = frexp(x, &ki);
    if (f1 < Sqrt2Half) {
        f1 *= 2;
        ki--;
    }
    double f = f1 - 1;
    double k = (double)ki;

    double s = f / (2 + f);
    double s2 = s * s;
    double s4 = s2 * s2;
    double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    double R = t1 + t2;
    double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

double orc_go_log2(double x)
{
    /* 1/Ln2, constant-folded exactly by the Go compiler then rounded once. */
    static const double InvLn2 = 1.44269504088896340735992468100189214;
    int e;
    double frac = frexp(x, &e);
    /* exact powers of two give an exact answer (log10.go: log2) */
    if (frac == 0.5) return (double)(e - 1);
    return orc_go_log(frac) * InvLn2 + (double)e;
}

/* Oracle hardening (no reference line: a check ON the restatement above).
 * term_weighting.go:37 narrows math.Log2(total/df) to float32.  go1.12/amd64 dispatches
 * math.Log to log_amd64.s, a hand transcription of the same FreeBSD algorithm as log.go;
 * whether or not the two agree to the last bit, the float32 idf is the same for every
 * input whose float64 log2 is not within a couple of ulps of a float32 rounding boundary.
 * For df in [df_lo, df_hi] and x = total_docs/df (rounded to float64 as Go does) this counts
 *   out[0]  float32(orc_go_log2(x)) != float32(log2 correctly rounded, via 80-bit log2l)
 *   out[1]  orc_go_log2(x) within `margin_ulps` x (ulp(result) + ulp(Log(frac))/Ln2) of a float32 rounding boundary
 *           ("sensitive": a last-ulp difference between two libm paths could flip the idf)
 *   out[2]  the long double value itself too close to a boundary to decide (2^-58 relative)
 *   out[3]  max |orc_go_log2 - log2l| in float64 ulps, rounded up
 * first_bad_df (nullable): the first df counted in out[0] or out[1], 0 if none. */
int orc_log2_sensitivity(double total_docs, uint64_t df_lo, uint64_t df_hi, double margin_ulps,
                         uint64_t out[4], uint64_t* first_bad_df)
{
    out[0] = out[1] = out[2] = out[3] = 0;
    if (first_bad_df) *first_bad_df = 0;
    for (uint64_t df = df_lo; df <= df_hi && df != 0; df++) {
        const double x = total_docs / (double)df;                 /* term_weighting.go:37 argument */
        const double y = orc_go_log2(x);
        const long double yl = log2l((long double)x);
        const float f = (float)y;
        const float fl = (float)yl;
        int bad = 0;
        if (!(f == fl) && !(isnan(f) && isnan(fl))) { out[0]++; bad = 1; }
        if (isfinite(y) && y != 0.0) {
            const double ulp = fabs(nextafter(y, INFINITY) - y);
            /* Log2 = Log(frac)*(1/Ln2) + exp cancels for x near a power of two: a last-ulp difference in
             * Log(frac) reaches y as an ABSOLUTE error of ulp(Log(frac))/Ln2, however small y is */
            int e2;
            const double lg = fabs(orc_go_log(frexp(x, &e2)));
            const double ulp_in = (nextafter(lg, INFINITY) - lg) * 1.4426950408889634;
            const double up = ((double)f + (double)nextafterf(f, INFINITY)) * 0.5;
            const double dn = ((double)f + (double)nextafterf(f, -INFINITY)) * 0.5;
            const double dist = fmin(fabs(y - up), fabs(y - dn));
            if (dist <= margin_ulps * (ulp + ulp_in)) { out[1]++; bad = 1; }
            const long double distl = fminl(fabsl(yl - (long double)up), fabsl(yl - (long double)dn));
            if (distl <= fabsl(yl) * 0x1p-58L) out[2]++;
            const long double err = fabsl((long double)y - yl) / (long double)ulp;
            const uint64_t e = (uint64_t)ceill(err);
            if (e > out[3]) out[3] = e;
        }
        if (bad && first_bad_df && *first_bad_df == 0) *first_bad_df = df;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* PageRank — ranking/pagerank.go:85-145                                      */
/* ------------------------------------------------------------------------- */
int orc_pagerank_topic(uint64_t n_nodes, const uint64_t* out_ptr, const uint32_t* out_dst,
                       double d, double eps, int32_t max_iter, int32_t n_init,
                       double* rank, int32_t* iters, double* last_change_out, double* last_total_out)
{
    const uint64_t N = n_nodes;
    double* a = (double*)malloc(sizeof(double) * (N ? N : 1));
    double* b = (double*)malloc(sizeof(double) * (N ? N : 1));
    if (!a || !b) { free(a); free(b); return -1; }
    double* cur = a;  /* currentRank */
    double* last = b; /* lastRank */

    const double teleport = 1.0 - d; /* pagerank.go:90 */
    double last_change = DBL_MAX;    /* pagerank.go:93: math.MaxFloat64 */
    double total = 0.0;
    int32_t iteration = 1;
    for (; last_change > eps; iteration++) {
        /* pagerank.go:94 */
        double* t = cur; cur = last; last = t;

        if (iteration > 1) {
            for (uint64_t v = 0; v < N; v++) cur[v] = 0.0;          /* :98-100 */
        } else {
            const double u = 1.0 / (double)n_init;                   /* :104-105 */
            for (uint64_t v = 0; v < N; v++) { cur[v] = u; last[v] = u; }
        }

        /* computeRankInherited, pagerank.go:126-145 */
        total = 0.0;
        for (uint64_t p = 0; p < N; p++) {
            const uint64_t beg = out_ptr[p], end = out_ptr[p + 1];
            if (end == beg) continue;                                /* :132-134 dangling dropped */
            const double w = d * last[p] / (double)(end - beg);      /* :136 */
            total += w;                                              /* :137 */
            for (uint64_t e = beg; e < end; e++) cur[out_dst[e]] += w; /* :140-142 */
        }
        total += teleport * (double)N;                               /* :112 */

        last_change = 0.0;
        for (uint64_t v = 0; v < N; v++) {                           /* :116-119 */
            cur[v] = (cur[v] + teleport) / total;
            last_change += fabs(cur[v] - last[v]);
        }
        if (max_iter > 0 && iteration >= max_iter) { iteration++; break; }
    }
    memcpy(rank, cur, sizeof(double) * N);
    if (iters) *iters = iteration - 1;
    if (last_change_out) *last_change_out = last_change;
    if (last_total_out) *last_total_out = total;
    free(a); free(b);
    return 0;
}

/* OPT-IN extension, no reference counterpart: the loop of pagerank.go:85-124 with a topic's teleport
 * set.  member[v] != 0: node v is in the set (n_members of them).  The reference adds the absolute (1-d) to EVERY node
 * (:117); here the same total mass (1-d)*N goes to the set's nodes only: (1-d)*N/n_members each.  `total` (:111-112),
 * the start value 1/n_init (:104) and the stop rule (:93) are the reference's.  member == NULL: orc_pagerank_topic. */
int orc_pagerank_topic_ts(uint64_t n_nodes, const uint64_t* out_ptr, const uint32_t* out_dst,
                          double d, double eps, int32_t max_iter, int32_t n_init,
                          const uint8_t* member, uint64_t n_members,
                          double* rank, int32_t* iters, double* last_change_out, double* last_total_out)
{
    if (!member || n_members == 0)
        return orc_pagerank_topic(n_nodes, out_ptr, out_dst, d, eps, max_iter, n_init, rank, iters, last_change_out, last_total_out);
    const uint64_t N = n_nodes;
    double* a = (double*)malloc(sizeof(double) * (N ? N : 1));
    double* b = (double*)malloc(sizeof(double) * (N ? N : 1));
    if (!a || !b) { free(a); free(b); return -1; }
    double* cur = a;
    double* last = b;
    const double teleport = 1.0 - d;                                 /* pagerank.go:90 */
    const double tin = teleport * (double)N / (double)n_members;     /* a member's teleport */
    double last_change = DBL_MAX;
    double total = 0.0;
    int32_t iteration = 1;
    for (; last_change > eps; iteration++) {
        double* t = cur; cur = last; last = t;
        if (iteration > 1) {
            for (uint64_t v = 0; v < N; v++) cur[v] = 0.0;
        } else {
            const double u = 1.0 / (double)n_init;
            for (uint64_t v = 0; v < N; v++) { cur[v] = u; last[v] = u; }
        }
        total = 0.0;
        for (uint64_t p = 0; p < N; p++) {
            const uint64_t beg = out_ptr[p], end = out_ptr[p + 1];
            if (end == beg) continue;
            const double w = d * last[p] / (double)(end - beg);
            total += w;
            for (uint64_t e = beg; e < end; e++) cur[out_dst[e]] += w;
        }
        total += teleport * (double)N;
        last_change = 0.0;
        for (uint64_t v = 0; v < N; v++) {
            cur[v] = (cur[v] + (member[v] ? tin : 0.0)) / total;
            last_change += fabs(cur[v] - last[v]);
        }
        if (max_iter > 0 && iteration >= max_iter) { iteration++; break; }
    }
    memcpy(rank, cur, sizeof(double) * N);
    if (iters) *iters = iteration - 1;
    if (last_change_out) *last_change_out = last_change;
    if (last_total_out) *last_total_out = total;
    free(a); free(b);
    return 0;
}

/* retrieval/main_retrieve.go:106-159 computeTopicProbs (disabled in the reference: :43 is commented out and :87 passes a nil
 * map).  K categories of forw[5] with their wordCount (:144); query token i has the category->frequency map of inv[2]
 * (tok_ptr[n_tok+1] into tok_cat / tok_freq; tok_missing[i] != 0: the word is not in inv[2] — the reference panics there,
 * :120-121: returns -2).  mode 0 = AS WRITTEN: `var probs float64` starts at 0 and is only ever multiplied (:142-145), so
 * every probability is 0; mode 1 = the evident intent: the product starts at 1 (multinomial naive Bayes, uniform prior
 * 1/K, :148).  Topics no query word maps to get 0 (:149-151). */
int orc_topic_probs(int32_t k_topics, const double* word_count, int32_t n_tok, const uint32_t* tok_ptr,
                    const uint32_t* tok_cat, const double* tok_freq, const uint8_t* tok_missing, int32_t mode, double* probs_out)
{
    for (int32_t i = 0; i < n_tok; i++)
        if (tok_missing && tok_missing[i]) return -2;                /* :120-121 panic(err) */
    for (int32_t k = 0; k < k_topics; k++) {
        double probs = mode ? 1.0 : 0.0;                             /* :142 `var probs float64` */
        int any = 0;
        for (int32_t i = 0; i < n_tok; i++)                          /* topicTF[topic] in token order (:118-135) */
            for (uint32_t e = tok_ptr[i]; e < tok_ptr[i + 1]; e++)
                if ((int32_t)tok_cat[e] == k) { probs *= (tok_freq[e] / word_count[k]); any = 1; }   /* :143-145 */
        probs_out[k] = any ? probs / (double)k_topics : 0.0;         /* :148 / :150 */
    }
    return 0;
}

int orc_pagerank(uint64_t n_nodes, const uint64_t* out_ptr, const uint32_t* out_dst,
                 double d, double eps, int32_t max_iter, int32_t k_topics,
                 const int32_t* n_topic, double* rank_out, int32_t* iters_out)
{
    /* pagerank.go:54-63: sequential loop over categories */
    for (int32_t k = 0; k < k_topics; k++) {
        int rc = orc_pagerank_topic(n_nodes, out_ptr, out_dst, d, eps, max_iter, n_topic[k],
                                    rank_out + (uint64_t)k * n_nodes,
                                    iters_out ? iters_out + k : NULL, NULL, NULL);
        if (rc) return rc;
    }
    return 0;
}

/* --- reference-shaped variant: string-keyed hash maps (CPU baseline only) --- */

typedef struct { char key[32]; double val; uint8_t used; } hm_slot;
typedef struct { hm_slot* s; uint64_t mask; } hmap;

static uint64_t hm_hash(const char* k)
{
    /* FNV-1a over the 32 key bytes, like hashing a Go string key */
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 32; i++) { h ^= (uint8_t)k[i]; h *= 1099511628211ull; }
    return h ^ (h >> 29);
}
static int hm_init(hmap* m, uint64_t n)
{
    uint64_t cap = 16;
    while (cap < n * 2) cap <<= 1;
    m->s = (hm_slot*)calloc(cap, sizeof(hm_slot));
    m->mask = cap - 1;
    return m->s ? 0 : -1;
}
static double* hm_at(hmap* m, const char* k)
{
    uint64_t i = hm_hash(k) & m->mask;
    for (;;) {
        hm_slot* s = &m->s[i];
        if (!s->used) { memcpy(s->key, k, 32); s->used = 1; s->val = 0.0; return &s->val; }
        if (memcmp(s->key, k, 32) == 0) return &s->val;
        i = (i + 1) & m->mask;
    }
}
static void id_to_key(uint32_t id, char* out)
{
    /* stand-in for md5-hex(url): 128 mixed bits printed as 32 hex chars */
    static const char hex[] = "0123456789abcdef";
    uint64_t x = (uint64_t)id * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    uint64_t y = x;
    for (int w = 0; w < 2; w++) {
        y ^= y >> 30; y *= 0xBF58476D1CE4E5B9ull; y ^= y >> 27; y *= 0x94D049BB133111EBull; y ^= y >> 31;
        for (int i = 0; i < 16; i++) out[w * 16 + i] = hex[(y >> (4 * i)) & 15];
        y += x;
    }
}

int orc_pagerank_topic_hashed(uint64_t n_nodes, const uint64_t* out_ptr, const uint32_t* out_dst,
                              double d, double eps, int32_t max_iter, int32_t n_init,
                              double* rank, int32_t* iters)
{
    const uint64_t N = n_nodes;
    char* keys = (char*)malloc(32 * (N ? N : 1));
    hmap A, B;
    if (!keys || hm_init(&A, N) || hm_init(&B, N)) return -1;
    for (uint64_t v = 0; v < N; v++) id_to_key((uint32_t)v, keys + 32 * v);
    hmap* cur = &A; hmap* last = &B;
    const double teleport = 1.0 - d;
    double last_change = DBL_MAX;
    int32_t iteration = 1;
    for (; last_change > eps; iteration++) {
        hmap* t = cur; cur = last; last = t;
        if (iteration > 1) {
            for (uint64_t v = 0; v < N; v++) *hm_at(cur, keys + 32 * v) = 0.0;
        } else {
            const double u = 1.0 / (double)n_init;
            for (uint64_t v = 0; v < N; v++) { *hm_at(cur, keys + 32 * v) = u; *hm_at(last, keys + 32 * v) = u; }
        }
        double total = 0.0;
        for (uint64_t p = 0; p < N; p++) {
            const uint64_t beg = out_ptr[p], end = out_ptr[p + 1];
            if (end == beg) continue;
            const double w = d * *hm_at(last, keys + 32 * p) / (double)(end - beg);
            total += w;
            for (uint64_t e = beg; e < end; e++) *hm_at(cur, keys + 32 * (uint64_t)out_dst[e]) += w;
        }
        total += teleport * (double)N;
        last_change = 0.0;
        for (uint64_t v = 0; v < N; v++) {
            double* c = hm_at(cur, keys + 32 * v);
            *c = (*c + teleport) / total;
            last_change += fabs(*c - *hm_at(last, keys + 32 * v));
        }
        if (max_iter > 0 && iteration >= max_iter) { iteration++; break; }
    }
    for (uint64_t v = 0; v < N; v++) rank[v] = *hm_at(cur, keys + 32 * v);
    if (iters) *iters = iteration - 1;
    free(A.s); free(B.s); free(keys);
    return 0;
}

/* --- "strong CPU" variant: flat in-edge lists, pull form, OpenMP over all cores. --- */
/* Same arithmetic per node; the float64 sums run in a different order (in-edge order per node, per-   */
/* thread partial sums), so it agrees with orc_pagerank_topic to ~1e-15, not bit for bit.              */
/* CPU baseline only.                                                                                  */
int orc_pagerank_topic_omp(uint64_t n_nodes, const uint64_t* out_ptr, const uint32_t* out_dst,
                           double d, double eps, int32_t max_iter, int32_t n_init,
                           double* rank, int32_t* iters, int32_t* threads_used)
{
    const uint64_t N = n_nodes, E = out_ptr[N];
    uint64_t* in_ptr = (uint64_t*)calloc(N + 2, sizeof(uint64_t));
    uint32_t* in_src = (uint32_t*)malloc(sizeof(uint32_t) * (E ? E : 1));
    double* cur = (double*)malloc(sizeof(double) * (N ? N : 1));
    double* last = (double*)malloc(sizeof(double) * (N ? N : 1));
    double* contrib = (double*)malloc(sizeof(double) * (N ? N : 1));
    if (!in_ptr || !in_src || !cur || !last || !contrib) return -1;
    /* transpose once (counting sort by destination) */
    for (uint64_t e = 0; e < E; e++) in_ptr[out_dst[e] + 2]++;
    for (uint64_t v = 0; v < N; v++) in_ptr[v + 2] += in_ptr[v + 1];
    for (uint64_t p = 0; p < N; p++)
        for (uint64_t e = out_ptr[p]; e < out_ptr[p + 1]; e++) in_src[in_ptr[out_dst[e] + 1]++] = (uint32_t)p;
    const double teleport = 1.0 - d;
    const double u = 1.0 / (double)n_init;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    #pragma omp parallel for schedule(static)
    for (uint64_t v = 0; v < N; v++) last[v] = u;
    double last_change = DBL_MAX;
    int32_t iteration = 1;
    for (; last_change > eps; iteration++) {
        double total = 0.0;
        #pragma omp parallel for schedule(static) reduction(+ : total)
        for (uint64_t p = 0; p < N; p++) {
            const uint64_t deg = out_ptr[p + 1] - out_ptr[p];
            double w = 0.0;
            if (deg) { w = d * last[p] / (double)deg; total += w; }
            contrib[p] = w;
        }
        total += teleport * (double)N;
        const double first = iteration == 1 ? u : 0.0;
        double change = 0.0;
        #pragma omp parallel for schedule(dynamic, 4096) reduction(+ : change)
        for (uint64_t v = 0; v < N; v++) {
            double y = first;
            for (uint64_t e = in_ptr[v]; e < in_ptr[v + 1]; e++) y += contrib[in_src[e]];
            const double x = (y + teleport) / total;
            change += fabs(x - last[v]);
            cur[v] = x;
        }
        last_change = change;
        double* t = cur; cur = last; last = t;
        if (max_iter > 0 && iteration >= max_iter) { iteration++; break; }
    }
    memcpy(rank, last, sizeof(double) * N);
    if (iters) *iters = iteration - 1;
    if (threads_used) *threads_used = nthreads;
    free(in_ptr); free(in_src); free(cur); free(last); free(contrib);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* TF-IDF build — ranking/term_weighting.go:10-57                             */
/* ------------------------------------------------------------------------- */
int orc_tfidf(uint64_t n_terms, const uint64_t* term_ptr, const uint32_t* post_doc,
              float* post_w, double total_docs, uint64_t n_docs, double* mag2, float* idf_out)
{
    for (uint64_t t = 0; t < n_terms; t++) {                        /* :29 */
        const uint64_t beg = term_ptr[t], end = term_ptr[t + 1];
        const double df = (double)(end - beg);                       /* len(val) */
        const float idf = (float)orc_go_log2(total_docs / df);       /* :37 */
        if (idf_out) idf_out[t] = idf;
        for (uint64_t i = beg; i < end; i++) {                       /* :40 */
            const float w = post_w[i] * idf;                         /* :42 float32 multiply */
            post_w[i] = w;
            const float sq = w * w;                                  /* :44 float32 product ... */
            if (post_doc[i] >= n_docs) return -2;
            mag2[post_doc[i]] += (double)sq;                         /* ... widened, float64 sum */
        }
    }
    return 0;
}

void orc_sqrt_inplace(uint64_t n, double* v)
{
    for (uint64_t i = 0; i < n; i++) v[i] = sqrt(v[i]);              /* term_weighting.go:72,97,105 */
}

/* ------------------------------------------------------------------------- */
/* Scoring — retrieval/main_retrieve.go:50-103, get_metadata.go:31-69,         */
/*           util.go:48-54                                                     */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint64_t n_docs;
    double* acc_t;      /* TitleRank accumulators (genAggrDocsPipeline :176-178) */
    double* acc_b;      /* BodyRank  accumulators (:180-182) */
    uint8_t* touched;
    uint32_t* cand;     /* touched doc list */
    uint64_t n_cand, cap_cand;
    orc_hit* rows;
    uint64_t cap_rows;
} score_ws;

static int ws_init(score_ws* w, uint64_t n_docs)
{
    memset(w, 0, sizeof(*w));
    w->n_docs = n_docs;
    w->acc_t = (double*)calloc(n_docs ? n_docs : 1, sizeof(double));
    w->acc_b = (double*)calloc(n_docs ? n_docs : 1, sizeof(double));
    w->touched = (uint8_t*)calloc(n_docs ? n_docs : 1, 1);
    w->cap_cand = 1024;
    w->cand = (uint32_t*)malloc(sizeof(uint32_t) * w->cap_cand);
    w->cap_rows = 1024;
    w->rows = (orc_hit*)malloc(sizeof(orc_hit) * w->cap_rows);
    return (w->acc_t && w->acc_b && w->touched && w->cand && w->rows) ? 0 : -1;
}
static void ws_free(score_ws* w)
{
    free(w->acc_t); free(w->acc_b); free(w->touched); free(w->cand); free(w->rows);
}
static int ws_touch(score_ws* w, uint32_t doc)
{
    if (!w->touched[doc]) {
        w->touched[doc] = 1;
        if (w->n_cand == w->cap_cand) {
            w->cap_cand *= 2;
            uint32_t* nc = (uint32_t*)realloc(w->cand, sizeof(uint32_t) * w->cap_cand);
            if (!nc) return -1;
            w->cand = nc;
        }
        w->cand[w->n_cand++] = doc;
    }
    return 0;
}

/* util.go:49: descending FinalRank; equal elements keep insertion order in the
 * reference (arrival order, unspecified) — fixed here as ascending doc id.
 * NaN finals sort last. */
static int hit_cmp(const void* pa, const void* pb)
{
    const orc_hit* a = (const orc_hit*)pa;
    const orc_hit* b = (const orc_hit*)pb;
    const int an = isnan(a->final), bn = isnan(b->final);
    if (an != bn) return an ? 1 : -1;
    if (!an) {
        if (a->final > b->final) return -1;
        if (a->final < b->final) return 1;
    }
    return (a->doc > b->doc) - (a->doc < b->doc);
}

static int score_one(score_ws* w, uint64_t n_terms,
                     const uint64_t* t_ptr, const uint32_t* t_doc, const float* t_w,
                     const uint64_t* b_ptr, const uint32_t* b_doc, const float* b_w,
                     const double* mag_title, const double* mag_body,
                     int32_t k_topics, const double* prior, const double* topic_probs,
                     const uint32_t* q_terms, int32_t n_q_terms, int32_t query_len,
                     int32_t n_extra, const uint32_t* extra_docs, const float* extra_title,
                     const float* extra_body, const uint8_t* extra_flags,
                     int32_t k, orc_hit* hits, int32_t* n_hits, uint64_t* n_cand_out)
{
    w->n_cand = 0;
    /* main_retrieve.go:55-69 — one getFromInverted result per query token
     * (duplicates fetched and appended twice), weights appended then summed in
     * float64 (:176-182).  Summing as we go is the same float64 sequence. */
    for (int32_t i = 0; i < n_q_terms; i++) {
        const uint32_t t = q_terms[i];
        if ((uint64_t)t >= n_terms) continue;                       /* ErrKeyNotFound => empty */
        if (b_ptr) for (uint64_t p = b_ptr[t]; p < b_ptr[t + 1]; p++) {  /* :226-232 */
            if (b_doc[p] >= w->n_docs || ws_touch(w, b_doc[p])) return -2;
            w->acc_b[b_doc[p]] += (double)b_w[p];
        }
        if (t_ptr) for (uint64_t p = t_ptr[t]; p < t_ptr[t + 1]; p++) {  /* :234-239 */
            if (t_doc[p] >= w->n_docs || ws_touch(w, t_doc[p])) return -2;
            w->acc_t[t_doc[p]] += (double)t_w[p];
        }
    }
    /* main_retrieve.go:73-78 — phrase contributions appended after the terms */
    for (int32_t i = 0; i < n_extra; i++) {
        const uint32_t doc = extra_docs[i];
        if (doc >= w->n_docs || ws_touch(w, doc)) return -2;
        if (extra_flags[i] & 1) w->acc_t[doc] += (double)extra_title[i];
        if (extra_flags[i] & 2) w->acc_b[doc] += (double)extra_body[i];
    }

    if (w->n_cand > w->cap_rows) {
        w->cap_rows = w->n_cand * 2;
        orc_hit* nr = (orc_hit*)realloc(w->rows, sizeof(orc_hit) * w->cap_rows);
        if (!nr) return -1;
        w->rows = nr;
    }
    const double qmag = sqrt((double)query_len);                    /* get_metadata.go:53 */
    for (uint64_t c = 0; c < w->n_cand; c++) {
        const uint32_t doc = w->cand[c];
        double sqd = 0.0;                                            /* :39-42 */
        if (topic_probs && prior)
            for (int32_t t = 0; t < k_topics; t++) sqd += topic_probs[t] * prior[(uint64_t)doc * k_topics + t];
        double body = w->acc_b[doc];
        double title = w->acc_t[doc];
        body /= (mag_body[doc] * qmag);                              /* :57 */
        title /= (mag_title[doc] * qmag);                            /* :58 */
        if (isnan(body)) body = 0;                                   /* :61-63 */
        if (isnan(title)) title = 0;                                 /* :64-66 */
        orc_hit* r = &w->rows[c];
        r->doc = doc; r->_pad = 0;
        r->title = title; r->body = body; r->pagerank = sqd;         /* :68 */
        r->final = (0.33 * sqd + 0.38 * title + 0.29 * body) * 100.0; /* :69 */
        w->acc_b[doc] = 0.0; w->acc_t[doc] = 0.0; w->touched[doc] = 0;
    }
    qsort(w->rows, w->n_cand, sizeof(orc_hit), hit_cmp);            /* util.go:48-54 */
    const uint64_t nh = w->n_cand < (uint64_t)k ? w->n_cand : (uint64_t)k; /* main_retrieve.go:99-103 */
    memcpy(hits, w->rows, sizeof(orc_hit) * nh);
    *n_hits = (int32_t)nh;
    if (n_cand_out) *n_cand_out = w->n_cand;
    return 0;
}

int orc_score_topk(uint64_t n_docs, uint64_t n_terms,
                   const uint64_t* t_ptr, const uint32_t* t_doc, const float* t_w,
                   const uint64_t* b_ptr, const uint32_t* b_doc, const float* b_w,
                   const double* mag_title, const double* mag_body,
                   int32_t k_topics, const double* prior, const double* topic_probs,
                   const uint32_t* q_terms, int32_t n_q_terms, int32_t query_len,
                   int32_t n_extra, const uint32_t* extra_docs, const float* extra_title,
                   const float* extra_body, const uint8_t* extra_flags,
                   int32_t k, orc_hit* hits, int32_t* n_hits, uint64_t* n_cand)
{
    score_ws w;
    if (ws_init(&w, n_docs)) { ws_free(&w); return -1; }
    int rc = score_one(&w, n_terms, t_ptr, t_doc, t_w, b_ptr, b_doc, b_w, mag_title, mag_body,
                       k_topics, prior, topic_probs, q_terms, n_q_terms, query_len,
                       n_extra, extra_docs, extra_title, extra_body, extra_flags,
                       k, hits, n_hits, n_cand);
    ws_free(&w);
    return rc;
}

/* "strong CPU" scoring (B2): the same per-query code, queries spread over all cores (the reference fans
 * out goroutines per term and per candidate, main_retrieve.go:55-57, get_metadata.go:21-23). */
int orc_score_topk_batch_omp(uint64_t n_docs, uint64_t n_terms,
                             const uint64_t* t_ptr, const uint32_t* t_doc, const float* t_w,
                             const uint64_t* b_ptr, const uint32_t* b_doc, const float* b_w,
                             const double* mag_title, const double* mag_body,
                             int32_t k_topics, const double* prior, const double* topic_probs,
                             int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms,
                             const int32_t* query_len, int32_t k, orc_hit* hits, int32_t* n_hits, int32_t* threads_used)
{
    int rc_all = 0;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
    if (nthreads > n_q) nthreads = n_q > 0 ? n_q : 1;
#endif
    if (threads_used) *threads_used = nthreads;
    #pragma omp parallel num_threads(nthreads)
    {
        score_ws w;
        int rc = ws_init(&w, n_docs);
        #pragma omp for schedule(dynamic, 1)
        for (int32_t q = 0; q < n_q; q++) {
            if (rc) continue;
            const int32_t nt = (int32_t)(q_ptr[q + 1] - q_ptr[q]);
            const int32_t ql = query_len ? query_len[q] : nt;
            rc = score_one(&w, n_terms, t_ptr, t_doc, t_w, b_ptr, b_doc, b_w, mag_title, mag_body,
                           k_topics, prior, topic_probs ? topic_probs + (uint64_t)q * k_topics : NULL,
                           q_terms + q_ptr[q], nt, ql, 0, NULL, NULL, NULL, NULL,
                           k, hits + (uint64_t)q * k, n_hits + q, NULL);
        }
        ws_free(&w);
        if (rc) {
            #pragma omp critical
            rc_all = rc;
        }
    }
    return rc_all;
}

int orc_score_topk_batch(uint64_t n_docs, uint64_t n_terms,
                         const uint64_t* t_ptr, const uint32_t* t_doc, const float* t_w,
                         const uint64_t* b_ptr, const uint32_t* b_doc, const float* b_w,
                         const double* mag_title, const double* mag_body,
                         int32_t k_topics, const double* prior, const double* topic_probs,
                         int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms,
                         const int32_t* query_len, int32_t k, orc_hit* hits, int32_t* n_hits)
{
    score_ws w;
    if (ws_init(&w, n_docs)) { ws_free(&w); return -1; }
    int rc = 0;
    for (int32_t q = 0; q < n_q && !rc; q++) {
        const int32_t nt = (int32_t)(q_ptr[q + 1] - q_ptr[q]);
        const int32_t ql = query_len ? query_len[q] : nt;
        rc = score_one(&w, n_terms, t_ptr, t_doc, t_w, b_ptr, b_doc, b_w, mag_title, mag_body,
                       k_topics, prior, topic_probs ? topic_probs + (uint64_t)q * k_topics : NULL,
                       q_terms + q_ptr[q], nt, ql, 0, NULL, NULL, NULL, NULL,
                       k, hits + (uint64_t)q * k, n_hits + q, NULL);
    }
    ws_free(&w);
    return rc;
}

/* --- "reference-shaped" scoring (CPU baseline only) ------------------------------------- */
/* The same arithmetic as score_one, but with the reference's data structures: documents are keyed by their  */
/* 32-character hex docHash in hash maps (aggregatedDocs map[string]Rank_term, main_retrieve.go:61-69; the    */
/* magnitudes behind forw[4].Get(docHash), get_metadata.go:46-50), the per-document weight slices are         */
/* appended then summed (main_retrieve.go:64-66,176-182), and the result list is built by appendSort          */
/* (util.go:48-54): binary search + memmove of Rank_combined-sized rows (util.go:25-36: 152 bytes), cut to    */
/* k at the end (main_retrieve.go:99-103).  Ties stay in arrival order as in the reference (here: map order), */
/* so only the FinalRank sequence — not the order of equal-score docs — is comparable with orc_score_topk.   */

typedef struct { char key[32]; double title, body; uint8_t used; } mag_slot;      /* forw[4] row */
struct orc_magmap { mag_slot* s; uint64_t mask; };

orc_magmap* orc_magmap_build(uint64_t n_docs, const double* mag_title, const double* mag_body)
{
    orc_magmap* m = (orc_magmap*)calloc(1, sizeof(orc_magmap));
    if (!m) return NULL;
    uint64_t cap = 16;
    while (cap < n_docs * 2) cap <<= 1;
    m->s = (mag_slot*)calloc(cap, sizeof(mag_slot));
    if (!m->s) { free(m); return NULL; }
    m->mask = cap - 1;
    char key[32];
    for (uint64_t d = 0; d < n_docs; d++) {
        id_to_key((uint32_t)d, key);
        uint64_t i = hm_hash(key) & m->mask;
        while (m->s[i].used) i = (i + 1) & m->mask;
        memcpy(m->s[i].key, key, 32);
        m->s[i].title = mag_title[d];
        m->s[i].body = mag_body[d];
        m->s[i].used = 1;
    }
    return m;
}
void orc_magmap_free(orc_magmap* m) { if (m) { free(m->s); free(m); } }
static const mag_slot* magmap_get(const orc_magmap* m, const char* key)
{
    uint64_t i = hm_hash(key) & m->mask;
    for (;;) {
        const mag_slot* s = &m->s[i];
        if (!s->used) return NULL;                                   /* missing key: zero values (Q8) */
        if (memcmp(s->key, key, 32) == 0) return s;
        i = (i + 1) & m->mask;
    }
}

#define AGG_INLINE 6
typedef struct {
    char key[32];
    uint32_t doc;                      /* carried for the output row only; never used as a key */
    uint16_t nt, nb;
    float tw[AGG_INLINE], bw[AGG_INLINE];   /* TitleWeights / BodyWeights (appended, util.go:11-17) */
    double t_run, b_run;               /* weights beyond the inline capacity, folded in arrival order */
} agg_entry;
typedef struct { unsigned char bytes[152 - 16]; uint32_t doc; uint32_t pad; double final; } ref_row;   /* sizeof(Rank_combined) */
/* aggregatedDocs as a Go map grows: an index table (doubled when half full) over an entry array (doubled when full);
 * one workspace per thread, re-used from query to query like a long-running server's allocator would */
typedef struct {
    uint32_t* idx; uint64_t idx_cap;   /* 0 = empty, else entry index + 1 */
    agg_entry* ent; uint64_t ent_cap, n_ent;
    ref_row* res; uint64_t res_cap;
} agg_ws;

static void aggws_free(agg_ws* w) { free(w->idx); free(w->ent); free(w->res); memset(w, 0, sizeof(*w)); }
static int aggws_grow_index(agg_ws* w)
{
    const uint64_t cap = w->idx_cap ? w->idx_cap * 2 : 1024;
    uint32_t* ni = (uint32_t*)calloc(cap, sizeof(uint32_t));
    if (!ni) return -1;
    for (uint64_t e = 0; e < w->n_ent; e++) {
        uint64_t h = hm_hash(w->ent[e].key) & (cap - 1);
        while (ni[h]) h = (h + 1) & (cap - 1);
        ni[h] = (uint32_t)e + 1;
    }
    free(w->idx);
    w->idx = ni;
    w->idx_cap = cap;
    return 0;
}
static agg_entry* aggws_at(agg_ws* w, const char* key, uint32_t doc)
{
    if ((w->n_ent + 1) * 2 > w->idx_cap && aggws_grow_index(w)) return NULL;
    uint64_t h = hm_hash(key) & (w->idx_cap - 1);
    for (;;) {
        const uint32_t e = w->idx[h];
        if (!e) break;
        if (memcmp(w->ent[e - 1].key, key, 32) == 0) return &w->ent[e - 1];
        h = (h + 1) & (w->idx_cap - 1);
    }
    if (w->n_ent == w->ent_cap) {
        const uint64_t cap = w->ent_cap ? w->ent_cap * 2 : 1024;
        agg_entry* ne = (agg_entry*)realloc(w->ent, cap * sizeof(agg_entry));
        if (!ne) return NULL;
        w->ent = ne;
        w->ent_cap = cap;
    }
    agg_entry* a = &w->ent[w->n_ent];
    memset(a, 0, sizeof(*a));
    memcpy(a->key, key, 32);
    a->doc = doc;
    w->idx[h] = (uint32_t)(++w->n_ent);
    return a;
}

static int score_one_hashed(agg_ws* w, const orc_magmap* mags, uint64_t n_terms,
                            const uint64_t* t_ptr, const uint32_t* t_doc, const float* t_w,
                            const uint64_t* b_ptr, const uint32_t* b_doc, const float* b_w,
                            const uint32_t* q_terms, int32_t n_q_terms, int32_t query_len,
                            int32_t k, orc_hit* hits, int32_t* n_hits)
{
    /* a fresh map per query (main_retrieve.go:60): entries dropped, index cleared */
    w->n_ent = 0;
    if (w->idx) memset(w->idx, 0, w->idx_cap * sizeof(uint32_t));
    char key[32];
    for (int32_t i = 0; i < n_q_terms; i++) {                        /* one getFromInverted result per token (:55-69) */
        const uint32_t t = q_terms[i];
        if ((uint64_t)t >= n_terms) continue;
        for (int field = 0; field < 2; field++) {                    /* body postings (:226-232), then title (:234-239) */
            const uint64_t* ptr = field ? t_ptr : b_ptr;
            const uint32_t* docs = field ? t_doc : b_doc;
            const float* ws = field ? t_w : b_w;
            for (uint64_t p = ptr[t]; p < ptr[t + 1]; p++) {
                id_to_key(docs[p], key);
                agg_entry* a = aggws_at(w, key, docs[p]);
                if (!a) return -1;
                if (field) { if (a->nt < AGG_INLINE) a->tw[a->nt++] = ws[p]; else a->t_run += (double)ws[p]; }
                else       { if (a->nb < AGG_INLINE) a->bw[a->nb++] = ws[p]; else a->b_run += (double)ws[p]; }
            }
        }
    }
    const uint64_t n_cand = w->n_ent;
    if (n_cand > w->res_cap) {
        ref_row* nr = (ref_row*)realloc(w->res, sizeof(ref_row) * n_cand * 2);
        if (!nr) return -1;
        w->res = nr;
        w->res_cap = n_cand * 2;
    }
    ref_row* res = w->res;                                           /* finalResult (:93) */
    uint64_t n_res = 0;
    const double qmag = sqrt((double)query_len);                     /* get_metadata.go:53 */
    for (uint64_t s = 0; s < w->idx_cap; s++) {                      /* range aggregatedDocs: map order */
        if (!w->idx[s]) continue;
        const agg_entry* a = &w->ent[w->idx[s] - 1];
        double title = 0.0, body = 0.0;                              /* genAggrDocsPipeline :176-182 */
        for (int i = 0; i < a->nt; i++) title += (double)a->tw[i];
        title += a->t_run;      /* exact either way: float32 addends in float64 */
        for (int i = 0; i < a->nb; i++) body += (double)a->bw[i];
        body += a->b_run;
        const mag_slot* m = magmap_get(mags, a->key);                /* forw[4].Get(docHash), get_metadata.go:46-50 */
        body /= ((m ? m->body : 0.0) * qmag);                        /* :57 */
        title /= ((m ? m->title : 0.0) * qmag);                      /* :58 */
        if (isnan(body)) body = 0;                                   /* :61-63 */
        if (isnan(title)) title = 0;                                 /* :64-66 */
        ref_row el;
        memset(&el, 0, sizeof(el));
        el.doc = a->doc;
        el.final = (0.33 * 0.0 + 0.38 * title + 0.29 * body) * 100.0; /* :69, topicProbs nil => sqd = 0 (main_retrieve.go:88) */
        /* appendSort, util.go:48-54: first index whose FinalRank < el.FinalRank, shift the tail, insert */
        uint64_t lo = 0, hi = n_res;
        while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (!(res[mid].final < el.final)) lo = mid + 1; else hi = mid; }
        memmove(&res[lo + 1], &res[lo], sizeof(ref_row) * (n_res - lo));
        res[lo] = el;
        n_res++;
    }
    const uint64_t nh = n_res < (uint64_t)k ? n_res : (uint64_t)k;   /* main_retrieve.go:99-103 */
    for (uint64_t i = 0; i < nh; i++) {
        memset(&hits[i], 0, sizeof(orc_hit));
        hits[i].doc = res[i].doc;
        hits[i].final = res[i].final;
    }
    *n_hits = (int32_t)nh;
    return 0;
}

int orc_score_topk_batch_hashed(const orc_magmap* mags, uint64_t n_terms,
                                const uint64_t* t_ptr, const uint32_t* t_doc, const float* t_w,
                                const uint64_t* b_ptr, const uint32_t* b_doc, const float* b_w,
                                int32_t n_q, const uint32_t* q_ptr, const uint32_t* q_terms, const int32_t* query_len,
                                int32_t k, int32_t use_threads, orc_hit* hits, int32_t* n_hits, int32_t* threads_used)
{
    int rc_all = 0;
    int nthreads = 1;
#ifdef _OPENMP
    if (use_threads) nthreads = omp_get_max_threads();
    if (nthreads > n_q) nthreads = n_q > 0 ? n_q : 1;
#endif
    if (threads_used) *threads_used = nthreads;
    #pragma omp parallel num_threads(nthreads)
    {
        agg_ws w;
        memset(&w, 0, sizeof(w));
        #pragma omp for schedule(dynamic, 1)
        for (int32_t q = 0; q < n_q; q++) {
            const int32_t nt = (int32_t)(q_ptr[q + 1] - q_ptr[q]);
            const int rc = score_one_hashed(&w, mags, n_terms, t_ptr, t_doc, t_w, b_ptr, b_doc, b_w, q_terms + q_ptr[q], nt,
                                            query_len ? query_len[q] : nt, k, hits + (uint64_t)q * k, n_hits + q);
            if (rc) {
                #pragma omp critical
                rc_all = rc;
            }
        }
        aggws_free(&w);
    }
    return rc_all;
}

/* ------------------------------------------------------------------------- */
/* Phrase search — retrieval/phrase.go:11-170, util.go:162-203                 */
/* ------------------------------------------------------------------------- */
static int f32_cmp(const void* a, const void* b)
{
    const float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* util.go:179-203 intersect: sort both (util.go:162-177), two-pointer match.
 * a is overwritten with the intersection; returns its length.  A NULL slice in
 * Go is represented by len < 0. */
static int64_t go_intersect(float* a, int64_t na, float* b, int64_t nb)
{
    if (na < 0 || nb < 0) return -1;                                 /* :180-182 */
    qsort(a, (size_t)na, sizeof(float), f32_cmp);
    qsort(b, (size_t)nb, sizeof(float), f32_cmp);
    int64_t i = 0, j = 0, n = 0;
    while (i != na && j != nb) {                                     /* :191 */
        if (a[i] == b[j]) { a[n++] = a[i]; i++; j++; }
        else if (a[i] > b[j]) j++;
        else i++;
    }
    return n ? n : -1;                                               /* ret stays nil when nothing appended */
}

static int64_t find_doc(const uint32_t* docs, uint64_t beg, uint64_t end, uint32_t doc)
{
    /* posting docs are not required to be sorted for the oracle: linear scan */
    for (uint64_t i = beg; i < end; i++) if (docs[i] == doc) return (int64_t)i;
    return -1;
}

int orc_phrase(uint64_t n_terms,
               const uint64_t* t_ptr, const uint32_t* t_doc, const float* t_w,
               const uint64_t* t_pos_ptr, const float* t_pos,
               const uint64_t* b_ptr, const uint32_t* b_doc, const float* b_w,
               const uint64_t* b_pos_ptr, const float* b_pos,
               const uint32_t* phrase_terms, int32_t n_phrase,
               int32_t cap, uint32_t* out_docs, float* out_title, float* out_body,
               uint8_t* out_flags, int32_t* n_out)
{
    *n_out = 0;
    if (n_phrase <= 0) return 0;
    /* Candidate docs: any doc appearing for phrase term 0 in body or title; a
     * doc must have an entry for EVERY term position (phrase.go:63), so term 0's
     * docs are a superset of the result. */
    const uint32_t t0 = phrase_terms[0];
    if ((uint64_t)t0 >= n_terms) return 0;
    uint64_t ncand = (b_ptr[t0 + 1] - b_ptr[t0]) + (t_ptr[t0 + 1] - t_ptr[t0]);
    uint32_t* cands = (uint32_t*)malloc(sizeof(uint32_t) * (ncand ? ncand : 1));
    if (!cands) return -1;
    uint64_t nc = 0;
    for (uint64_t p = b_ptr[t0]; p < b_ptr[t0 + 1]; p++) cands[nc++] = b_doc[p];
    for (uint64_t p = t_ptr[t0]; p < t_ptr[t0 + 1]; p++) {
        if (find_doc(b_doc, b_ptr[t0], b_ptr[t0 + 1], t_doc[p]) < 0) cands[nc++] = t_doc[p];
    }
    /* deterministic output order: ascending doc id */
    for (uint64_t i = 1; i < nc; i++) { /* insertion sort: oracle sizes are small */
        uint32_t v = cands[i]; uint64_t j = i;
        while (j > 0 && cands[j - 1] > v) { cands[j] = cands[j - 1]; j--; }
        cands[j] = v;
    }

    int rc = 0;
    for (uint64_t c = 0; c < nc && !rc; c++) {
        const uint32_t doc = cands[c];
        float sum_body = 0.0f, sum_title = 0.0f;                     /* phrase.go:59 */
        float* bi = NULL; int64_t nbi = -1;                          /* bodyIntersect (nil) */
        float* ti = NULL; int64_t nti = -1;                          /* titleIntersect (nil) */
        int all_present = 1;
        for (int32_t idx = 0; idx < n_phrase; idx++) {
            const uint32_t t = phrase_terms[idx];
            int64_t pb = -1, pt = -1;
            if ((uint64_t)t < n_terms) {
                pb = find_doc(b_doc, b_ptr[t], b_ptr[t + 1], doc);
                pt = find_doc(t_doc, t_ptr[t], t_ptr[t + 1], doc);
            }
            if (pb < 0 && pt < 0) { all_present = 0; break; }        /* :63 len(termWeights) != lengthPhrase */
            /* getPosTerm :142-163: positions shifted by the term's index in the phrase */
            if (pb >= 0) {
                const uint64_t np = b_pos_ptr[pb + 1] - b_pos_ptr[pb];
                float* sh = (float*)malloc(sizeof(float) * (np ? np : 1));
                for (uint64_t i = 0; i < np; i++) sh[i] = b_pos[b_pos_ptr[pb] + i] - (float)idx; /* :145 */
                sum_body += b_w[pb];                                 /* :69 / :83 */
                if (idx == 0) { bi = sh; nbi = (int64_t)np; }        /* :70 */
                else { nbi = go_intersect(bi, nbi, sh, (int64_t)np); free(sh); } /* :84 */
            } else if (idx > 0) {
                nbi = -1;                                            /* :80-81 */
            }
            if (pt >= 0) {
                const uint64_t np = t_pos_ptr[pt + 1] - t_pos_ptr[pt];
                float* sh = (float*)malloc(sizeof(float) * (np ? np : 1));
                for (uint64_t i = 0; i < np; i++) sh[i] = t_pos[t_pos_ptr[pt] + i] - (float)idx; /* :157 */
                sum_title += t_w[pt];                                /* :73 / :90 */
                if (idx == 0) { ti = sh; nti = (int64_t)np; }        /* :74 */
                else { nti = go_intersect(ti, nti, sh, (int64_t)np); free(sh); } /* :91 */
            } else if (idx > 0) {
                nti = -1;                                            /* :87-88 */
            }
        }
        if (all_present && (nbi > 0 || nti > 0)) {                   /* :97 */
            if (*n_out >= cap) { rc = -3; }
            else {
                const int32_t o = (*n_out)++;
                out_docs[o] = doc;
                out_flags[o] = (uint8_t)((nti > 0 ? 1 : 0) | (nbi > 0 ? 2 : 0));
                out_title[o] = nti > 0 ? sum_title : 0.0f;           /* :102-104 */
                out_body[o] = nbi > 0 ? sum_body : 0.0f;             /* :99-101 */
            }
        }
        free(bi); free(ti);
    }
    free(cands);
    return rc;
}
