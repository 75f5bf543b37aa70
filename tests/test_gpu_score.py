"""GPU parity: HIP cosine scorer + PageRank blend + top-k (through the C ABI) vs the CPU oracle.

Reference: retrieval/main_retrieve.go:50-103, get_metadata.go:31-69, util.go:48-54.
Gate: top-k doc ids identical (ties by ascending doc id, Q10); scores within 1e-6
relative.  float32 weights summed in float64 are exact and the
library is built without FMA contraction, so the tests assert BIT-EXACT scores.
"""
import math

import numpy as np
import pytest

from spaghettisearch_amd import engine, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[0, 1], ids=["small-kernel-off", "small-kernel-on"])
def _small_query_routing(request, ss_ctx):
    """Every test of this file runs twice: with k_score_small switched off (option "score.small" = 0: the queries of these small
    tables then reach k_score_slices / k_score_wave, the kernels most tests were written for) and with the default routing, where
    queries of up to 1664 postings take k_score_small whatever the call's length; the library's default, 2, sends only short calls
    that consist of such queries there — tests/test_gpu_score.py::test_small_query_kernel_auto_routing)."""
    ss_ctx.set_option("score.small", request.param)
    yield
    ss_ctx.set_option("score.small", None)


def make_scorer(ss_ctx, n_docs, title, body, mag_t, mag_b):
    from spaghettisearch_amd import engine
    ti = engine.InvertedIndex(ss_ctx, n_docs, *title)
    bi = engine.InvertedIndex(ss_ctx, n_docs, *body)
    ti.set_weighted(mag_t)
    bi.set_weighted(mag_b)
    return engine.Scorer(ss_ctx, ti, bi), ti, bi


def close_all(sc, ti, bi):
    sc.close()
    ti.close()
    bi.close()


def assert_same_hits(hits, n_hits, ref, ref_n, exact=True):
    assert n_hits.tolist() == ref_n.tolist()
    for q in range(len(n_hits)):
        n = int(n_hits[q])
        assert hits["doc"][q, :n].tolist() == ref["doc"][q, :n].tolist(), f"query {q}"
        for f in ("title", "body", "pagerank", "final"):
            a, b = hits[f][q, :n], ref[f][q, :n]
            if exact:
                assert np.array_equal(a, b), (q, f, a[:5], b[:5])
            else:
                np.testing.assert_allclose(a, b, rtol=1e-6)


def tiny_index():
    b_ptr = np.array([0, 3, 5, 6], dtype=np.uint64)
    b_doc = np.array([0, 1, 2, 1, 3, 4], dtype=np.uint32)
    b_w = np.array([1.0, 2.0, 0.5, 4.0, 1.0, 3.0], dtype=np.float32)
    t_ptr = np.array([0, 1, 2, 2], dtype=np.uint64)
    t_doc = np.array([1, 3], dtype=np.uint32)
    t_w = np.array([8.0, 2.0], dtype=np.float32)
    mag_b = np.array([2.0, 4.0, 0.0, 1.0, 0.0], dtype=np.float64)
    mag_t = np.array([0.0, 2.0, 0.0, 4.0, 0.0], dtype=np.float64)
    return (t_ptr, t_doc, t_w), (b_ptr, b_doc, b_w), mag_t, mag_b


def test_kat(ss_ctx, oracle):
    # hand-worked cases of tests/test_oracle_kat.py: NaN->0, x/0=+Inf, OR sums, duplicates, unknown term, cut
    title, body, mag_t, mag_b = tiny_index()
    sc, ti, bi = make_scorer(ss_ctx, 5, title, body, mag_t, mag_b)
    try:
        q_terms = np.array([0, 1, 0, 0, 0, 0xFFFFFFFF, 2, 0xFFFFFFFF], dtype=np.uint32)
        q_ptr = np.array([0, 2, 4, 6, 7, 8, 8], dtype=np.uint32)
        for k in (10, 2, 1):
            hits, n_hits = sc.score_topk(q_ptr, q_terms, k)
            ref, ref_n = oracle.score_topk_batch(5, title, body, mag_t, mag_b, q_ptr, q_terms, k)
            assert_same_hits(hits, n_hits, ref, ref_n)
        hits, n_hits = sc.score_topk(q_ptr, q_terms, 10)
        assert hits["doc"][0, :4].tolist() == [2, 1, 3, 0]
        assert math.isinf(hits["final"][0, 0]) and hits["title"][0, 3] == 0.0
        assert n_hits.tolist() == [4, 3, 3, 1, 0, 0]
        # query_len = len(queryTokenised)+len(phraseTokenised) (main_retrieve.go:90)
        qlen = np.array([4, 2, 2, 9, 1, 0], dtype=np.int32)
        hits, n_hits = sc.score_topk(q_ptr, q_terms, 10, query_len=qlen)
        ref, ref_n = oracle.score_topk_batch(5, title, body, mag_t, mag_b, q_ptr, q_terms, 10, query_len=qlen)
        assert_same_hits(hits, n_hits, ref, ref_n)
    finally:
        close_all(sc, ti, bi)


def build_weighted(oracle, n_docs, n_terms, p_body, p_title, seed):
    tp, pd, tf = synth.zipf_index(n_docs, n_terms, p_body, seed=seed)
    tp2, pd2, tf2 = synth.zipf_index(n_docs, n_terms, p_title, seed=seed + 1)
    wb, mb, _ = oracle.tfidf(tp, pd, tf, n_docs, n_docs)
    wt, mt, _ = oracle.tfidf(tp2, pd2, tf2, n_docs, n_docs)
    return (tp2, pd2, wt), (tp, pd, wb), mt, mb


@pytest.mark.parametrize("n_docs,n_terms,p_body,p_title,n_q,k", [
    (400, 120, 4000, 600, 64, 15),
    (50000, 3000, 600000, 40000, 128, 100),      # multi-slice queries, compaction
    (200000, 2000, 1500000, 100000, 48, 1000),   # large k
])
def test_random_index(ss_ctx, oracle, n_docs, n_terms, p_body, p_title, n_q, k):
    title, body, mt, mb = build_weighted(oracle, n_docs, n_terms, p_body, p_title, seed=n_docs)
    sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
    try:
        rng = np.random.default_rng(n_q)
        lens = rng.integers(1, 6, size=n_q)
        q_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        # head-heavy term choice incl. duplicates and a few unknown ids
        q_terms = np.minimum(rng.geometric(0.02, size=int(lens.sum())) - 1, n_terms + 5).astype(np.uint32)
        hits, n_hits = sc.score_topk(q_ptr, q_terms, k)
        ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mt, mb, q_ptr, q_terms, k)
        assert_same_hits(hits, n_hits, ref, ref_n)
        assert ref_n.max() == min(k, ref_n.max())
    finally:
        close_all(sc, ti, bi)


def test_three_term_or_batch_like_config3(ss_ctx, oracle):
    # BASELINE config 3 shape, scaled down: 3 distinct head/torso terms per query, top-100
    n_docs, n_terms = 300000, 20000
    title, body, mt, mb = build_weighted(oracle, n_docs, n_terms, 6000000, 400000, seed=44)
    sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
    try:
        q_ptr, q_terms = synth.make_queries(256, 3, 2000, seed=45)
        hits, n_hits = sc.score_topk(q_ptr, q_terms, 100)
        ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mt, mb, q_ptr, q_terms, 100)
        assert_same_hits(hits, n_hits, ref, ref_n)
    finally:
        close_all(sc, ti, bi)


def test_oversize_window_fallback(ss_ctx, oracle):
    # a short list that is locally far denser than the driver list: the planned window overflows its
    # capacity and the kernel has to bisect the window's doc range (score.hip oversize path)
    n_docs = 1_000_000
    rng = np.random.default_rng(12)
    a = np.sort(rng.choice(n_docs, 9000, replace=False)).astype(np.uint32)            # driver: spread out
    b = (500_000 + np.sort(rng.choice(6000, 4000, replace=False))).astype(np.uint32)  # 4000 docs inside 6000 ids
    c = np.sort(rng.choice(n_docs, 300, replace=False)).astype(np.uint32)
    one = np.uint32(777_777)
    body_docs = [a, b, c, np.array([one], dtype=np.uint32)]
    title_docs = [a[::7], b[::3], np.zeros(0, np.uint32), np.array([one], dtype=np.uint32)]

    def table(lists, seed):
        ptr = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
        doc = np.concatenate(lists).astype(np.uint32)
        return ptr, doc, synth.make_tf(len(doc), np.random.default_rng(seed))
    bt, tt = table(body_docs, 1), table(title_docs, 2)
    wb, mb, _ = oracle.tfidf(*bt, n_docs, n_docs)
    wt, mt, _ = oracle.tfidf(*tt, n_docs, n_docs)
    title, body = (tt[0], tt[1], wt), (bt[0], bt[1], wb)
    sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
    try:
        q_terms = np.array([0, 1, 1, 0, 2, 1, 2, 3, 1, 3], dtype=np.uint32)
        q_ptr = np.array([0, 2, 5, 7, 8, 10], dtype=np.uint32)
        for k in (10, 300):
            hits, n_hits = sc.score_topk(q_ptr, q_terms, k)
            ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mt, mb, q_ptr, q_terms, k)
            assert_same_hits(hits, n_hits, ref, ref_n)
    finally:
        close_all(sc, ti, bi)


def test_pagerank_blend(ss_ctx, oracle):
    # Q9: sqd = sum_t topicProbs[t]*PR[doc][t], weight 0.33 (get_metadata.go:39-42,69); nil probs => 0
    n_docs, n_terms, K = 20000, 1000, 16
    title, body, mt, mb = build_weighted(oracle, n_docs, n_terms, 200000, 20000, seed=7)
    rng = np.random.default_rng(46)
    prior = rng.random((K, n_docs)) * 50.0          # large enough to reorder results
    sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
    try:
        sc.set_prior(prior)
        q_ptr, q_terms = synth.make_queries(64, 3, 300, seed=3)
        probs = rng.dirichlet(np.ones(K), size=64)
        hits, n_hits = sc.score_topk(q_ptr, q_terms, 50, topic_probs=probs)
        ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mt, mb, q_ptr, q_terms, 50,
                                             prior=np.ascontiguousarray(prior.T), topic_probs=probs)
        assert_same_hits(hits, n_hits, ref, ref_n)
        assert (hits["pagerank"][:, 0] > 0).all()
        # without topic_probs the prior is ignored (reference default, main_retrieve.go:88)
        hits0, n0 = sc.score_topk(q_ptr, q_terms, 50)
        ref0, refn0 = oracle.score_topk_batch(n_docs, title, body, mt, mb, q_ptr, q_terms, 50)
        assert_same_hits(hits0, n0, ref0, refn0)
        assert (hits0["pagerank"] == 0).all()
    finally:
        close_all(sc, ti, bi)


def test_end_to_end_offline_then_online(ss_ctx, oracle):
    """start_crawl.go:175-177 order: PageRank -> UpdateTermWeights(title) -> (body), then Retrieve."""
    from spaghettisearch_amd import engine
    n = 30000
    ptr, dst = synth.rmat_graph(n, 150000, seed=9)
    n_topic = synth.topic_sizes(n, 4)
    g = engine.Graph(ss_ctx, n, ptr, dst)
    rank, iters = g.pagerank(0.75, 1e-9, n_topic)
    g.close()
    tp, pd, tf = synth.zipf_index(n, 2000, 300000, seed=1)
    tp2, pd2, tf2 = synth.zipf_index(n, 2000, 30000, seed=2)
    bi = engine.InvertedIndex(ss_ctx, n, tp, pd, tf)
    ti = engine.InvertedIndex(ss_ctx, n, tp2, pd2, tf2)
    wt, mt, _ = ti.tfidf_build(n)        # N = len(forw[3]) = PageRank nodes (Q7)
    wb, mb, _ = bi.tfidf_build(n)
    sc = engine.Scorer(ss_ctx, ti, bi)
    sc.set_prior(rank)
    q_ptr, q_terms = synth.make_queries(32, 3, 500, seed=8)
    probs = np.random.default_rng(1).dirichlet(np.ones(4), size=32)
    hits, n_hits = sc.score_topk(q_ptr, q_terms, 50, topic_probs=probs)
    close_all(sc, ti, bi)
    # oracle pipeline
    r_rank, r_iters = oracle.pagerank(n, ptr, dst, 0.75, 1e-9, n_topic)
    r_wb, r_mb, _ = oracle.tfidf(tp, pd, tf, n, n)
    r_wt, r_mt, _ = oracle.tfidf(tp2, pd2, tf2, n, n)
    assert iters.tolist() == r_iters.tolist()
    assert np.array_equal(wb, r_wb) and np.array_equal(wt, r_wt)
    ref, ref_n = oracle.score_topk_batch(n, (tp2, pd2, r_wt), (tp, pd, r_wb), r_mt, r_mb, q_ptr, q_terms, 50,
                                         prior=np.ascontiguousarray(r_rank.T), topic_probs=probs)
    # magnitudes / ranks differ in the last bits (summation order): ids must match, scores to 1e-9
    assert n_hits.tolist() == ref_n.tolist()
    same = sum(hits["doc"][q, :n_hits[q]].tolist() == ref["doc"][q, :ref_n[q]].tolist() for q in range(32))
    assert same >= 31, same      # a near-tie may legitimately swap (SURVEY.md §8d parity gates)
    for q in range(32):
        nn = int(n_hits[q])
        np.testing.assert_allclose(np.sort(hits["final"][q, :nn]), np.sort(ref["final"][q, :nn]), rtol=1e-9)


def test_errors(ss_ctx, oracle):
    from spaghettisearch_amd import SpaghettiError, engine
    title, body, mag_t, mag_b = tiny_index()
    ti = engine.InvertedIndex(ss_ctx, 5, *title)
    bi = engine.InvertedIndex(ss_ctx, 5, *body)
    with pytest.raises(SpaghettiError) as ei:
        engine.Scorer(ss_ctx, ti, bi)                 # weights not built yet
    assert ei.value.code == 6
    ti.set_weighted(mag_t)
    bi.set_weighted(mag_b)
    sc = engine.Scorer(ss_ctx, ti, bi)
    q_ptr = np.array([0, 1], dtype=np.uint32)
    q_terms = np.array([0], dtype=np.uint32)
    with pytest.raises(SpaghettiError):
        sc.score_topk(q_ptr, q_terms, 0)
    with pytest.raises(SpaghettiError):
        sc.score_topk(q_ptr, q_terms, 5000)
    with pytest.raises(SpaghettiError):
        sc.score_topk(q_ptr, q_terms, 5, topic_probs=np.ones((1, 3)))   # no prior set
    with pytest.raises(SpaghettiError):
        ti.close()                                    # still in use by the scorer
    close_all(sc, ti, bi)


def test_fuzz_many_shapes_and_determinism(ss_ctx, oracle):
    """Random query shapes (1-8 tokens, duplicates, unknown ids), several k, heavy and light lists; the
    batch is scored three times: results must match the oracle bit for bit and be identical run to run."""
    n_docs, n_terms = 120000, 800
    title, body, mt, mb = build_weighted(oracle, n_docs, n_terms, 2500000, 150000, seed=77)
    sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
    try:
        rng = np.random.default_rng(99)
        n_q = 300
        lens = rng.integers(1, 9, size=n_q)
        q_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        q_terms = np.minimum(rng.geometric(0.01, size=int(lens.sum())) - 1, n_terms + 2).astype(np.uint32)
        for k in (1, 7, 100, 513):
            ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mt, mb, q_ptr, q_terms, k)
            first = None
            for _ in range(3):
                hits, n_hits = sc.score_topk(q_ptr, q_terms, k)
                assert_same_hits(hits, n_hits, ref, ref_n)
                if first is None:
                    first = hits.tobytes()
                assert hits.tobytes() == first
    finally:
        close_all(sc, ti, bi)


def test_concurrent_callers_share_one_scorer(ss_ctx, oracle):
    # retrieval.Retrieve runs on one goroutine per HTTP request (cmd/server/server.go:47): many threads score
    # against the same resident index at once; every caller must get exactly its own query's serial result
    import threading
    n_docs, n_terms = 60000, 4000
    title, body, mt, mb = build_weighted(oracle, n_docs, n_terms, 900000, 70000, seed=12)
    sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
    try:
        batches = [synth.make_queries(1 + 3 * i, 3, 800, seed=100 + i) for i in range(8)]
        serial = [sc.score_topk(qp, qt, 50) for qp, qt in batches]
        got = [None] * len(batches)
        errors = []

        def worker(i):
            try:
                for _ in range(5):
                    got[i] = sc.score_topk(batches[i][0], batches[i][1], 50)
            except Exception as exc:                # noqa: BLE001 - reported below
                errors.append(exc)

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(batches))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for (h, n), (rh, rn) in zip(got, serial):
            assert n.tolist() == rn.tolist() and h.tobytes() == rh.tobytes()
    finally:
        close_all(sc, ti, bi)


def test_long_queries_many_lists(ss_ctx, oracle):
    # up to SS_MAX_QUERY_TERMS distinct terms: 2*64 posting lists per query (window plan rows of 129 cursors,
    # general offset search, oversize-window bisection when the plan cannot hold enough windows)
    n_docs, n_terms = 120000, 3000
    title, body, mt, mb = build_weighted(oracle, n_docs, n_terms, 2500000, 200000, seed=3)
    sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
    try:
        rng = np.random.default_rng(8)
        lens = [64, 40, 17, 7, 64]
        q_terms = np.concatenate([rng.choice(400, size=n, replace=False) for n in lens]).astype(np.uint32)
        q_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        for k in (10, 100, 300):
            hits, n_hits = sc.score_topk(q_ptr, q_terms, k)
            ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mt, mb, q_ptr, q_terms, k)
            assert_same_hits(hits, n_hits, ref, ref_n)
    finally:
        close_all(sc, ti, bi)


def test_fused_and_separate_merge_agree(ss_ctx, oracle):
    # the per-query merge runs inside k_score_slices (last slice of the query to finish); option "score.separate_merge" keeps
    # it as its own launch: same hits, bit for bit, for single- and multi-slice queries, twice in a row (tickets reset)
    n_docs, n_terms = 300000, 2000
    title = synth.zipf_index(n_docs, n_terms, 400000, seed=31)
    body = synth.zipf_index(n_docs, n_terms, 6000000, seed=32)
    sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, np.ones(n_docs), np.ones(n_docs))
    try:
        q_ptr, q_terms = synth.make_queries(96, 3, 40, seed=33)          # head terms: several slices per query
        q2_ptr, q2_terms = synth.make_queries(96, 2, n_terms, seed=34)
        runs = []
        for mode in ("fused", "fused", "separate", "fused"):
            ss_ctx.set_option("score.separate_merge", 1 if mode == "separate" else None)
            runs.append((sc.score_topk(q_ptr, q_terms, 100), sc.score_topk(q2_ptr, q2_terms, 7)))
        ref = oracle.score_topk_batch(n_docs, title, body, np.ones(n_docs), np.ones(n_docs), q_ptr, q_terms, 100)
        assert_same_hits(*runs[0][0], *ref)
        for r in runs[1:]:
            for (h, n), (h0, n0) in zip(r, runs[0]):
                assert np.array_equal(n, n0) and h.tobytes() == h0.tobytes()
    finally:
        close_all(sc, ti, bi)


def _lists_table(lists, seed):
    ptr = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
    doc = np.concatenate(lists).astype(np.uint32)
    return ptr, doc, synth.make_tf(len(doc), np.random.default_rng(seed))


@pytest.mark.parametrize("wave_target", [None, 1024, 40000])
def test_wave_kernel_forced_on_small_tables(ss_ctx, oracle, wave_target):
    """k_score_wave (one wave per slice) takes a batch only when its queries suit it, which at test sizes they do not; option
    "score.wave_min_list" = 0 drops the demand on list lengths so that the kernel runs here — whatever the slice size — and must
    agree with the oracle and with k_score_slices ("score.wave" = 0) bit for bit, with and without the PageRank blend."""
    n_docs, n_terms = 300000, 20000
    title, body, mt, mb = build_weighted(oracle, n_docs, n_terms, 6000000, 400000, seed=44)
    sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
    try:
        q_ptr, q_terms = synth.make_queries(192, 3, 300, seed=46)              # every list > 1024 postings
        rng = np.random.default_rng(3)
        prior = rng.random((4, n_docs)) * 1e-3
        probs = rng.dirichlet(np.ones(4), size=192)
        for blend in (False, True):
            sc.set_prior(prior if blend else None)
            kw = {"topic_probs": probs} if blend else {}
            ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mt, mb, q_ptr, q_terms, 100,
                                                 **({"prior": np.ascontiguousarray(prior.T), "topic_probs": probs} if blend else {}))
            with ss_ctx.options(score__wave_min_list=0, score__wave_slice_target=wave_target):
                hits, n_hits = sc.score_topk(q_ptr, q_terms, 100, **kw)
            with ss_ctx.options(score__wave=0):
                hits0, n0 = sc.score_topk(q_ptr, q_terms, 100, **kw)
            assert_same_hits(hits, n_hits, ref, ref_n)
            assert hits.tobytes() == hits0.tobytes() and n_hits.tolist() == n0.tolist()
    finally:
        close_all(sc, ti, bi)


@pytest.mark.parametrize("grade", [(0, 115, 40), (50, 200, 25), (99, 300, 10)])
def test_wave_graded_slices_agree(ss_ctx, oracle, grade):
    """The wave kernel's slices are graded (the first part of a batch's postings in larger slices, the rest in smaller ones,
    options "score.wave_big_pct" / "_big_x100" / "_small_x100"): how a query's doc range is cut must not show in the hits."""
    n_docs, n_terms = 300000, 20000
    title, body, mt, mb = build_weighted(oracle, n_docs, n_terms, 6000000, 400000, seed=47)
    sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
    try:
        q_ptr, q_terms = synth.make_queries(160, 3, 300, seed=48)
        ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mt, mb, q_ptr, q_terms, 50)
        with ss_ctx.options(score__wave_min_list=0, score__wave_slice_target=6000, score__wave_big_pct=grade[0],
                            score__wave_big_x100=grade[1], score__wave_small_x100=grade[2]):
            hits, n_hits = sc.score_topk(q_ptr, q_terms, 50)
        assert_same_hits(hits, n_hits, ref, ref_n)
    finally:
        close_all(sc, ti, bi)


def test_batches_in_flight_with_host_results(ss_ctx, oracle):
    """ss_score_topk_submit / ss_score_topk_collect: up to SS_SCORE_INFLIGHT batches in flight, hits to host memory.  Batches of changing
    size and k are submitted ahead and collected late and out of order; every batch must equal the oracle's.  A fourth submit is
    refused, an unknown or spent ticket too, and a synchronous call between submits does not disturb the batches in flight."""
    from spaghettisearch_amd import SpaghettiError
    n_docs, n_terms = 300000, 20000
    title, body, mt, mb = build_weighted(oracle, n_docs, n_terms, 6000000, 400000, seed=52)
    sc = ti = bi = None
    try:
        sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
        batches = [synth.make_queries(64 + 40 * (i % 4), 3, 300, seed=80 + i) for i in range(9)]
        ks = [40, 10, 100, 40, 7, 40, 64, 40, 40]
        refs = [oracle.score_topk_batch(n_docs, title, body, mt, mb, qp, qt, k) for (qp, qt), k in zip(batches, ks)]
        with ss_ctx.options(score__wave_min_list=0):
            flight = []
            done = {}
            for i, ((qp, qt), k) in enumerate(zip(batches, ks)):
                if len(flight) == 3:
                    with pytest.raises(SpaghettiError):
                        sc.submit(qp, qt, k)                           # SS_SCORE_INFLIGHT batches out already
                    j, t = flight.pop(1 if i % 2 else 0)               # collect out of order
                    done[j] = sc.collect(t)
                    with pytest.raises(SpaghettiError):
                        sc.collect(t)                                  # spent
                flight.append((i, sc.submit(qp, qt, k)))
                if i == 4:                                             # a synchronous call in between
                    hits, n_hits = sc.score_topk(batches[0][0], batches[0][1], ks[0])
                    assert_same_hits(hits, n_hits, *refs[0])
            for j, t in reversed(flight):
                done[j] = sc.collect(t)
            with pytest.raises(SpaghettiError):
                sc.collect((12345, 1, 1))
            for j in range(len(batches)):
                assert_same_hits(done[j][0], done[j][1], *refs[j])
            # an empty batch has a ticket too
            t = sc.submit(np.zeros(1, np.uint32), np.zeros(0, np.uint32), 5)
            hits, n_hits = sc.collect(t)
            assert hits.shape == (0, 5) and n_hits.shape == (0,)
    finally:
        close_all(sc, ti, bi)


@pytest.mark.parametrize("pipeline,wave", [(1, 1), (0, 1), (1, 0), (0, 0), (2, 2)])
def test_pipelined_batches_agree(ss_ctx, oracle, pipeline, wave):
    """(pipeline = 0: the same stream of calls in the default mode — batches of different sizes back to back reuse and regrow the
    per-turn device buffers while earlier batches are still running, which once went unguarded.)
    Option "score.pipeline" (default 1 since round 4): a batch's k_wave_prep / k_score_wave run on the context's wave stream and its
    k_merge_flat on the caller's stream behind an event, so the next batch's k_score_wave starts under this batch's merge — and the
    hits are still complete IN STREAM ORDER: right behind every call the test enqueues a copy of the output buffers on the same
    stream (no synchronize in between) and checks the COPIES against the oracle.  Twelve batches of changing size back to back
    (stream shared with the caller, so the calls only enqueue); then a host-output call and a prior change right behind
    pipelined ones."""
    import torch
    n_docs, n_terms = 300000, 20000
    title, body, mt, mb = build_weighted(oracle, n_docs, n_terms, 6000000, 400000, seed=51)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ss_ctx.set_stream(stream.cuda_stream)
    sc = ti = bi = None
    try:
        with torch.cuda.stream(stream):
            sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
            batches = [synth.make_queries(96 + 48 * (i % 5), 3, 300, seed=60 + i) for i in range(12)]
            if wave == 2:
                # SPLIT batches: 3-term queries (wave kernel) and 8-term queries (more than score.wave_max_terms: k_score_slices, on a
                # second internal stream, unfused) in one batch
                split = []
                for i, (qp, qt) in enumerate(batches):
                    lp, lt = synth.make_queries(40 + 8 * (i % 3), 8, 300, seed=160 + i)
                    split.append((np.concatenate([qp, lp[1:] + qp[-1]]).astype(np.uint32), np.concatenate([qt, lt]).astype(np.uint32)))
                batches = split
                wave = 1
            k = 40
            outs = [(torch.zeros(len(qp) * k * 40, dtype=torch.uint8, device=dev), torch.zeros(len(qp), dtype=torch.int32, device=dev))
                    for qp, _ in batches]
            # (wave = 0: every batch is all k_score_slices — pipelined too since late round 4: the slices kernel on an internal stream,
            #  k_merge_topk on the caller's stream behind an event)
            with ss_ctx.options(score__wave_min_list=0, score__pipeline=pipeline, score__wave=wave):
                snaps = []
                for (qp, qt), out in zip(batches, outs):
                    sc.score_topk(qp, qt, k, out=out)
                    snaps.append((out[0].clone(), out[1].clone()))      # ordered behind the call on the shared stream, nothing else
                    out[0].fill_(0xEE)                                   # ... and the buffer is scribbled over right behind the copy
                stream.synchronize()
                for (qp, qt), (dh, dn) in zip(batches, snaps):
                    nq = len(qp) - 1
                    ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mt, mb, qp, qt, k)
                    hits = dh.cpu().numpy()[: nq * k * 40].view(engine.HIT_DTYPE).reshape(nq, k)
                    assert_same_hits(hits, dn.cpu().numpy()[:nq], ref, ref_n)
                # a host-output call right behind two pipelined ones, and a prior change behind a pipelined one
                qp, qt = batches[0]
                sc.score_topk(qp, qt, k, out=outs[0])
                sc.score_topk(batches[1][0], batches[1][1], k, out=outs[1])
                hits, n_hits = sc.score_topk(qp, qt, k)
                ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mt, mb, qp, qt, k)
                assert_same_hits(hits, n_hits, ref, ref_n)
                sc.score_topk(qp, qt, k, out=outs[0])
                rng = np.random.default_rng(5)
                prior = rng.random((4, n_docs)) * 1e-3
                sc.set_prior(prior)
                ss_ctx.synchronize()
                nq = len(qp) - 1
                hits0 = outs[0][0].cpu().numpy()[: nq * k * 40].view(engine.HIT_DTYPE).reshape(nq, k)
                assert_same_hits(hits0, outs[0][1].cpu().numpy()[:nq], ref, ref_n)       # scored before the prior changed
    finally:
        for x in (sc, ti, bi):
            if x is not None:
                x.close()
        ss_ctx.set_stream(None)


def test_wave_round_makes_progress_beside_much_denser_lists(ss_ctx, oracle):
    """Regression: a list a hundred times sparser than its neighbours got ONE skip entry per planning round; after a round that
    ended at one of its block boundaries that entry equalled the round's start, the round ended where it began and the wave
    never returned (config-3 index, term ranks U[1,3000], slices above 32k postings).  Dense + sparse lists, large slices."""
    n_docs = 2_000_000
    rng = np.random.default_rng(21)
    dense1 = np.sort(rng.choice(n_docs, 500_000, replace=False)).astype(np.uint32)
    dense2 = np.sort(rng.choice(n_docs, 450_000, replace=False)).astype(np.uint32)
    sparse = np.sort(rng.choice(n_docs, 6_000, replace=False)).astype(np.uint32)
    mid = np.sort(rng.choice(n_docs, 40_000, replace=False)).astype(np.uint32)
    bt = _lists_table([dense1, dense2, sparse, mid], 1)
    tt = _lists_table([dense1[::9], dense2[::11], sparse[::2], mid[::5]], 2)
    wb, mb, _ = oracle.tfidf(*bt, n_docs, n_docs)
    wt, mt, _ = oracle.tfidf(*tt, n_docs, n_docs)
    title, body = (tt[0], tt[1], wt), (bt[0], bt[1], wb)
    sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
    try:
        q_terms = np.array([0, 1, 2, 0, 2, 3, 1, 2, 0, 1, 3], dtype=np.uint32)
        q_ptr = np.array([0, 3, 6, 8, 11], dtype=np.uint32)
        ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mt, mb, q_ptr, q_terms, 100)
        for target in (40000, 49152, 12000):
            with ss_ctx.options(score__wave_min_list=0, score__wave_slice_target=target):
                hits, n_hits = sc.score_topk(q_ptr, q_terms, 100)
            assert_same_hits(hits, n_hits, ref, ref_n)
    finally:
        close_all(sc, ti, bi)


def test_wave_kernel_forced_many_terms(ss_ctx, oracle):
    """Queries of up to 12 distinct terms in k_score_wave (by default it takes queries of at most 6: more lists leave a window
    one driver block and most windows go through the slow path — slower, but it must still be exact), duplicates included."""
    n_docs, n_terms = 300000, 20000
    title, body, mt, mb = build_weighted(oracle, n_docs, n_terms, 6000000, 400000, seed=44)
    sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
    try:
        rng = np.random.default_rng(17)
        lens = rng.integers(7, 15, size=48)
        q_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        q_terms = rng.integers(0, 200, size=int(lens.sum())).astype(np.uint32)        # head terms, some drawn twice
        ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mt, mb, q_ptr, q_terms, 50)
        with ss_ctx.options(score__wave_min_list=0, score__wave_max_terms=12):
            hits, n_hits = sc.score_topk(q_ptr, q_terms, 50)
        assert_same_hits(hits, n_hits, ref, ref_n)
    finally:
        close_all(sc, ti, bi)


def test_small_query_kernel_edges(ss_ctx, oracle):
    """k_score_small (one workgroup per query; every posting exactly, no filter): the cap boundary (a query of exactly / one over
    score.small_cap postings), duplicates (Q8 multiplicity), unknown and repeated unknown terms, empty queries, hostile inputs the
    filter of the other kernels would need switched off (negative and NaN weights, zero / NaN / infinite magnitudes), a blended
    prior with hostile probabilities, k = 1 / 100 / 256 (its largest) and 257 (one over: the other kernels), one batch that holds
    small, wave and slice queries at once; and the same hits whether the small kernel or the big ones score a query."""
    rng = np.random.default_rng(7)
    n_docs, n_terms = 60000, 500
    title, body, mt, mb = build_weighted(oracle, n_docs, n_terms, 900000, 60000, seed=5)
    # hostile corner: a few negative / NaN weights and odd magnitudes
    bw = body[2].copy()
    bw[rng.integers(0, len(bw), 50)] *= -1.0
    bw[rng.integers(0, len(bw), 5)] = np.nan
    body_h = (body[0], body[1], bw)
    mb_h = mb.copy()
    mb_h[rng.integers(0, n_docs, 40)] = 0.0
    mb_h[rng.integers(0, n_docs, 5)] = np.nan
    mb_h[rng.integers(0, n_docs, 5)] = np.inf
    b_len = np.diff(body[0].astype(np.int64)) + np.diff(title[0].astype(np.int64))
    order = np.argsort(b_len)
    for (tt, bb, m_t, m_b), exact in (((title, body, mt, mb), True), ((title, body_h, mt, mb_h), True)):
        sc, ti, bi = make_scorer(ss_ctx, n_docs, tt, bb, m_t, m_b)
        try:
            qs = []
            qs.append([])                                                # no token at all
            qs.append([n_terms + 5])                                     # only unknown words
            qs.append([int(order[3]), int(order[3]), n_terms + 1, int(order[3])])   # a triple token beside an unknown one
            qs += [list(rng.choice(order[:200], size=int(rng.integers(1, 6)))) for _ in range(40)]        # tail terms
            qs += [list(rng.choice(order[-30:], size=3)) for _ in range(6)]                                # head terms: the big kernels
            qs += [[int(order[-1]), int(order[5])], [int(order[100]), int(order[101]), int(order[-2])]]    # mixed
            q_ptr = np.concatenate([[0], np.cumsum([len(x) for x in qs])]).astype(np.uint32)
            q_terms = np.array([t for x in qs for t in x], dtype=np.uint32)
            for k in (1, 100, 256, 257):
                ref, ref_n = oracle.score_topk_batch(n_docs, tt, bb, m_t, m_b, q_ptr, q_terms, k)
                hits, n_hits = sc.score_topk(q_ptr, q_terms, k)
                assert_same_hits(hits, n_hits, ref, ref_n, exact=exact)
                with ss_ctx.options(score__small=0):
                    h0, n0 = sc.score_topk(q_ptr, q_terms, k)
                assert h0.tobytes() == hits.tobytes() and n0.tolist() == n_hits.tolist()
            # the cap boundary: shrink the cap to the postings of one query, then to one below
            tq = [int(order[150]), int(order[151]), int(order[152])]
            tot = int(b_len[tq].sum())
            qp, qt = np.array([0, 3], dtype=np.uint32), np.array(tq, dtype=np.uint32)
            ref, ref_n = oracle.score_topk_batch(n_docs, tt, bb, m_t, m_b, qp, qt, 10)
            for cap in (tot, tot - 1, 0):
                with ss_ctx.options(score__small_cap=cap):
                    hits, n_hits = sc.score_topk(qp, qt, 10)
                assert_same_hits(hits, n_hits, ref, ref_n, exact=exact)
            # blended with a prior and hostile topic probabilities (negative, zero): exact stage semantics, no filter to switch off
            prior = rng.random((4, n_docs)) * 1e-3
            prior[:, rng.integers(0, n_docs, 20)] = -1.0
            sc.set_prior(prior)
            probs = rng.dirichlet(np.ones(4), size=len(qs))
            probs[3] = [-0.5, 0.0, 2.0, 0.1]
            ref, ref_n = oracle.score_topk_batch(n_docs, tt, bb, m_t, m_b, q_ptr, q_terms, 50, prior=np.ascontiguousarray(prior.T), topic_probs=probs)
            hits, n_hits = sc.score_topk(q_ptr, q_terms, 50, topic_probs=probs)
            assert_same_hits(hits, n_hits, ref, ref_n, exact=exact)
            sc.set_prior(None)
        finally:
            close_all(sc, ti, bi)


def test_small_query_kernel_selection_ties_and_many_lists(ss_ctx, oracle):
    """k_score_small picks its k best by radix selection on {FinalRank, doc id}: whole tiers of EQUAL FinalRanks (the selection then has
    to go on through the doc id's bytes), k at / around the tier sizes and the buffer sizes (63, 64, 65, 128, 129, 256), and queries
    with more distinct terms than its list table holds (they take the other kernels) — against the oracle and against the big kernels."""
    rng = np.random.default_rng(11)
    n_docs, n_terms = 20000, 12
    # terms 0..2: 700 / 500 / 300 postings of weight exactly 1 over documents of magnitude 1: every document with the same set of
    # terms has the same FinalRank; terms 3..11: 40 postings each (an 11-term query has 22 lists with the title's)
    lens = [700, 500, 300] + [40] * 9
    docs = [np.sort(rng.choice(n_docs, size=n, replace=False)).astype(np.uint32) for n in lens]
    b_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    b_doc = np.concatenate(docs)
    b_w = np.ones(len(b_doc), dtype=np.float32)
    t_lens = [50, 0, 30] + [5] * 9
    t_docs = [np.sort(rng.choice(n_docs, size=n, replace=False)).astype(np.uint32) for n in t_lens]
    t_ptr = np.concatenate([[0], np.cumsum(t_lens)]).astype(np.uint64)
    t_doc = np.concatenate(t_docs)
    t_w = np.ones(len(t_doc), dtype=np.float32)
    mag = np.ones(n_docs, dtype=np.float64)
    title, body = (t_ptr, t_doc, t_w), (b_ptr, b_doc, b_w)
    sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mag, mag)
    try:
        qs = [[0], [1], [0, 1], [0, 1, 2], [2, 2, 1], list(range(3, 12)), list(range(0, 12)), [0, 3, 4, 5, 6, 7, 8, 9, 10]]
        q_ptr = np.concatenate([[0], np.cumsum([len(x) for x in qs])]).astype(np.uint32)
        q_terms = np.array([t for x in qs for t in x], dtype=np.uint32)
        for k in (1, 63, 64, 65, 100, 128, 129, 256):
            ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mag, mag, q_ptr, q_terms, k)
            hits, n_hits = sc.score_topk(q_ptr, q_terms, k)
            assert_same_hits(hits, n_hits, ref, ref_n)
            with ss_ctx.options(score__small=0):
                h0, n0 = sc.score_topk(q_ptr, q_terms, k)
            assert h0.tobytes() == hits.tobytes() and n0.tolist() == n_hits.tolist()
    finally:
        close_all(sc, ti, bi)


def test_small_query_kernel_auto_routing(ss_ctx, oracle):
    """The default routing ("score.small" = 2): a call of at most "score.small_max_batch" queries that are all small takes k_score_small
    in ONE launch (queries of both table sizes together), any other call takes the big kernels — same hits either way, and equal to
    the oracle's."""
    rng = np.random.default_rng(3)
    n_docs, n_terms = 60000, 500
    title, body, mt, mb = build_weighted(oracle, n_docs, n_terms, 900000, 60000, seed=5)
    b_len = np.diff(body[0].astype(np.int64)) + np.diff(title[0].astype(np.int64))
    order = np.argsort(b_len)
    sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
    try:
        small_terms = order[:250]                                     # a 3-term query of these stays far below the cap
        mid_terms = order[(b_len[order] > 300) & (b_len[order] < 500)]
        for n_q, pool in ((1, small_terms), (5, small_terms), (9, mid_terms), (64, small_terms), (65, small_terms)):
            qs = [list(rng.choice(pool, size=3, replace=False)) for _ in range(n_q)]
            if n_q == 5:
                qs[2] = [int(order[-1]), int(order[3])]                 # one long query: the whole call takes the big kernels
            q_ptr = np.concatenate([[0], np.cumsum([len(x) for x in qs])]).astype(np.uint32)
            q_terms = np.array([t for x in qs for t in x], dtype=np.uint32)
            ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mt, mb, q_ptr, q_terms, 20)
            with ss_ctx.options(score__small=2):
                hits, n_hits = sc.score_topk(q_ptr, q_terms, 20)
            assert_same_hits(hits, n_hits, ref, ref_n)
            with ss_ctx.options(score__small=0):
                h0, n0 = sc.score_topk(q_ptr, q_terms, 20)
            assert h0.tobytes() == hits.tobytes() and n0.tolist() == n_hits.tolist()
    finally:
        close_all(sc, ti, bi)


@pytest.mark.parametrize("mix", ["small+slices", "small+wave", "small+wave+slices", "all-small"])
def test_small_queries_staged_in_pipelined_batches(ss_ctx, oracle, mix):
    """Option "score.small_batch" = 1: in a longer call with device outputs the small queries take k_score_small on an INTERNAL stream
    (beside the slices kernel, beside the wave kernel on a side stream, or alone), their rows go to a staging block and k_small_copy
    moves them into the caller's buffer on the caller's stream — so the hits are still complete in stream order: copies enqueued right
    behind every call (buffers scribbled over behind the copies) must equal the oracle, batch after batch of changing size."""
    import torch
    n_docs, n_terms = 300000, 20000
    title, body, mt, mb = build_weighted(oracle, n_docs, n_terms, 6000000, 400000, seed=51)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ss_ctx.set_stream(stream.cuda_stream)
    rng = np.random.default_rng(77)
    sc = ti = bi = None
    try:
        with torch.cuda.stream(stream):
            sc, ti, bi = make_scorer(ss_ctx, n_docs, title, body, mt, mb)
            batches = []
            for i in range(10):
                parts = [rng.integers(2000, n_terms, size=(70 + 20 * (i % 4), 3))]                  # tail terms: small queries
                if "wave" in mix:
                    parts.append(np.stack([rng.choice(300, size=3, replace=False) for _ in range(40 + 8 * (i % 3))]))
                rows = [list(map(int, r)) for p_ in parts for r in p_]
                if "slices" in mix:
                    rows += [list(map(int, rng.choice(300, size=8, replace=False))) for _ in range(12 + 4 * (i % 2))]   # 8 terms: k_score_slices
                order = rng.permutation(len(rows))
                rows = [rows[j] for j in order]
                qp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint32)
                qt = np.array([t for r in rows for t in r], dtype=np.uint32)
                batches.append((qp, qt))
            k = 40
            outs = [(torch.zeros(len(qp) * k * 40, dtype=torch.uint8, device=dev), torch.zeros(len(qp), dtype=torch.int32, device=dev))
                    for qp, _ in batches]
            with ss_ctx.options(score__wave_min_list=0, score__small=2, score__small_batch=1):
                snaps = []
                for (qp, qt), out in zip(batches, outs):
                    sc.score_topk(qp, qt, k, out=out)
                    snaps.append((out[0].clone(), out[1].clone()))
                    out[0].fill_(0xEE)
                    out[1].fill_(-7)
                stream.synchronize()
                for (qp, qt), (dh, dn) in zip(batches, snaps):
                    nq = len(qp) - 1
                    ref, ref_n = oracle.score_topk_batch(n_docs, title, body, mt, mb, qp, qt, k)
                    hits = dh.cpu().numpy()[: nq * k * 40].view(engine.HIT_DTYPE).reshape(nq, k)
                    assert_same_hits(hits, dn.cpu().numpy()[:nq], ref, ref_n)
    finally:
        for x in (sc, ti, bi):
            if x is not None:
                x.close()
        ss_ctx.set_stream(None)
