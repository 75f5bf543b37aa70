"""GPU parity at BASELINE.json's full sizes (config 2: 2^20 nodes / 5M edges, one vector, to eps 1e-6; config 3: 10M docs /
1M terms / 641M+41M postings; config 4's graph: 10M nodes / 50M edges, 16 topics; config 5: config 3 blended with config 4's ranks), through the oracle where it finishes in seconds and through
size-independent properties elsewhere.  Inputs are generated on the device (nothing is shipped)."""
import numpy as np
import pytest

from spaghettisearch_amd import sharding, synth

pytestmark = pytest.mark.gpu

N, E, K = 10_000_000, 50_000_000, 16
ND, NT, PB, PT = 10_000_000, 1_000_000, 640_000_000, 40_000_000
D = 0.75


def test_pagerank_config4_graph(ss_ctx, oracle):
    import torch
    from spaghettisearch_amd import engine
    dev = torch.device("cuda", 0)
    out_ptr, out_dst = synth.rmat_graph_torch(N, E, seed=42, device=dev)
    n_topic = synth.topic_sizes(N, K)
    g = engine.Graph(ss_ctx, N, out_ptr, out_dst)
    rank, iters = g.pagerank(D, 1e-6, n_topic)                      # the reference's loop, stop rule on the device
    assert rank.shape == (K, N) and np.isfinite(rank).all()
    # (1) the oracle itself on two of the topics (3 iterations of 50M edges: seconds)
    h_ptr = out_ptr.cpu().numpy().view(np.uint64)
    h_dst = out_dst.cpu().numpy().view(np.uint32)
    for k in (0, K - 1):
        ref, ref_it = oracle.pagerank(N, h_ptr, h_dst, D, 1e-6, [int(n_topic[k])])
        assert int(iters[k]) == int(ref_it[0])
        np.testing.assert_allclose(rank[k], ref[0], rtol=1e-12)
    # (2) mass balance of one sweep (pagerank.go:112-117,136-142): every parent hands d*x/outdeg to EACH child but
    #     counts it ONCE in the normaliser S, so  sum(x') * S = d * sum over non-dangling of x + N(1-d)
    pr = engine.PageRankState(g, D, -1.0, n_topic, max_iter=0)
    pr.begin()
    pr.step(3)
    x3 = pr.read()
    pr.step(1)
    x4 = pr.read()
    outdeg = np.diff(h_ptr.astype(np.int64))
    nd = outdeg > 0
    for k in (0, 7, K - 1):
        S = (D * x3[k][nd] / outdeg[nd]).sum() + (1.0 - D) * N
        np.testing.assert_allclose(x4[k].sum() * S, D * x3[k][nd].sum() + (1.0 - D) * N, rtol=1e-10)
    # (3) a state that met the stop rule is a fixed point to the stop tolerance: one more sweep moves it by < eps
    assert np.abs(x4 - x3).sum(axis=1).max() < 1e-6
    pr.close()
    # (4) a topic run on its own and a pair of topics (k_pr_step<1> and <2>: 3.2M non-dangling rows x 64 B leave the caches,
    #     so the narrow block-item kernel is what the default picks here) against the oracle and the 16-wide run
    alone, it1 = g.pagerank(D, 1e-6, [int(n_topic[5])])
    assert int(it1[0]) == int(iters[5])
    np.testing.assert_allclose(alone[0], rank[5], rtol=1e-13)
    pair, it2 = g.pagerank(D, 1e-6, [int(n_topic[0]), int(n_topic[K - 1])])
    assert it2.tolist() == [int(iters[0]), int(iters[K - 1])]
    np.testing.assert_allclose(pair[0], rank[0], rtol=1e-13)
    np.testing.assert_allclose(pair[1], rank[K - 1], rtol=1e-13)
    ref1, ref1_it = oracle.pagerank(N, h_ptr, h_dst, D, 1e-6, [int(n_topic[5])])
    assert int(it1[0]) == int(ref1_it[0])
    np.testing.assert_allclose(alone[0], ref1[0], rtol=1e-12)
    # (5) two doc-range shards (one process plays the all-gather) reproduce the unsharded ranks
    stream = torch.cuda.Stream()
    ss_ctx.set_stream(stream.cuda_stream)
    try:
        with torch.cuda.stream(stream):
            graphs = [engine.Graph(ss_ctx, N, out_ptr, out_dst, rank=r, world=2) for r in range(2)]
            states = [engine.PageRankState(gr, D, 1e-6, n_topic[:4]) for gr in graphs]
            srank, siters = sharding.run_sharded(states, sharding.LocalExchange(states, dev))
            for s in states:
                s.close()
            for gr in graphs:
                gr.close()
    finally:
        torch.cuda.synchronize()
        ss_ctx.set_stream(None)
    assert siters.tolist() == iters[:4].tolist()
    np.testing.assert_allclose(srank, rank[:4], rtol=1e-12)
    g.close()
    del out_ptr, out_dst
    torch.cuda.empty_cache()


def test_pagerank_config2_to_convergence(ss_ctx, oracle):
    """BASELINE config 2: R-MAT scale 20 (N = 2^20), 5M unique edges (seed 42, id permutation seed 43), ONE topic
    vector (the K=1 kernel classes W_WAVE / W_GROUP), d = 0.75, PageRank to eps = 1e-6 — the whole vector against
    the oracle (pagerank.go:85-145): iteration count equal, x to 1e-12, inherited part y = x*S-(1-d) to 1e-6."""
    import torch
    from spaghettisearch_amd import engine
    n, e = 1 << 20, 5_000_000
    dev = torch.device("cuda", 0)
    out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
    h_ptr = out_ptr.cpu().numpy().view(np.uint64)
    h_dst = out_dst.cpu().numpy().view(np.uint32)
    assert int(h_ptr[-1]) == e
    n_topic = synth.topic_sizes(n, 1)
    g = engine.Graph(ss_ctx, n, out_ptr, out_dst)
    rank, iters = g.pagerank(D, 1e-6, n_topic)
    ref, ref_it, change, total = oracle.pagerank_topic_detail(n, h_ptr, h_dst, D, 1e-6, int(n_topic[0]))
    assert int(iters[0]) == int(ref_it) and ref_it > 1
    np.testing.assert_allclose(rank[0], ref, rtol=1e-12)
    y, y_ref = rank[0] * total - (1.0 - D), ref * total - (1.0 - D)
    nz = y_ref > 1e-12
    np.testing.assert_allclose(y[nz], y_ref[nz], rtol=1e-6)
    np.testing.assert_allclose(y, y_ref, atol=1e-9 * y_ref.max())
    # the reference's own stop threshold (start_crawl.go:175: eps = 1e-20): iteration counts equal or off by one at the last bit
    rank20, it20 = g.pagerank(D, 1e-20, n_topic, max_iter=500)
    ref20, rit20 = oracle.pagerank(n, h_ptr, h_dst, D, 1e-20, n_topic, max_iter=500)
    assert abs(int(it20[0]) - int(rit20[0])) <= 1
    np.testing.assert_allclose(rank20[0], ref20[0], rtol=1e-12)
    # fixed-iteration state == the oracle after the same number of sweeps (the mode bench.py times)
    st = engine.PageRankState(g, D, -1.0, n_topic, max_iter=0)
    st.begin()
    st.step(10)
    x10 = st.read()[0]
    st.close()
    ref10, _ = oracle.pagerank(n, h_ptr, h_dst, D, -1.0, n_topic, max_iter=10)
    np.testing.assert_allclose(x10, ref10[0], rtol=1e-12)
    g.close()
    del out_ptr, out_dst
    torch.cuda.empty_cache()


def test_index_config3_build_and_topk(ss_ctx, oracle):
    import torch
    from spaghettisearch_amd import engine
    dev = torch.device("cuda", 0)
    b_ptr, b_doc, b_tf = synth.zipf_index_torch(ND, NT, PB, seed=44, device=dev)
    t_ptr, t_doc, t_tf = synth.zipf_index_torch(ND, NT, PT, seed=144, device=dev)
    bi = engine.InvertedIndex(ss_ctx, ND, b_ptr, b_doc, b_tf)
    ti = engine.InvertedIndex(ss_ctx, ND, t_ptr, t_doc, t_tf)
    wt, mt, idf_t = ti.tfidf_build(ND)
    wb, mb, idf_b = bi.tfidf_build(ND)
    h_bptr = b_ptr.cpu().numpy().view(np.uint64)
    h_tptr = t_ptr.cpu().numpy().view(np.uint64)
    h_bdoc = b_doc.cpu().numpy().view(np.uint32)
    h_tdoc = t_doc.cpu().numpy().view(np.uint32)
    h_btf = b_tf.cpu().numpy()
    # ---- TF-IDF build (term_weighting.go:29-50): idf of sampled terms bit-exact against Go's Log2 restated
    rng = np.random.default_rng(1)
    for t in np.concatenate([np.arange(20), rng.integers(0, NT, 200)]):
        df = int(h_bptr[t + 1] - h_bptr[t])
        if df:
            assert idf_b[t] == np.float32(oracle.go_log2(float(ND) / float(df)))
    # every weight = float32(tf * idf) (:42)
    per_post_idf = np.repeat(idf_b, np.diff(h_bptr.astype(np.int64)))
    assert np.array_equal(wb, h_btf * per_post_idf)
    # sum over docs of magnitude^2 = sum over postings of float64(float32(w*w)) (:44,:72), and sampled docs exactly
    sq = (wb * wb).astype(np.float32).astype(np.float64)
    np.testing.assert_allclose((mb * mb).sum(), sq.sum(), rtol=1e-9)
    sample = rng.integers(0, ND, 64).astype(np.uint32)
    sel = np.isin(h_bdoc, sample)
    acc = np.zeros(ND)
    np.add.at(acc, h_bdoc[sel].astype(np.int64), sq[sel])
    assert np.array_equal(mb[sample], np.sqrt(acc[sample]))
    del per_post_idf, sq, sel, acc, h_btf
    # ---- scoring (main_retrieve.go:50-103, get_metadata.go:31-69): 1024 x 3-term OR, top-100
    sc = engine.Scorer(ss_ctx, ti, bi)
    q_ptr, q_terms = synth.make_queries(1024, 3, 10_000, seed=45)
    hits, n_hits = sc.score_topk(q_ptr, q_terms, 100)
    ns = 24                                                           # the oracle on a sample of the batch (~20 ms per query)
    ref, ref_n = oracle.score_topk_batch(ND, (h_tptr, h_tdoc, wt), (h_bptr, h_bdoc, wb), mt, mb, q_ptr[:ns + 1], q_terms[:3 * ns], 100)
    assert n_hits[:ns].tolist() == ref_n.tolist()
    for f in ("doc", "title", "body", "final"):
        assert np.array_equal(hits[f][:ns], ref[f]), f
    # properties over the whole batch: order, determinism, prefix property of k
    fin = hits["final"]
    assert (n_hits == 100).all()
    assert (np.diff(fin, axis=1) <= 0).all()
    tie = np.diff(fin, axis=1) == 0
    assert (np.diff(hits["doc"].astype(np.int64), axis=1)[tie] > 0).all()          # equal finals: ascending doc id
    again, _ = sc.score_topk(q_ptr, q_terms, 100)
    assert again.tobytes() == hits.tobytes()
    top50, n50 = sc.score_topk(q_ptr, q_terms, 50)
    assert top50.tobytes() == np.ascontiguousarray(hits[:, :50]).tobytes()
    # a batch of one (small slices spread over the chip) and the full batch agree
    one, _ = sc.score_topk(q_ptr[:2], q_terms[:3], 100)
    assert one.tobytes() == hits[:1].tobytes()
    # ---- BASELINE config 5 at full size: the same index and batch blended with a 16-topic PageRank prior and per-query
    #      topicProbs (get_metadata.go:31-42,68-69), against the oracle on a sample of the batch
    out_ptr, out_dst = synth.rmat_graph_torch(N, E, seed=42, device=dev)
    g = engine.Graph(ss_ctx, N, out_ptr, out_dst)
    prior, _ = g.pagerank(D, 1e-6, synth.topic_sizes(N, K))          # [K][N]: node i = doc i
    g.close()
    del out_ptr, out_dst
    torch.cuda.empty_cache()
    sc.set_prior(prior)
    probs = np.random.default_rng(46).dirichlet(np.ones(K), size=1024)
    h5, n5 = sc.score_topk(q_ptr, q_terms, 100, topic_probs=probs)
    n5s = 16
    r5, rn5 = oracle.score_topk_batch(ND, (h_tptr, h_tdoc, wt), (h_bptr, h_bdoc, wb), mt, mb, q_ptr[:n5s + 1], q_terms[:3 * n5s], 100,
                                      prior=np.ascontiguousarray(prior.T), topic_probs=probs[:n5s])
    assert n5[:n5s].tolist() == rn5.tolist()
    for f in ("doc", "title", "body", "pagerank", "final"):
        assert np.array_equal(h5[f][:n5s], r5[f]), f
    assert (n5 == 100).all() and (np.diff(h5["final"], axis=1) <= 0).all()
    # the blend is 0.33*sqd*100 on top of the cosine part: recompute every row's final from its own fields
    fin5 = (0.33 * h5["pagerank"] + 0.38 * h5["title"] + 0.29 * h5["body"]) * 100.0
    assert np.array_equal(fin5, h5["final"])
    sqd = np.einsum("qkt,qt->qk", prior.T[h5["doc"].astype(np.int64)], probs)
    np.testing.assert_allclose(h5["pagerank"], sqd, rtol=1e-12)
    # nil topicProbs with a prior loaded (main_retrieve.go:88): identical to the unblended run
    h0, _ = sc.score_topk(q_ptr, q_terms, 100)
    assert h0.tobytes() == hits.tobytes()
    sc.set_prior(None)
    del prior, sqd
    # ---- two doc-range shards + ss_merge_hits = the unsharded result
    sc.close()
    parts = np.zeros((2, 1024, 100), dtype=engine.HIT_DTYPE)
    pn = np.zeros((2, 1024), dtype=np.int32)
    df_b = np.diff(h_bptr.astype(np.int64)).astype(np.uint64)
    df_t = np.diff(h_tptr.astype(np.int64)).astype(np.uint64)
    for r in range(2):
        lo, hi = sharding.doc_range(ND, r, 2)
        sb = sharding.shard_index_by_docs(b_ptr, b_doc, b_tf, lo, hi)
        st = sharding.shard_index_by_docs(t_ptr, t_doc, t_tf, lo, hi)
        sbi = engine.InvertedIndex(ss_ctx, hi - lo, *sb)
        sti = engine.InvertedIndex(ss_ctx, hi - lo, *st)
        del sb, st
        sbi.set_doc_freq(df_b)
        sti.set_doc_freq(df_t)
        sti.tfidf_build(ND, want_w=False, want_mag=False, want_idf=False)
        sbi.tfidf_build(ND, want_w=False, want_mag=False, want_idf=False)
        ssc = engine.Scorer(ss_ctx, sti, sbi)
        parts[r], pn[r] = ssc.score_topk(q_ptr, q_terms, 100)
        ssc.close()
        sti.close()
        sbi.close()
    merged, mn = ss_ctx.merge_hits(parts, pn, 100, np.array([0, ND // 2], dtype=np.uint32))
    assert merged.tobytes() == hits.tobytes() and mn.tolist() == n_hits.tolist()
    ti.close()
    bi.close()
    del b_ptr, b_doc, b_tf, t_ptr, t_doc, t_tf
    torch.cuda.empty_cache()
