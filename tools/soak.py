"""Randomised parity soak on the GPU box (not part of the test suite): many seeds, odd shapes, hostile values.
Every round compares the HIP path with the CPU oracle bit for bit (scoring, TF-IDF) or to 1e-12 (PageRank).

    gpurun --timeout 900 -- 'python tools/soak.py --seconds 600'
"""
import argparse, sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import pyoracle
from spaghettisearch_amd import engine, sharding, synth


def same_hits(hits, n_hits, ref, ref_n, what):
    assert n_hits.tolist() == ref_n.tolist(), what
    for q in range(len(n_hits)):
        n = int(n_hits[q])
        assert hits["doc"][q, :n].tolist() == ref["doc"][q, :n].tolist(), (what, q)
        for f in ("title", "body", "pagerank", "final"):
            a, b = hits[f][q, :n], ref[f][q, :n]
            assert np.array_equal(a, b, equal_nan=True), (what, q, f)


def round_scoring(ctx, rng):
    # (round 5) k_score_small on or off for the whole round, sometimes with a small cap so that a batch holds queries of both kinds
    ctx.set_option("score.small", int(rng.integers(0, 3)))                 # off / every query that fits / short all-small calls (the default)
    ctx.set_option("score.small_cap", int(rng.choice([1664, 1664, 300, 40])))
    try:
        _round_scoring(ctx, rng)
    finally:
        ctx.set_option("score.small", None)
        ctx.set_option("score.small_cap", None)


def _round_scoring(ctx, rng):
    n_docs = int(rng.choice([1, 2, 7, 300, 5000, 60000]))
    n_terms = int(rng.integers(1, 400))
    pb = int(min(n_docs * n_terms // 3 + 1, rng.integers(1, 800000)))
    pt = int(min(n_docs * n_terms // 3 + 1, rng.integers(1, 60000)))
    body = synth.zipf_index(n_docs, n_terms, pb, seed=int(rng.integers(1 << 30)))
    title = synth.zipf_index(n_docs, n_terms, pt, seed=int(rng.integers(1 << 30)))
    total = int(rng.choice([n_docs, max(1, n_docs // 3), n_docs * 2 + 5]))        # N != #indexed docs, also idf < 0 (Q7)
    wt, mt, _ = pyoracle.tfidf(*title, total, n_docs)
    wb, mb, _ = pyoracle.tfidf(*body, total, n_docs)
    ti = engine.InvertedIndex(ctx, n_docs, *title)
    bi = engine.InvertedIndex(ctx, n_docs, *body)
    gwt, gmt, _ = ti.tfidf_build(total)
    gwb, gmb, _ = bi.tfidf_build(total)
    assert np.array_equal(gwt.view(np.uint32), wt.view(np.uint32)) and np.array_equal(gwb.view(np.uint32), wb.view(np.uint32))
    assert np.array_equal(gmt, mt, equal_nan=True) and np.array_equal(gmb, mb, equal_nan=True)
    sc = engine.Scorer(ctx, ti, bi)
    kt = int(rng.choice([0, 1, 3, 16]))
    prior = probs = None
    n_q = int(rng.integers(1, 200))
    if kt:
        prior = rng.standard_normal((kt, n_docs)) * 10.0 ** rng.integers(-6, 3)
        if rng.random() < 0.2:
            prior[rng.integers(kt), rng.integers(n_docs)] = np.nan
        if rng.random() < 0.2:
            prior[rng.integers(kt), rng.integers(n_docs)] = np.inf
        sc.set_prior(prior)
        probs = rng.standard_normal((n_q, kt)) if rng.random() < 0.5 else rng.dirichlet(np.ones(kt), size=n_q)
    lens = rng.integers(0, 9, size=n_q)
    q_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    q_terms = np.minimum(rng.geometric(rng.choice([0.5, 0.05, 0.01]), size=int(lens.sum())) - 1, n_terms + 2).astype(np.uint32)
    qlen = None if rng.random() < 0.5 else rng.integers(0, 12, size=n_q).astype(np.int32)
    for k in rng.choice([1, 2, 50, 100, 128, 129, 400, 1024], size=2, replace=False):
        k = int(k)
        hits, n_hits = sc.score_topk(q_ptr, q_terms, k, query_len=qlen, topic_probs=probs)
        ref, ref_n = pyoracle.score_topk_batch(n_docs, (title[0], title[1], wt), (body[0], body[1], wb), mt, mb, q_ptr, q_terms, k,
                                               prior=None if prior is None else np.ascontiguousarray(prior.T), topic_probs=probs, query_len=qlen)
        same_hits(hits, n_hits, ref, ref_n, ("score", n_docs, n_terms, pb, k, kt))
    sc.close(); ti.close(); bi.close()
    # the same corpus in doc-range shards + ss_merge_hits
    if n_docs >= 7 and kt == 0:
        world = int(rng.choice([2, 3, 5]))
        k = int(rng.choice([1, 30, 200]))
        ref, ref_n = pyoracle.score_topk_batch(n_docs, (title[0], title[1], wt), (body[0], body[1], wb), mt, mb, q_ptr, q_terms, k, query_len=qlen)
        parts = np.zeros((world, n_q, k), dtype=engine.HIT_DTYPE); pn = np.zeros((world, n_q), dtype=np.int32)
        base = np.zeros(world, dtype=np.uint32)
        dft = np.diff(title[0].astype(np.int64)).astype(np.uint64); dfb = np.diff(body[0].astype(np.int64)).astype(np.uint64)
        for r in range(world):
            lo, hi = sharding.doc_range(n_docs, r, world)
            base[r] = lo
            st = sharding.shard_index_by_docs(*title, lo, hi); sb = sharding.shard_index_by_docs(*body, lo, hi)
            sti = engine.InvertedIndex(ctx, max(hi - lo, 1), *st); sbi = engine.InvertedIndex(ctx, max(hi - lo, 1), *sb)
            sti.set_doc_freq(dft); sbi.set_doc_freq(dfb)
            sti.tfidf_build(total, False, False, False); sbi.tfidf_build(total, False, False, False)
            ssc = engine.Scorer(ctx, sti, sbi)
            parts[r], pn[r] = ssc.score_topk(q_ptr, q_terms, k, query_len=qlen)
            ssc.close(); sti.close(); sbi.close()
        hits, n_hits = ctx.merge_hits(parts, pn, k, base)
        same_hits(hits, n_hits, ref, ref_n, ("sharded", n_docs, world, k))


def positional_table(n_docs, n_terms, n_post, seed, max_pos, anchor_frac):
    tp, pd, _ = synth.zipf_index(n_docs, n_terms, n_post, seed=seed)
    rng = np.random.default_rng(seed + 100)
    pos_ptr = [0]; pos = []
    tf = np.zeros(len(pd), dtype=np.float32)
    for i in range(len(pd)):
        c = int(rng.integers(1, 6))
        ps = sorted(rng.choice(max_pos, size=min(c, max_pos), replace=False).astype(float).tolist())
        if rng.random() < anchor_frac:
            ps.append(-100.0)
        pos += ps; pos_ptr.append(len(pos))
        tf[i] = np.float32(len(ps)) / np.float32(8)
    return (tp, pd, tf), (np.array(pos_ptr, np.uint64), np.array(pos, np.float32))


def round_phrase(ctx, rng):
    n_docs = int(rng.choice([5, 200, 3000])); n_terms = int(rng.integers(2, 40))
    (bt, bpos) = positional_table(n_docs, n_terms, int(min(n_docs * n_terms // 2 + 1, rng.integers(1, 30000))), int(rng.integers(1 << 30)),
                                  int(rng.choice([4, 12, 60])), float(rng.choice([0.0, 0.1, 0.5])))
    (tt, tpos) = positional_table(n_docs, n_terms, int(min(n_docs * n_terms // 2 + 1, rng.integers(1, 4000))), int(rng.integers(1 << 30)),
                                  int(rng.choice([3, 8])), float(rng.choice([0.0, 0.5])))
    wb, mb, _ = pyoracle.tfidf(*bt, n_docs, n_docs); wt, mt, _ = pyoracle.tfidf(*tt, n_docs, n_docs)
    title, body = (tt[0], tt[1], wt), (bt[0], bt[1], wb)
    ti = engine.InvertedIndex(ctx, n_docs, *title); bi = engine.InvertedIndex(ctx, n_docs, *body)
    ti.set_weighted(mt); bi.set_weighted(mb); ti.set_positions(*tpos); bi.set_positions(*bpos)
    sc = engine.Scorer(ctx, ti, bi)
    cases = []
    for _ in range(int(rng.integers(1, 12))):
        q = rng.integers(0, n_terms + 1, size=int(rng.integers(0, 4))).tolist()
        ph = rng.integers(0, n_terms + (1 if rng.random() < 0.1 else 0), size=int(rng.choice([0, 1, 2, 2, 3, 5]))).tolist()
        cases.append((q, ph))
    q_terms = np.array([t for q, _ in cases for t in q], dtype=np.uint32)
    q_ptr = np.concatenate([[0], np.cumsum([len(q) for q, _ in cases])]).astype(np.uint32)
    p_terms = np.array([t for _, ph in cases for t in ph], dtype=np.uint32)
    p_ptr = np.concatenate([[0], np.cumsum([len(ph) for _, ph in cases])]).astype(np.uint32)
    k = int(rng.choice([1, 20, 200]))
    hits, n_hits = sc.score_topk_phrase(q_ptr, q_terms, p_ptr, p_terms, k)
    for qi, (q, ph) in enumerate(cases):
        extra = None
        if ph:
            if all(t < n_terms for t in ph):
                extra = pyoracle.phrase(title, body, tpos, bpos, ph)
            else:
                extra = (np.zeros(0, np.uint32), np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.uint8))
        ref, _ = pyoracle.score_topk(n_docs, title, body, mt, mb, np.array(q, np.uint32), k, query_len=len(q) + len(ph), extra=extra)
        n = int(n_hits[qi])
        assert n == len(ref), ("phrase n", qi, n, len(ref), q, ph)
        assert hits["doc"][qi, :n].tolist() == ref["doc"].tolist(), ("phrase docs", qi, q, ph)
        for f in ("title", "body", "final"):
            assert np.array_equal(hits[f][qi, :n], ref[f], equal_nan=True), ("phrase", qi, f)
    sc.close(); ti.close(); bi.close()


N_AFF = [0, 0]      # two-vector rounds, topics whose iteration count moved by one


def round_pagerank(ctx, rng):
    n = int(rng.choice([1, 2, 5, 64, 1000, 30000, 200000]))
    e = int(rng.integers(0, max(1, min(n * n, 8 * n)) + 1))
    if n >= 64 and e >= n:
        ptr, dst = synth.rmat_graph(n, e, seed=int(rng.integers(1 << 30)))
    else:
        src = rng.integers(0, n, size=e); d = rng.integers(0, n, size=e)
        pairs = np.unique(np.stack([src, d], 1), axis=0) if e else np.zeros((0, 2), dtype=np.int64)
        ptr = np.zeros(n + 1, dtype=np.uint64); np.add.at(ptr, pairs[:, 0] + 1, 1); ptr = np.cumsum(ptr).astype(np.uint64)
        dst = pairs[:, 1].astype(np.uint32)
    kt = int(rng.choice([1, 2, 3, 5, 8, 16, 21]))
    n_topic = rng.integers(1, 2 * n + 2, size=kt).astype(np.int32)
    d = float(rng.choice([0.75, 0.85, 0.5]))
    eps = float(rng.choice([1e-6, 1e-9, 1e-3]))
    mi = int(rng.choice([0, 0, 1, 4]))
    g = engine.Graph(ctx, n, ptr, dst)
    pmode = int(rng.integers(0, 3)) if kt <= 2 else 0      # (round 5) K <= 2: one launch per sweep, or the sweeps inside one launch (sc1 / fences)
    ctx.set_option("pr.persistent", pmode)
    try:
        rank, iters = g.pagerank(d, eps, n_topic, max_iter=mi)
    finally:
        ctx.set_option("pr.persistent", None)
    if n >= 64 and rng.random() < 0.35:
        # (round 5) the two-vector form on in-process doc-range shards, lagged (one exchange per iteration) or not: the oracle's ranks and counts
        world = int(rng.choice([2, 3, 8]))
        lagged = int(rng.integers(0, 2))
        shards = [engine.Graph(ctx, n, ptr, dst, rank=r, world=world) for r in range(world)]
        ctx.set_option("pr.affine", 1); ctx.set_option("pr.affine_lag", lagged)
        try:
            rs, its = engine.Graph.pagerank_group(shards, d, eps, n_topic, max_iter=mi)
        finally:
            ctx.set_option("pr.affine", None); ctx.set_option("pr.affine_lag", None)
            for sg in shards: sg.close()
        refs, refs_it = pyoracle.pagerank(n, ptr, dst, d, eps, n_topic, max_iter=mi)
        assert np.max(np.abs(its.astype(np.int64) - refs_it.astype(np.int64))) <= 1, ("sharded affine iters", n, len(dst), kt, world, lagged, its.tolist(), refs_it.tolist())
        ok = its == refs_it
        if ok.any():
            np.testing.assert_allclose(rs[ok], refs[ok], rtol=1e-10, atol=0, err_msg=str(("sharded affine", n, len(dst), kt, world, lagged)))
    # the two-vector form (option pr.affine): the same ranks from two vectors; a different operation order, so an iteration count
    # may move by one where a topic's L1 change sits within rounding of eps — then its ranks are compared an iteration apart
    ctx.set_option("pr.affine", 1)
    try:
        rank2, iters2 = g.pagerank(d, eps, n_topic, max_iter=mi)
    finally:
        ctx.set_option("pr.affine", None)
    g.close()
    ref, ref_it = pyoracle.pagerank(n, ptr, dst, d, eps, n_topic, max_iter=mi)
    assert iters.tolist() == ref_it.tolist(), ("pr iters", n, len(dst), kt, d, eps, mi, iters.tolist(), ref_it.tolist())
    np.testing.assert_allclose(rank, ref, rtol=1e-11, atol=0, err_msg=str(("pr", n, len(dst), kt)))
    assert np.max(np.abs(iters2.astype(np.int64) - ref_it.astype(np.int64))) <= 1, ("affine iters", n, len(dst), kt, d, eps, mi, iters2.tolist(), ref_it.tolist())
    same = iters2 == ref_it
    if same.any():
        np.testing.assert_allclose(rank2[same], ref[same], rtol=1e-10, atol=0, err_msg=str(("affine", n, len(dst), kt, d, eps, mi)))
    N_AFF[0] += 1
    N_AFF[1] += int((~same).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    ctx = engine.Context(0)
    rng = np.random.default_rng(a.seed)
    t0 = time.time(); n = [0, 0]; last = t0
    while time.time() - t0 < a.seconds:
        u = rng.random()
        if u < 0.55:
            round_scoring(ctx, rng); n[0] += 1
        elif u < 0.75:
            round_phrase(ctx, rng); n[0] += 1
        else:
            round_pagerank(ctx, rng); n[1] += 1
        if time.time() - last > 30:
            print(f"[soak] {n[0]} scoring rounds, {n[1]} pagerank rounds, {time.time() - t0:.0f}s", flush=True); last = time.time()
    print(f"[soak] PASSED: {n[0]} scoring rounds, {n[1]} pagerank rounds (each also in the two-vector form: {N_AFF[1]} topic runs ended one "
          f"iteration apart from the oracle)", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
