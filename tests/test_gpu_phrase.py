"""GPU parity: quoted-phrase search merged into the scorer (ss_score_topk_phrase) vs the CPU oracle.

Reference: retrieval/phrase.go:11-170 (getPhraseFromInverted, evalPhraseOccurrence, getPosTerm),
util.go:162-203 (intersect), merged at main_retrieve.go:73-78; query length = query tokens + phrase
tokens (main_retrieve.go:90).  Positions are float32 as in the reference (-100 = anchor/meta text);
the intersection is exact float32 equality, the doc ids must match bit for bit and so must the scores.
"""
import numpy as np
import pytest

from spaghettisearch_amd import synth

pytestmark = pytest.mark.gpu


def positional_table(n_docs, n_terms, n_post, seed, max_pos=60, anchor_frac=0.1):
    """Index with positions: a doc's occurrences of a term are `c` distinct positions in [0, max_pos)
    (+ sometimes a -100 anchor entry appended, unsorted, as parser.getWordInfo does)."""
    tp, pd, _ = synth.zipf_index(n_docs, n_terms, n_post, seed=seed)
    rng = np.random.default_rng(seed + 100)
    pos_ptr = [0]
    pos = []
    tf = np.zeros(len(pd), dtype=np.float32)
    for i in range(len(pd)):
        c = int(rng.integers(1, 6))
        ps = sorted(rng.choice(max_pos, size=c, replace=False).astype(float).tolist())
        if rng.random() < anchor_frac:
            ps.append(-100.0)
        pos += ps
        pos_ptr.append(len(pos))
        tf[i] = np.float32(len(ps)) / np.float32(8)
    return (tp, pd, tf), (np.array(pos_ptr, np.uint64), np.array(pos, np.float32))


def test_phrase_matches_oracle(ss_ctx, oracle):
    from spaghettisearch_amd import engine
    n_docs, n_terms = 3000, 40
    (bt, bpos) = positional_table(n_docs, n_terms, 30000, seed=5)
    (tt, tpos) = positional_table(n_docs, n_terms, 4000, seed=6, max_pos=8, anchor_frac=0.5)
    wb, mb, _ = oracle.tfidf(*bt, n_docs, n_docs)
    wt, mt, _ = oracle.tfidf(*tt, n_docs, n_docs)
    title, body = (tt[0], tt[1], wt), (bt[0], bt[1], wb)
    ti = engine.InvertedIndex(ss_ctx, n_docs, *title)
    bi = engine.InvertedIndex(ss_ctx, n_docs, *body)
    ti.set_weighted(mt)
    bi.set_weighted(mb)
    sc = engine.Scorer(ss_ctx, ti, bi)
    q_ptr0 = np.array([0, 1], dtype=np.uint32)
    with pytest.raises(Exception):                       # positions not loaded yet
        sc.score_topk_phrase(q_ptr0, np.array([0], np.uint32), q_ptr0, np.array([1], np.uint32), 5)
    ti.set_positions(*tpos)
    bi.set_positions(*bpos)
    # (terms, phrase): plain, phrase only, both, duplicate phrase word, unknown phrase word, 3-word phrase
    cases = [([0, 3], [1, 2]), ([], [0, 1]), ([5], [2, 0]), ([2, 2], [1, 1]), ([4], [0, 99]), ([7, 1], [0, 1, 2]),
             ([9], []), ([], [3]), ([1], [3, 2, 1, 0])]
    q_terms = np.array([t for q, _ in cases for t in q], dtype=np.uint32)
    q_ptr = np.concatenate([[0], np.cumsum([len(q) for q, _ in cases])]).astype(np.uint32)
    p_terms = np.array([t for _, ph in cases for t in ph], dtype=np.uint32)
    p_ptr = np.concatenate([[0], np.cumsum([len(ph) for _, ph in cases])]).astype(np.uint32)
    n_phrase_docs = 0
    for k in (20, 200):
        hits, n_hits = sc.score_topk_phrase(q_ptr, q_terms, p_ptr, p_terms, k)
        for qi, (q, ph) in enumerate(cases):
            extra = None
            if ph:
                if all(t < n_terms for t in ph):
                    extra = oracle.phrase(title, body, tpos, bpos, ph)
                    n_phrase_docs += len(extra[0])
                else:
                    extra = (np.zeros(0, np.uint32), np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.uint8))
            ref, _ = oracle.score_topk(n_docs, title, body, mt, mb, np.array(q, np.uint32), k,
                                       query_len=len(q) + len(ph), extra=extra)
            n = int(n_hits[qi])
            assert n == len(ref), (qi, n, len(ref))
            assert hits["doc"][qi, :n].tolist() == ref["doc"].tolist(), qi
            for f in ("title", "body", "final"):
                assert np.array_equal(hits[f][qi, :n], ref[f]), (qi, f)
    assert n_phrase_docs > 20          # the cases do exercise real phrase matches
    # the same batch in flight (ss_score_topk_submit with phrases) and collected: the synchronous call's hits.  Three batches at
    # once: their phrase-match and scoring kernels overlap on internal streams (per-turn phrase result lists), the merges follow
    # in order on the context's stream
    hs, ns = sc.score_topk_phrase(q_ptr, q_terms, p_ptr, p_terms, 20)
    for _ in range(3):
        t1 = sc.submit(q_ptr, q_terms, 200, p_ptr=p_ptr, p_terms=p_terms)
        t2 = sc.submit(q_ptr, q_terms, 20, p_ptr=p_ptr, p_terms=p_terms)
        t3 = sc.submit(q_ptr, q_terms, 200, p_ptr=p_ptr, p_terms=p_terms)
        hc2, nc2 = sc.collect(t2)
        hc3, nc3 = sc.collect(t3)
        hc1, nc1 = sc.collect(t1)
        assert np.array_equal(hc1, hits) and np.array_equal(nc1, n_hits)
        assert np.array_equal(hc3, hits) and np.array_equal(nc3, n_hits)
        assert np.array_equal(hc2, hs) and np.array_equal(nc2, ns)
    # the plain entry point is the phrase entry point without phrases
    h1, n1 = sc.score_topk(q_ptr, q_terms, 20)
    h2, n2 = sc.score_topk_phrase(q_ptr, q_terms, np.zeros(len(cases) + 1, np.uint32), np.zeros(0, np.uint32), 20)
    assert n1.tolist() == n2.tolist() and h1.tobytes() == h2.tobytes()
    sc.close()
    ti.close()
    bi.close()


def positional_table_fast(n_docs, n_terms, n_post, seed, max_pos=12):
    """vectorised variant for larger tables: c consecutive positions from a random start, sometimes a -100 anchor entry"""
    tp, pd, _ = synth.zipf_index(n_docs, n_terms, n_post, seed=seed, q=1000.0, clip_frac=0.9)
    rng = np.random.default_rng(seed + 100)
    n = len(pd)
    c = rng.integers(1, 4, size=n)
    anchor = rng.random(n) < 0.1
    cnt = c + anchor
    pos_ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
    start = rng.integers(0, max_pos, size=n).astype(np.float32)
    within = (np.arange(int(pos_ptr[-1])) - np.repeat(pos_ptr[:-1].astype(np.int64), cnt)).astype(np.float32)
    pos = np.repeat(start, cnt) + within
    last = np.repeat(c, cnt) == within.astype(np.int64)            # the extra slot of anchored postings
    pos[last] = np.float32(-100.0)
    tf = (cnt / 8.0).astype(np.float32)
    return (tp, pd, tf), (pos_ptr, pos.astype(np.float32))


def test_phrase_candidates_over_several_workgroups(ss_ctx, oracle):
    # the rarest phrase term has more postings than one k_phrase_match workgroup takes (8192): the parts' match runs are
    # closed up per query by k_phrase_close; bit-exact against the oracle, title pass included
    from spaghettisearch_amd import engine
    n_docs, n_terms = 60000, 10
    (bt, bpos) = positional_table_fast(n_docs, n_terms, 330000, seed=15)
    (tt, tpos) = positional_table_fast(n_docs, n_terms, 150000, seed=16, max_pos=4)
    assert np.diff(bt[0].astype(np.int64)).min() > 2 * 8192 and np.diff(tt[0].astype(np.int64)).min() > 8192
    wb, mb, _ = oracle.tfidf(*bt, n_docs, n_docs)
    wt, mt, _ = oracle.tfidf(*tt, n_docs, n_docs)
    title, body = (tt[0], tt[1], wt), (bt[0], bt[1], wb)
    ti = engine.InvertedIndex(ss_ctx, n_docs, *title)
    bi = engine.InvertedIndex(ss_ctx, n_docs, *body)
    ti.set_weighted(mt)
    bi.set_weighted(mb)
    ti.set_positions(*tpos)
    bi.set_positions(*bpos)
    sc = engine.Scorer(ss_ctx, ti, bi)
    try:
        cases = [([], [0, 1]), ([2], [3, 4]), ([], [5, 6, 7]), ([1], [9, 8]), ([], [4])]
        q_terms = np.array([t for q, _ in cases for t in q], dtype=np.uint32)
        q_ptr = np.concatenate([[0], np.cumsum([len(q) for q, _ in cases])]).astype(np.uint32)
        p_terms = np.array([t for _, ph in cases for t in ph], dtype=np.uint32)
        p_ptr = np.concatenate([[0], np.cumsum([len(ph) for _, ph in cases])]).astype(np.uint32)
        hits, n_hits = sc.score_topk_phrase(q_ptr, q_terms, p_ptr, p_terms, 300)
        n_phrase_docs = 0
        for qi, (q, ph) in enumerate(cases):
            extra = oracle.phrase(title, body, tpos, bpos, ph)
            n_phrase_docs += len(extra[0])
            ref, _ = oracle.score_topk(n_docs, title, body, mt, mb, np.array(q, np.uint32), 300, query_len=len(q) + len(ph), extra=extra)
            n = int(n_hits[qi])
            assert n == len(ref), (qi, n, len(ref))
            assert hits["doc"][qi, :n].tolist() == ref["doc"].tolist(), qi
            for f in ("title", "body", "final"):
                assert np.array_equal(hits[f][qi, :n], ref[f]), (qi, f)
        assert n_phrase_docs > 1000
    finally:
        sc.close()
        ti.close()
        bi.close()
