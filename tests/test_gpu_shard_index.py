"""GPU parity of the doc-range-sharded index: ss_index_set_doc_freq + ss_tfidf_build
per shard, ss_score_topk per shard, ss_merge_hits — all shards in one process on one GPU, compared with the
oracle on the UNSHARDED index (term_weighting.go:29-50, main_retrieve.go:50-103, util.go:48-54).  Bit-exact."""
import numpy as np
import pytest

from spaghettisearch_amd import sharding, synth
from tests.shard_model import merge_hits_model
from tests.test_gpu_score import assert_same_hits

pytestmark = pytest.mark.gpu

ND, NT, TOTAL = 6000, 700, 6400


def corpus():
    rng = np.random.default_rng(11)
    body = synth.zipf_index(ND, NT, 160_000, seed=44)
    title = synth.zipf_index(ND, NT, 12_000, seed=45)
    rank = rng.random((4, ND)) * 1e-3
    q_ptr, q_terms = synth.make_queries(96, 3, 300, seed=7)
    q_terms[7] = q_terms[6]
    q_terms[12] = 0xFFFFFFFF
    probs = rng.dirichlet(np.ones(4), size=96)
    return title, body, rank, q_ptr, q_terms, probs


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_build_and_topk_match_unsharded_oracle(ss_ctx, oracle, world):
    from spaghettisearch_amd import engine
    title, body, rank, q_ptr, q_terms, probs = corpus()
    k = 40
    wt, mt, idf_t = oracle.tfidf(*title, TOTAL, ND)
    wb, mb, idf_b = oracle.tfidf(*body, TOTAL, ND)
    ref, ref_n = oracle.score_topk_batch(ND, (title[0], title[1], wt), (body[0], body[1], wb), mt, mb, q_ptr, q_terms, k,
                                         prior=np.ascontiguousarray(rank.T), topic_probs=probs)
    shards = []
    for r in range(world):
        lo, hi = sharding.doc_range(ND, r, world)
        shards.append((lo, hi, sharding.shard_index_by_docs(*title, lo, hi), sharding.shard_index_by_docs(*body, lo, hi)))
    df_t = sum(sharding.local_doc_freq(s[2][0]) for s in shards).astype(np.uint64)      # what the all-reduce computes
    df_b = sum(sharding.local_doc_freq(s[3][0]) for s in shards).astype(np.uint64)
    parts = np.zeros((world, len(q_ptr) - 1, k), dtype=engine.HIT_DTYPE)
    n_hits = np.zeros((world, len(q_ptr) - 1), dtype=np.int32)
    for r, (lo, hi, lt, lb) in enumerate(shards):
        ti = engine.InvertedIndex(ss_ctx, hi - lo, *lt)
        bi = engine.InvertedIndex(ss_ctx, hi - lo, *lb)
        ti.set_doc_freq(df_t)
        bi.set_doc_freq(df_b)
        w_t, m_t, i_t = ti.tfidf_build(TOTAL)
        w_b, m_b, i_b = bi.tfidf_build(TOTAL)
        # build parity: idf of the whole list, weights of the slice, magnitudes of the doc range
        assert np.array_equal(i_b.view(np.uint32), idf_b.view(np.uint32))
        sel = (body[1] >= lo) & (body[1] < hi)
        assert np.array_equal(w_b.view(np.uint32), wb[sel].view(np.uint32))
        assert np.array_equal(m_b, mb[lo:hi]) and np.array_equal(m_t, mt[lo:hi])
        sc = engine.Scorer(ss_ctx, ti, bi)
        sc.set_prior(np.ascontiguousarray(rank[:, lo:hi]))
        parts[r], n_hits[r] = sc.score_topk(q_ptr, q_terms, k, topic_probs=probs)
        sc.close()
        ti.close()
        bi.close()
    base = np.array([s[0] for s in shards], dtype=np.uint32)
    hits, n = ss_ctx.merge_hits(parts, n_hits, k, base)
    assert_same_hits(hits, n, ref, ref_n)
    # rows past n_hits are zero-filled like ss_score_topk's
    for q in range(len(n)):
        assert not hits[q, n[q]:].view(np.uint8).any()


def test_merge_hits_order_ties_nan_and_device_buffers(ss_ctx):
    import torch
    from spaghettisearch_amd import engine
    rng = np.random.default_rng(5)
    for n_parts, n_q, k in ((1, 3, 5), (2, 17, 1), (3, 40, 64), (8, 9, 1024)):
        parts = np.zeros((n_parts, n_q, k), dtype=engine.HIT_DTYPE)
        n_hits = rng.integers(0, k + 1, size=(n_parts, n_q)).astype(np.int32)
        n_hits[0, 0] = 0
        base = (np.arange(n_parts) * 5000).astype(np.uint32)
        for p in range(n_parts):
            for q in range(n_q):
                m = int(n_hits[p, q])
                fin = rng.integers(0, 12, size=m).astype(np.float64) / 4.0          # many equal finals across shards
                fin[rng.random(m) < 0.05] = np.nan
                fin[rng.random(m) < 0.03] = np.inf
                doc = rng.choice(5000, size=m, replace=False).astype(np.uint32)
                nan = np.isnan(fin)
                order = np.lexsort((doc, -np.where(nan, 0.0, fin), nan))             # each shard's list in result order
                parts["final"][p, q, :m] = fin[order]
                parts["doc"][p, q, :m] = doc[order]
                parts["title"][p, q, :m] = rng.random(m)
                parts["body"][p, q, :m] = rng.random(m)
                parts["pagerank"][p, q, :m] = rng.random(m)
        ref, ref_n = merge_hits_model(parts, n_hits, k, base)
        hits, n = ss_ctx.merge_hits(parts, n_hits, k, base)
        assert n.tolist() == ref_n.tolist()
        assert hits.tobytes() == ref.tobytes()
        # device in, device out
        d_parts = torch.from_numpy(parts.view(np.uint8).reshape(-1)).cuda()
        d_n = torch.from_numpy(n_hits).cuda()
        d_out = torch.zeros(n_q * k * 40, dtype=torch.uint8, device="cuda")
        d_nout = torch.zeros(n_q, dtype=torch.int32, device="cuda")
        ss_ctx.merge_hits(d_parts, d_n, k, base, out=(d_out, d_nout))
        assert d_out.cpu().numpy().tobytes() == ref.tobytes() and d_nout.cpu().numpy().tolist() == ref_n.tolist()
    # no doc_base: ids already global
    hits2, _ = ss_ctx.merge_hits(parts[:1], n_hits[:1], k, None)
    assert np.array_equal(hits2["doc"][1, :n_hits[0, 1]], parts["doc"][0, 1, :n_hits[0, 1]])


def test_shard_api_errors(ss_ctx):
    from spaghettisearch_amd import engine
    from spaghettisearch_amd._lib import SpaghettiError as SsError
    title, body, *_ = corpus()
    lo, hi = 0, ND // 2
    lb = sharding.shard_index_by_docs(*body, lo, hi)
    bi = engine.InvertedIndex(ss_ctx, hi - lo, *lb)
    df = sharding.local_doc_freq(lb[0]).astype(np.uint64)
    bad = df.copy()
    bad[3] -= 1                                             # smaller than the local list: impossible
    with pytest.raises(SsError):
        bi.set_doc_freq(bad)
    bi.set_doc_freq(df)
    bi.set_doc_freq(None)
    bi.tfidf_build(TOTAL, want_w=False, want_mag=False, want_idf=False)
    with pytest.raises(SsError):
        bi.set_doc_freq(df)                                 # already weighted
    bi.close()
    parts = np.zeros((2, 1, 4), dtype=engine.HIT_DTYPE)
    with pytest.raises(SsError):
        ss_ctx.merge_hits(parts, np.array([[5], [0]], dtype=np.int32), 4)
