#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/.

The reference (Go) ships no fixtures for this path and cannot be built here
(SURVEY.md §8c), so these vectors come from this repo's OWN restatements: they are
emitted by oracle/oracle.c and only written when the independent numpy restatement
(oracle/oracle_np.py) agrees.  PARITY UNPINNED by the reference; the vectors pin the
oracle against drift and give the GPU tests fixed inputs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_np as onp          # noqa: E402
from oracle import pyoracle as po            # noqa: E402
from spaghettisearch_amd import synth        # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    po.build()
    # ---- PageRank: 600-node R-MAT, 3 topics, to convergence and fixed 2 iterations
    n, e = 600, 2600
    ptr, dst = synth.rmat_graph(n, e, seed=1234)
    n_topic = np.array([600, 211, 5], dtype=np.int32)
    rank, iters = po.pagerank(n, ptr, dst, 0.75, 1e-9, n_topic)
    rank2, iters2 = onp.pagerank(n, ptr, dst, 0.75, 1e-9, n_topic)
    assert iters.tolist() == iters2.tolist()
    np.testing.assert_allclose(rank, rank2, rtol=1e-12)
    r_fix, it_fix = po.pagerank(n, ptr, dst, 0.85, 0.0, n_topic, max_iter=2)
    np.savez_compressed(os.path.join(HERE, "pagerank_rmat600.npz"), n=n, out_ptr=ptr, out_dst=dst, n_topic=n_topic,
                        d=0.75, eps=1e-9, rank=rank, iters=iters, d_fix=0.85, rank_fix2=r_fix)
    # ---- TF-IDF + scoring: 300 docs, 80 terms
    nd, nt = 300, 80
    b = synth.zipf_index(nd, nt, 2500, seed=77)
    t = synth.zipf_index(nd, nt, 400, seed=78)
    total_docs = 333                       # N = #PageRank nodes != #indexed docs (Q7)
    wb, mb, idfb = po.tfidf(*b, total_docs, nd)
    wt, mt, idft = po.tfidf(*t, total_docs, nd)
    wb2, mb2, _ = onp.tfidf(*b, total_docs, nd)
    assert np.array_equal(wb, wb2)
    np.testing.assert_allclose(mb, mb2, rtol=1e-13)
    rng = np.random.default_rng(5)
    lens = rng.integers(1, 5, size=24)
    q_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    q_terms = rng.integers(0, nt + 3, size=int(lens.sum())).astype(np.uint32)   # a few unknown ids
    K = 4
    prior = rng.random((nd, K))
    probs = rng.dirichlet(np.ones(K), size=24)
    k = 10
    hits, n_hits = po.score_topk_batch(nd, (t[0], t[1], wt), (b[0], b[1], wb), mt, mb, q_ptr, q_terms, k)
    hits_p, n_hits_p = po.score_topk_batch(nd, (t[0], t[1], wt), (b[0], b[1], wb), mt, mb, q_ptr, q_terms, k,
                                           prior=prior, topic_probs=probs)
    for q in range(24):
        d2, T2, B2, S2, F2 = onp.score_topk(nd, (t[0], t[1], wt), (b[0], b[1], wb), mt, mb,
                                            q_terms[q_ptr[q]:q_ptr[q + 1]], k, prior=prior, topic_probs=probs[q])
        assert d2.tolist() == hits_p["doc"][q, :n_hits_p[q]].tolist()
        assert np.array_equal(F2, hits_p["final"][q, :n_hits_p[q]])
    np.savez_compressed(os.path.join(HERE, "index_300x80.npz"), n_docs=nd, total_docs=total_docs,
                        b_ptr=b[0], b_doc=b[1], b_tf=b[2], t_ptr=t[0], t_doc=t[1], t_tf=t[2],
                        b_w=wb, t_w=wt, b_mag=mb, t_mag=mt, b_idf=idfb, t_idf=idft,
                        q_ptr=q_ptr, q_terms=q_terms, k=k, prior=prior, probs=probs,
                        hits=hits, n_hits=n_hits, hits_prior=hits_p, n_hits_prior=n_hits_p)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
