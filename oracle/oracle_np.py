"""Second, independently written restatement of the reference arithmetic (numpy/scipy).

TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).  It exists to cross-check
oracle.c — two restatements written in different styles (scalar push loops in C,
vectorised pull/bincount here) agreeing is the substitute for the Go reference,
which cannot be built in this image.  PARITY UNPINNED.

Reference lines followed are cited per function.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def pagerank_topic(n_nodes, out_ptr, out_dst, d, eps, n_init, max_iter=0):
    """ranking/pagerank.go:85-145, written as a pull SpMV (A^T @ contrib)."""
    out_ptr = np.asarray(out_ptr, dtype=np.int64)
    out_dst = np.asarray(out_dst, dtype=np.int64)
    N = int(n_nodes)
    outdeg = np.diff(out_ptr)
    src = np.repeat(np.arange(N, dtype=np.int64), outdeg)
    # in-edge matrix M[c, p] = multiplicity of edge p->c (duplicates count, Q5)
    M = sp.csr_matrix((np.ones(len(out_dst)), (out_dst, src)), shape=(N, N))
    nd = outdeg > 0
    teleport = 1.0 - d                                  # :90
    last = np.full(N, 1.0 / float(n_init))              # :104-105
    it = 0
    while True:
        it += 1
        contrib = np.zeros(N)
        contrib[nd] = d * last[nd] / outdeg[nd]         # :136
        y = M @ contrib                                 # :140-142
        if it == 1:
            y = y + 1.0 / float(n_init)                 # Q4: iteration 1 accumulates onto 1/n (:97-107)
        total = contrib.sum() + teleport * N            # :137, :112
        cur = (y + teleport) / total                    # :117
        change = np.abs(cur - last).sum()               # :118
        last = cur
        if not (change > eps):                          # :93
            break
        if max_iter and it >= max_iter:
            break
    return last, it


def pagerank(n_nodes, out_ptr, out_dst, d, eps, n_topic, max_iter=0):
    """pagerank.go:54-63"""
    ranks, iters = [], []
    for n in np.atleast_1d(n_topic):
        r, i = pagerank_topic(n_nodes, out_ptr, out_dst, d, eps, int(n), max_iter)
        ranks.append(r)
        iters.append(i)
    return np.stack(ranks), np.asarray(iters, dtype=np.int32)


def tfidf(term_ptr, post_doc, post_tf, total_docs, n_docs):
    """ranking/term_weighting.go:29-50 + :72 (sqrt).  np.log2 stands in for Go's
    math.Log2 (may differ in the last float64 bit before narrowing to float32)."""
    term_ptr = np.asarray(term_ptr, dtype=np.int64)
    df = np.diff(term_ptr).astype(np.float64)
    with np.errstate(divide="ignore"):
        idf = np.log2(np.float64(total_docs) / df).astype(np.float32)       # :37
    per_post_idf = np.repeat(idf, np.diff(term_ptr))
    w = (np.asarray(post_tf, dtype=np.float32) * per_post_idf).astype(np.float32)   # :42
    sq = (w * w).astype(np.float32).astype(np.float64)                      # :44
    mag2 = np.bincount(np.asarray(post_doc, dtype=np.int64), weights=sq, minlength=n_docs)
    return w, np.sqrt(mag2), idf


def score_topk(n_docs, title, body, mag_title, mag_body, q_terms, k, query_len=None,
               prior=None, topic_probs=None):
    """main_retrieve.go:50-103 + get_metadata.go:31-69 + util.go:48-54, one query."""
    t_ptr, t_doc, t_w = title
    b_ptr, b_doc, b_w = body
    n_terms = len(b_ptr) - 1
    accT = np.zeros(n_docs)
    accB = np.zeros(n_docs)
    hit = np.zeros(n_docs, dtype=bool)
    for t in q_terms:                                   # duplicates counted twice (Q8)
        t = int(t)
        if t >= n_terms:
            continue
        s, e = int(b_ptr[t]), int(b_ptr[t + 1])
        np.add.at(accB, b_doc[s:e], b_w[s:e].astype(np.float64))
        hit[b_doc[s:e]] = True
        s, e = int(t_ptr[t]), int(t_ptr[t + 1])
        np.add.at(accT, t_doc[s:e], t_w[s:e].astype(np.float64))
        hit[t_doc[s:e]] = True
    docs = np.nonzero(hit)[0]
    if query_len is None:
        query_len = len(q_terms)
    qmag = np.sqrt(np.float64(query_len))               # get_metadata.go:53
    with np.errstate(divide="ignore", invalid="ignore"):
        B = accB[docs] / (np.asarray(mag_body)[docs] * qmag)    # :57
        T = accT[docs] / (np.asarray(mag_title)[docs] * qmag)   # :58
    B[np.isnan(B)] = 0.0                                # :61-66
    T[np.isnan(T)] = 0.0
    if prior is not None and topic_probs is not None:
        sqd = np.zeros(len(docs))
        for t in range(prior.shape[1]):                 # :40-42, topic order
            sqd = sqd + topic_probs[t] * prior[docs, t]
    else:
        sqd = np.zeros(len(docs))
    with np.errstate(invalid="ignore"):
        final = (0.33 * sqd + 0.38 * T + 0.29 * B) * 100.0  # :69
    # descending final, ties ascending doc, NaN last
    key = np.where(np.isnan(final), -np.inf, final)
    nanflag = np.isnan(final)
    order = np.lexsort((docs, -key, nanflag))
    order = order[:k]
    return docs[order].astype(np.uint32), T[order], B[order], sqd[order], final[order]
