"""GPU parity: HIP TF-IDF weight / magnitude build vs the CPU oracle.

Reference: ranking/term_weighting.go:10-57 (+ sqrt at :72).  idf and the float32
weights must be BIT-EXACT (same IEEE operation sequence as the oracle, built with
-ffp-contract=off); magnitudes are float64 sums in a different order: 1e-12.
"""
import math

import numpy as np
import pytest

from spaghettisearch_amd import synth

pytestmark = pytest.mark.gpu


def build_both(ss_ctx, oracle, n_docs, term_ptr, post_doc, tf, total_docs):
    from spaghettisearch_amd import engine
    idx = engine.InvertedIndex(ss_ctx, n_docs, term_ptr, post_doc, tf)
    try:
        w, mag, idf = idx.tfidf_build(total_docs)
    finally:
        idx.close()
    w_ref, mag_ref, idf_ref = oracle.tfidf(term_ptr, post_doc, tf, total_docs, n_docs)
    return (w, mag, idf), (w_ref, mag_ref, idf_ref)


def test_kat(ss_ctx, oracle):
    # the hand-worked table of tests/test_oracle_kat.py::test_tfidf_by_hand (N=8 != 4 indexed docs, Q7)
    term_ptr = np.array([0, 2, 6, 9, 9], dtype=np.uint64)
    post_doc = np.array([0, 1, 0, 1, 2, 3, 1, 2, 3], dtype=np.uint32)
    tf = np.array([.5, 1, .25, .25, .25, .25, 1, .5, .125], dtype=np.float32)
    (w, mag, idf), (w_ref, mag_ref, idf_ref) = build_both(ss_ctx, oracle, 4, term_ptr, post_doc, tf, 8)
    assert idf[0] == 2.0 and idf[1] == 1.0 and idf[2] == np.float32(math.log2(8 / 3)) and np.isinf(idf[3])
    assert np.array_equal(w, w_ref)
    np.testing.assert_allclose(mag, mag_ref, rtol=1e-15)


@pytest.mark.parametrize("n_docs,n_terms,n_post,total", [(500, 200, 5000, 700), (20000, 5000, 300000, 20000),
                                                          (100000, 30000, 2000000, 131072)])
def test_zipf_index(ss_ctx, oracle, n_docs, n_terms, n_post, total):
    tp, pd, tf = synth.zipf_index(n_docs, n_terms, n_post, seed=n_terms)
    (w, mag, idf), (w_ref, mag_ref, idf_ref) = build_both(ss_ctx, oracle, n_docs, tp, pd, tf, total)
    live = np.diff(tp.astype(np.int64)) > 0
    assert np.array_equal(idf[live], idf_ref[live])          # bit-exact float32 idf
    assert np.array_equal(w, w_ref)                          # bit-exact float32 weights
    np.testing.assert_allclose(mag, mag_ref, rtol=1e-12)


def test_not_idempotent_like_the_reference(ss_ctx, oracle):
    # UpdateTermWeights multiplies the stored weight in place (term_weighting.go:42)
    from spaghettisearch_amd import engine
    tp, pd, tf = synth.zipf_index(300, 50, 1500, seed=2)
    idx = engine.InvertedIndex(ss_ctx, 300, tp, pd, tf)
    w1, _, idf = idx.tfidf_build(300)
    w2, mag2, _ = idx.tfidf_build(300)
    idx.close()
    per_post_idf = np.repeat(idf, np.diff(tp.astype(np.int64)))
    assert np.array_equal(w2, (w1 * per_post_idf).astype(np.float32))


def test_rejects_unsorted_and_out_of_range(ss_ctx):
    from spaghettisearch_amd import SpaghettiError, engine
    tp = np.array([0, 3, 5], dtype=np.uint64)
    ok_docs = np.array([0, 2, 4, 1, 3], dtype=np.uint32)    # descent only at the term boundary: fine
    engine.InvertedIndex(ss_ctx, 5, tp, ok_docs, np.ones(5, np.float32)).close()
    with pytest.raises(SpaghettiError) as ei:
        engine.InvertedIndex(ss_ctx, 5, tp, np.array([0, 4, 2, 1, 3], dtype=np.uint32), np.ones(5, np.float32))
    assert ei.value.code == 5
    with pytest.raises(SpaghettiError) as ei:
        engine.InvertedIndex(ss_ctx, 5, tp, np.array([0, 2, 2, 1, 3], dtype=np.uint32), np.ones(5, np.float32))
    assert ei.value.code == 5                                 # duplicate doc inside a term
    with pytest.raises(SpaghettiError):
        engine.InvertedIndex(ss_ctx, 5, tp, np.array([0, 2, 9, 1, 3], dtype=np.uint32), np.ones(5, np.float32))


@pytest.mark.parametrize("n_docs,n_terms,n_post,total", [(1, 1, 1, 1), (9000, 300, 120000, 9000), (70000, 5000, 900000, 80000),
                                                          (20000, 40, 400000, 20000)])
def test_bucketed_magnitude_pass(ss_ctx, oracle, n_docs, n_terms, n_post, total):
    # large tables sum the squares per doc-range bucket in LDS instead of one global float64 atomic per posting;
    # option "tfidf.bucket_min" = 1 sends these small tables down that path (several buckets, a partial last bucket, one doc)
    from spaghettisearch_amd import engine
    tp, pd, tf = synth.zipf_index(n_docs, n_terms, min(n_post, n_docs * n_terms // 2 + 1), seed=n_docs)
    w_ref, mag_ref, idf_ref = oracle.tfidf(tp, pd, tf, total, n_docs)
    for knob in (1, 1 << 62):                                         # bucketed, then the atomic pass: same bits
        ix = engine.InvertedIndex(ss_ctx, n_docs, tp, pd, tf)
        with ss_ctx.options(tfidf__bucket_min=knob):
            w, mag, idf = ix.tfidf_build(total)
        ix.close()
        assert np.array_equal(w.view(np.uint32), w_ref.view(np.uint32))
        assert np.array_equal(idf.view(np.uint32), idf_ref.view(np.uint32))
        assert np.array_equal(mag, mag_ref)


@pytest.mark.parametrize("n_docs,n_terms,n_post", [(200_000, 60, 3_000_000), (100_000, 2000, 2_500_000), (80_000, 3, 240_000)])
def test_head_lists_are_summed_in_place(ss_ctx, oracle, n_docs, n_terms, n_post):
    """Long posting lists skip the partition: the bucket kernel reads bucket b's run of every head list straight from the table
    and weights it on the way.  Which lists are head (option "tfidf.head_min_run": 0 = none, 1 = every list longer than two
    chunks) only moves postings between two exact paths: weights, idf and magnitudes keep their bits — also for the
    magnitudes-only pass (ss_index_refresh_magnitudes)."""
    from spaghettisearch_amd import engine
    tp, pd, tf = synth.zipf_index(n_docs, n_terms, n_post, seed=n_terms)
    assert int(np.diff(tp.astype(np.int64)).max()) > 16385            # at least one list takes the head path
    w_ref, mag_ref, idf_ref = oracle.tfidf(tp, pd, tf, n_docs, n_docs)
    for min_run in (0, 1, 32):
        ix = engine.InvertedIndex(ss_ctx, n_docs, tp, pd, tf)
        with ss_ctx.options(tfidf__bucket_min=1, tfidf__head_min_run=min_run):
            w, mag, idf = ix.tfidf_build(n_docs)
            mag2 = ix.refresh_magnitudes()
        ix.close()
        assert np.array_equal(w.view(np.uint32), w_ref.view(np.uint32)), min_run
        assert np.array_equal(idf.view(np.uint32), idf_ref.view(np.uint32))
        assert np.array_equal(mag, mag_ref), min_run
        assert np.array_equal(mag2, mag_ref), min_run
