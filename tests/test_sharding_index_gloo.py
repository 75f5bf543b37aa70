"""Doc-range-sharded index on CPU: N processes over torch.distributed/gloo, spaghettisearch_amd.sharding
driving CPU shard models (tests/shard_model.py) exactly the way the GPU path drives engine.Scorer:
global document frequencies by all-reduce, local scoring, one all-gather of the hits, merge.
The result must be identical to the oracle on the unsharded index."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pyoracle
from spaghettisearch_amd import sharding, synth
from tests.shard_model import CpuIndexShard, merge_hits_model

ND, NT, K_TOP = 900, 160, 25


def _corpus():
    rng = np.random.default_rng(3)
    b_ptr, b_doc, b_tf = synth.zipf_index(ND, NT, 14000, seed=44)
    t_ptr, t_doc, t_tf = synth.zipf_index(ND, NT, 1500, seed=45)
    rank = rng.random((3, ND)) * 1e-3
    q_ptr, q_terms = synth.make_queries(24, 3, NT, seed=7)
    q_terms[5] = q_terms[4]                       # duplicate term inside a query (Q8)
    q_terms[9] = 0xFFFFFFFF                       # unknown word
    probs = rng.dirichlet(np.ones(3), size=24)
    return (t_ptr, t_doc, t_tf), (b_ptr, b_doc, b_tf), rank, q_ptr, q_terms, probs


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        title, body, pr, q_ptr, q_terms, probs = _corpus()
        lo, hi = sharding.doc_range(ND, rank, world)
        lt = sharding.shard_index_by_docs(*title, lo, hi)
        lb = sharding.shard_index_by_docs(*body, lo, hi)
        df_t = sharding.global_doc_freq(lt[0])                      # the all-reduce of the build
        df_b = sharding.global_doc_freq(lb[0])
        assert df_t.tolist() == np.diff(title[0].astype(np.int64)).tolist()
        shard = CpuIndexShard(hi - lo, lt, lb, df_t, df_b, ND + 50)
        shard.set_prior(pr[:, lo:hi])
        sc = sharding.DocShardedScorer(shard, merge_hits_model, ND, rank, world)
        hits, n_hits = sc.score_topk(q_ptr, q_terms, K_TOP, topic_probs=probs)
        got = [None] * world
        dist.all_gather_object(got, (hits.tobytes(), n_hits.tobytes()))
        assert all(g == got[0] for g in got)                        # every rank holds the same merged result
        if rank == 0:
            np.savez(out_path, hits=hits, n_hits=n_hits)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_doc_sharded_index_gloo_matches_unsharded_oracle(tmp_path, world):
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    title, body, pr, q_ptr, q_terms, probs = _corpus()
    wt, mt, _ = pyoracle.tfidf(*title, ND + 50, ND)
    wb, mb, _ = pyoracle.tfidf(*body, ND + 50, ND)
    ref, ref_n = pyoracle.score_topk_batch(ND, (title[0], title[1], wt), (body[0], body[1], wb), mt, mb, q_ptr, q_terms, K_TOP,
                                           prior=np.ascontiguousarray(pr.T), topic_probs=probs)
    assert got["n_hits"].tolist() == ref_n.tolist()
    for f in ("doc", "title", "body", "pagerank", "final"):
        assert np.array_equal(got["hits"][f], ref[f]), f


def test_shard_index_by_docs_numpy_and_torch_agree():
    import torch
    title, body, *_ = _corpus()
    pos_ptr = np.arange(len(body[1]) + 1, dtype=np.uint64) * 2
    pos = np.arange(2 * len(body[1]), dtype=np.float32)
    for lo, hi in ((0, 300), (300, 301), (301, 900), (500, 500)):
        a = sharding.shard_index_by_docs(*body, lo, hi, pos_ptr, pos)
        tb = (torch.from_numpy(body[0].view(np.int64)), torch.from_numpy(body[1].view(np.int32)), torch.from_numpy(body[2]),
              torch.from_numpy(pos_ptr.view(np.int64)), torch.from_numpy(pos))
        b = sharding.shard_index_by_docs(tb[0], tb[1], tb[2], lo, hi, tb[3], tb[4])
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x).astype(np.float64), y.numpy().astype(np.float64))
        assert int(a[0][-1]) == len(a[1]) and (a[1] < max(hi - lo, 1)).all()
        # per term: ascending local docs
        for t in range(NT):
            seg = a[1][int(a[0][t]):int(a[0][t + 1])].astype(np.int64)
            assert (np.diff(seg) > 0).all()


def test_merge_model_order():
    dt = sharding._hit_dtype()
    parts = np.zeros((2, 1, 4), dtype=dt)
    parts["final"][0, 0] = [9.0, 5.0, 5.0, np.nan]
    parts["doc"][0, 0] = [3, 1, 2, 0]
    parts["final"][1, 0, :3] = [9.0, 7.0, np.nan]
    parts["doc"][1, 0, :3] = [0, 5, 1]
    n = np.array([[4], [3]], dtype=np.int32)
    out, n_out = merge_hits_model(parts, n, 6, np.array([0, 10], dtype=np.uint32))
    assert n_out.tolist() == [6]
    assert out["doc"][0].tolist() == [3, 10, 15, 1, 2, 0]           # 9@3, 9@10, 7@15, 5@1, 5@2, NaN@0 (NaN@11 cut)
