"""BASELINE configs 4 and 5 are 8 ranks: everything the doc-range-sharded PageRank does at world 8 that a one-GPU box can run.
RCCL refuses two ranks on one GPU, so the 8 shards live in this process and ss_pagerank_run_group — the library's own sharded
loop (topic blocks, exchange on the second stream, events between the streams) with device copies standing in for the RCCL
all-gather — drives them: the edge-balanced 8-way row deal, the tail rows of 8 slices, the rank-order sums of 8 partials, the
float32 wire, the two-vector form and the 2 x 4 topic-group x doc-shard layout (ranking/pagerank.go:52-63 runs the topics one
after the other on one machine; the split is this repo's).  Against the oracle and the single-GPU run."""
import numpy as np
import pytest

from spaghettisearch_amd import sharding, synth

pytestmark = pytest.mark.gpu

D = 0.75
WORLD = 8


def _shards(ctx, n, ptr, dst, world=WORLD):
    from spaghettisearch_amd import engine
    return [engine.Graph(ctx, n, ptr, dst, rank=r, world=world) for r in range(world)]


def test_eight_way_deal_is_edge_balanced_and_complete(ss_ctx):
    n, e = 1 << 20, 5_000_000
    import torch
    ptr, dst = synth.rmat_graph_torch(n, e, seed=42, device=torch.device("cuda", 0))
    graphs = _shards(ss_ctx, n, ptr, dst)
    try:
        infos = [g.info() for g in graphs]
        assert sum(i.n_rows_local for i in infos) == n and sum(i.n_edges_local for i in infos) == e
        assert max(i.n_edges_local for i in infos) < 1.1 * e / WORLD            # R-MAT hubs are cut into the same deal
        assert len({i.n_nondangling for i in infos}) == 1                        # every rank numbers the same table
    finally:
        for g in graphs:
            g.close()


@pytest.mark.parametrize("k_topics,blocks", [(16, None), (16, 4), (2, None), (1, None)])
def test_world8_group_matches_oracle_on_a_1m_node_graph(ss_ctx, oracle, k_topics, blocks):
    """2^20 nodes / 5M edges (BASELINE config 2's graph) on 8 in-process shards, K = 16 (two 8-wide topic blocks, and four
    blocks), 2 and 1 (the wave-item kernel, whose shards also end in the two tail rows): ranks to 1e-12, iteration counts equal,
    bit-identical run to run, and equal to the single-GPU run to 1e-13."""
    import torch
    from spaghettisearch_amd import engine
    n, e = 1 << 20, 5_000_000
    dev = torch.device("cuda", 0)
    ptr, dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
    h_ptr, h_dst = ptr.cpu().numpy().view(np.uint64), dst.cpu().numpy().view(np.uint32)
    n_topic = synth.topic_sizes(n, k_topics)
    ref, ref_iters = oracle.pagerank(n, h_ptr, h_dst, D, 1e-6, n_topic)
    graphs = _shards(ss_ctx, n, ptr, dst)
    one = engine.Graph(ss_ctx, n, ptr, dst)
    try:
        with ss_ctx.options(pr__topic_blocks=blocks):
            rank, iters = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic)
            rank2, iters2 = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic)
        assert iters.tolist() == ref_iters.tolist()
        np.testing.assert_allclose(rank, ref, rtol=1e-12)
        assert rank.tobytes() == rank2.tobytes() and iters.tolist() == iters2.tolist()
        single, it1 = one.pagerank(D, 1e-6, n_topic)
        assert it1.tolist() == iters.tolist()
        np.testing.assert_allclose(rank, single, rtol=1e-13)
        # a fixed number of sweeps (the mode bench.py --gpus N times)
        ref5, _ = oracle.pagerank(n, h_ptr, h_dst, D, -1.0, n_topic[:2], max_iter=5)
        rank5, it5 = engine.Graph.pagerank_group(graphs, D, -1.0, n_topic[:2], max_iter=5)
        assert it5.tolist() == [5] * len(n_topic[:2])
        np.testing.assert_allclose(rank5, ref5, rtol=1e-12)
    finally:
        one.close()
        for g in graphs:
            g.close()


def test_world8_float32_wire_and_two_vector_form(ss_ctx, oracle):
    """The two opt-ins at world 8: float32 on the wire (ranks inside the 1e-6 gate, iteration counts within one) and the
    two-vector form (ranks to 1e-12, iteration counts equal), alone and together."""
    import torch
    from spaghettisearch_amd import engine
    n, e = 1 << 20, 5_000_000
    ptr, dst = synth.rmat_graph_torch(n, e, seed=42, device=torch.device("cuda", 0))
    h_ptr, h_dst = ptr.cpu().numpy().view(np.uint64), dst.cpu().numpy().view(np.uint32)
    graphs = _shards(ss_ctx, n, ptr, dst)
    try:
        for k_topics in (16, 2):
            n_topic = synth.topic_sizes(n, k_topics)
            ref, ref_iters = oracle.pagerank(n, h_ptr, h_dst, D, 1e-6, n_topic)
            with ss_ctx.options(pr__wire_f32=1):
                r32, i32 = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic)
                r32b, i32b = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic)
            assert np.abs(i32.astype(int) - ref_iters.astype(int)).max() <= 1
            same = i32 == ref_iters
            assert same.any()
            np.testing.assert_allclose(r32[same], ref[same], rtol=1e-6)
            assert r32.tobytes() == r32b.tobytes() and i32.tolist() == i32b.tolist()
            with ss_ctx.options(pr__affine=1):
                ra, ia = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic)
                ra2, ia2 = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic)
            assert ia.tolist() == ref_iters.tolist()
            np.testing.assert_allclose(ra, ref, rtol=1e-12)
            assert ra.tobytes() == ra2.tobytes() and ia.tolist() == ia2.tolist()
            with ss_ctx.options(pr__affine=1, pr__wire_f32=1):
                rb, ib = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic)
            assert np.abs(ib.astype(int) - ref_iters.astype(int)).max() <= 1
            sameb = ib == ref_iters
            np.testing.assert_allclose(rb[sameb], ref[sameb], rtol=1e-6)
        # 40 topics from two vectors on 8 shards (more topics than a K-wide state holds)
        n_topic = synth.topic_sizes(n, 40)
        ref, ref_iters = oracle.pagerank(n, h_ptr, h_dst, D, 1e-6, n_topic)
        with ss_ctx.options(pr__affine=1):
            ra, ia = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic)
        assert ia.tolist() == ref_iters.tolist()
        np.testing.assert_allclose(ra, ref, rtol=1e-12)
    finally:
        for g in graphs:
            g.close()


def test_two_topic_groups_by_four_doc_shards(ss_ctx, oracle):
    """The 2 x 4 layout of bench.py's `topic_groups_2_x_doc_shards_4` (ss_comm_split(color, key)): every one of the 8 ranks
    computes its (group, shard, topics) from sharding.topic_group_layout; the ranks of a group hold the 4 doc shards of ITS
    graph and run ITS 8 topics.  Here each group's loop runs through ss_pagerank_run_group; the union of what the 8 ranks
    own covers every (topic, node) exactly once and equals the oracle."""
    import torch
    from spaghettisearch_amd import engine
    n, e, kt, G = 300_000, 1_600_000, 16, 2
    ptr, dst = synth.rmat_graph(n, e, seed=9)
    n_topic = synth.topic_sizes(n, kt)
    ref, ref_iters = oracle.pagerank(n, ptr, dst, D, 1e-8, n_topic)
    layout = [sharding.topic_group_layout(r, WORLD, G, kt) for r in range(WORLD)]
    # (color, key) pairs are distinct and complete; a group's ranks sit G apart
    assert sorted((c, k) for c, k, _, _, _ in layout) == [(c, k) for c in range(G) for k in range(WORLD // G)]
    assert [r for r in range(WORLD) if layout[r][0] == 0] == [0, 2, 4, 6]
    covered = np.zeros((kt, n), dtype=np.int32)
    got = np.zeros((kt, n))
    got_iters = np.zeros(kt, dtype=np.int32)
    for color in range(G):
        members = sorted((lay[1], r) for r, lay in enumerate(layout) if lay[0] == color)       # by key = doc shard
        S, lo, hi = layout[members[0][1]][2], layout[members[0][1]][3], layout[members[0][1]][4]
        assert [k for k, _ in members] == list(range(S)) and S == 4 and hi - lo == 8
        graphs = [engine.Graph(ss_ctx, n, ptr, dst, rank=k, world=S) for k, _ in members]
        try:
            rank, iters = engine.Graph.pagerank_group(graphs, D, 1e-8, n_topic[lo:hi])
            # what each RANK of the group owns: its shard's rows of the group's topics
            states = [engine.PageRankState(g, D, 1e-8, n_topic[lo:hi]) for g in graphs]
            for st in states:
                ids, _ = st.read_local()
                covered[lo:hi, np.asarray(ids, dtype=np.int64)] += 1
                st.close()
            got[lo:hi] = rank
            got_iters[lo:hi] = iters
        finally:
            for g in graphs:
                g.close()
    assert (covered == 1).all()
    assert got_iters.tolist() == ref_iters.tolist()
    np.testing.assert_allclose(got, ref, rtol=1e-12)
    with pytest.raises(ValueError):
        sharding.topic_group_layout(0, 8, 3, 16)


def test_world8_on_the_config4_graph(ss_ctx, oracle):
    """BASELINE config 4 itself — 10M nodes / 50M edges / 16 topics — on 8 in-process doc-range shards: the library's pipelined
    loop against the single-GPU 16-wide run (1e-12, equal iteration counts), the oracle on the first and last topic, and the
    two opt-in variants (float32 wire inside the 1e-6 gate, two-vector form to 1e-12)."""
    import torch
    from spaghettisearch_amd import engine
    N, E, K = 10_000_000, 50_000_000, 16
    dev = torch.device("cuda", 0)
    out_ptr, out_dst = synth.rmat_graph_torch(N, E, seed=42, device=dev)
    n_topic = synth.topic_sizes(N, K)
    one = engine.Graph(ss_ctx, N, out_ptr, out_dst)
    single, it1 = one.pagerank(D, 1e-6, n_topic)
    one.close()
    graphs = _shards(ss_ctx, N, out_ptr, out_dst)
    try:
        infos = [g.info() for g in graphs]
        assert sum(i.n_edges_local for i in infos) == E and max(i.n_edges_local for i in infos) < 1.05 * E / WORLD
        rank, iters = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic)
        assert iters.tolist() == it1.tolist()
        np.testing.assert_allclose(rank, single, rtol=1e-12)
        h_ptr = out_ptr.cpu().numpy().view(np.uint64)
        h_dst = out_dst.cpu().numpy().view(np.uint32)
        for k in (0, K - 1):
            ref, ref_it = oracle.pagerank(N, h_ptr, h_dst, D, 1e-6, [int(n_topic[k])])
            assert int(iters[k]) == int(ref_it[0])
            np.testing.assert_allclose(rank[k], ref[0], rtol=1e-12)
        del h_ptr, h_dst
        with ss_ctx.options(pr__wire_f32=1):
            r32, i32 = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic)
        assert np.abs(i32.astype(int) - iters.astype(int)).max() <= 1
        same = i32 == iters
        np.testing.assert_allclose(r32[same], single[same], rtol=1e-6)
        del r32
        with ss_ctx.options(pr__affine=1):
            ra, ia = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic)
        assert ia.tolist() == iters.tolist()
        np.testing.assert_allclose(ra, single, rtol=1e-12)
        # K = 2 and K = 1 shards of the big graph (the wave-item kernel on 1/8 of the rows)
        for kk in (2, 1):
            rk, ik = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic[:kk])
            assert ik.tolist() == iters[:kk].tolist()
            np.testing.assert_allclose(rk, single[:kk], rtol=1e-12)
    finally:
        for g in graphs:
            g.close()
        del out_ptr, out_dst
        torch.cuda.empty_cache()
