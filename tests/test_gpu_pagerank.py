"""GPU parity: HIP PageRank (through the C ABI) vs the CPU oracle.

Reference: ranking/pagerank.go:14-145.  Gates: ranks within 1e-6
relative on x AND on the inherited part y = x*S-(1-d); iteration counts equal.
In practice fp64 agrees to ~1e-13; the tests assert much tighter than the gate.
"""
import numpy as np
import pytest

from spaghettisearch_amd import synth

pytestmark = pytest.mark.gpu

D = 0.75


def inherited(x, total, d=D):
    return x * total - (1.0 - d)


def run_both(ss_ctx, oracle, n, out_ptr, out_dst, n_topic, eps, max_iter=0, d=D):
    from spaghettisearch_amd import engine
    g = engine.Graph(ss_ctx, n, out_ptr, out_dst)
    try:
        rank, iters = g.pagerank(d, eps, n_topic, max_iter=max_iter)
    finally:
        g.close()
    ref, ref_iters = oracle.pagerank(n, out_ptr, out_dst, d, eps, n_topic, max_iter=max_iter)
    return rank, iters, ref, ref_iters


def csr(n, edges):
    edges = sorted(edges)
    ptr = np.zeros(n + 1, dtype=np.uint64)
    for s, _ in edges:
        ptr[s + 1] += 1
    return np.cumsum(ptr).astype(np.uint64), np.array([d for _, d in edges], dtype=np.uint32)


def test_kat_graph(ss_ctx, oracle):
    # the hand-worked 5-node graph of tests/test_oracle_kat.py (self-loop, frontier child, n_init != N)
    ptr, dst = csr(5, [(0, 1), (0, 2), (1, 2), (2, 0), (2, 3), (4, 4)])
    rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 5, ptr, dst, [4], 0.0, max_iter=1)
    np.testing.assert_allclose(rank[0], [19 / 58, 19 / 58, 25 / 58, 19 / 58, 22 / 58], rtol=1e-15)
    assert iters.tolist() == [1]
    rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 5, ptr, dst, [4, 5, 1000], 1e-12)
    assert iters.tolist() == ref_iters.tolist()
    np.testing.assert_allclose(rank, ref, rtol=1e-13)


@pytest.mark.parametrize("k_topics", [1, 2, 3, 4, 8, 16])
def test_rmat_small_all_group_widths(ss_ctx, oracle, k_topics):
    n, e = 20000, 100000
    ptr, dst = synth.rmat_graph(n, e, seed=100 + k_topics)
    n_topic = synth.topic_sizes(n, k_topics)
    rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, n, ptr, dst, n_topic, 1e-9)
    assert iters.tolist() == ref_iters.tolist()
    np.testing.assert_allclose(rank, ref, rtol=1e-12)


def test_more_topics_than_one_state(ss_ctx, oracle):
    # K > 16 is run as successive groups of 16 (topics are independent, pagerank.go:54-63)
    n, e = 3000, 12000
    ptr, dst = synth.rmat_graph(n, e, seed=5)
    n_topic = synth.topic_sizes(n, 19)
    rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, n, ptr, dst, n_topic, 1e-10)
    assert iters.tolist() == ref_iters.tolist()
    np.testing.assert_allclose(rank, ref, rtol=1e-12)


def test_inherited_part_and_fixed_iterations(ss_ctx, oracle):
    # ranks are nearly uniform at scale: compare the inherited part too
    n, e = 200000, 1000000
    ptr, dst = synth.rmat_graph(n, e, seed=42)
    from spaghettisearch_amd import engine
    g = engine.Graph(ss_ctx, n, ptr, dst)
    st = engine.PageRankState(g, D, -1.0, [n], max_iter=0)
    st.begin()
    st.step(6)
    s = st.status()
    x = st.read()[0]
    st.close()
    g.close()
    assert s["sweeps"] == 6 and s["n_active"] == 1 and s["iters"].tolist() == [6]
    ref, it, change, total = oracle.pagerank_topic_detail(n, ptr, dst, D, -1.0, n, max_iter=6)
    assert it == 6
    np.testing.assert_allclose(x, ref, rtol=1e-12)
    assert s["delta"][0] == pytest.approx(change, rel=1e-9)
    # y = x*S - (1-d) with the normaliser of the last sweep
    y, y_ref = inherited(x, total), inherited(ref, total)
    nz = y_ref > 1e-12
    np.testing.assert_allclose(y[nz], y_ref[nz], rtol=1e-6)
    np.testing.assert_allclose(y, y_ref, atol=1e-9 * y_ref.max())


def test_skewed_rows_block_per_segment(ss_ctx, oracle):
    # one hub with ~60k in-edges (multi-block row), a few medium rows, many empty rows
    rng = np.random.default_rng(3)
    n = 70000
    edges = {(int(s), 0) for s in range(1, 60001)}
    edges |= {(int(s), 1) for s in rng.choice(n, 3000, replace=False)}
    edges |= {(int(s), 2) for s in rng.choice(n, 300, replace=False)}
    edges |= {(int(a), int(b)) for a, b in rng.integers(0, n, size=(50000, 2))}
    ptr, dst = csr(n, list(edges))
    rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, n, ptr, dst, [n, 7], 1e-10)
    assert iters.tolist() == ref_iters.tolist()
    np.testing.assert_allclose(rank, ref, rtol=1e-12)
    rank16, iters16, ref16, ref_iters16 = run_both(ss_ctx, oracle, n, ptr, dst, synth.topic_sizes(n, 16), 1e-10)
    assert iters16.tolist() == ref_iters16.tolist()
    np.testing.assert_allclose(rank16, ref16, rtol=1e-12)


@pytest.mark.parametrize("k_topics", [1, 16])
def test_graph_build_chunk_rows(ss_ctx, oracle, k_topics):
    # ss_graph_create finds the row of every edge slot per 2048-slot chunk (k_chunk_first_rows + LDS marks + running maximum,
    # a bisection where a chunk holds a long stretch of empty rows).  A graph built to hit the corners: rows that start exactly
    # on chunk boundaries, rows spanning several chunks, > 8192 edge-less rows inside one chunk in BOTH directions (dangling
    # sources in id order; non-dangling pages nobody links to, which sort to the end of their class), an edge count that is a
    # multiple of the chunk, and the last row ending at the last slot.
    n = 120000
    edges = []
    hubs = list(range(100000, 100008))
    for s in range(3):                                             # three sources with exactly 2048 children: rows on chunk boundaries
        edges += [(s, 50000 + j) for j in range(2048)]
    edges += [(3, 20000 + j) for j in range(5000)]                 # a row over several chunks
    # sources 4 .. 29999 have no out-edges (dangling, a 26k-row empty stretch in the out-edge CSR)
    edges += [(s, hubs[s % 8]) for s in range(30000, 60000)]       # 30k non-dangling pages with no in-edges of their own (50000.. get some)
    edges += [(s, s + 1) for s in range(110000, 110100)]
    fill = (-len(edges)) % 2048
    edges += [(n - 1, 70000 + j) for j in range(fill)]             # the last row ends at the last slot of the last chunk
    assert len(edges) % 2048 == 0 and len(set(edges)) == len(edges)
    ptr, dst = csr(n, edges)
    n_topic = synth.topic_sizes(n, k_topics)
    rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, n, ptr, dst, n_topic, 1e-10)
    assert iters.tolist() == ref_iters.tolist()
    np.testing.assert_allclose(rank, ref, rtol=1e-12)


@pytest.mark.parametrize("opts", [{"graph__late_free": 0}, {"pr__deal_global": 0}, {"pr__deal_global": 0, "pr__deal_snake": 1},
                                  {"pr__deal_global": 1, "pr__deal_snake": 0}, {"pr__deal_global": 2}, {"pr__deal_global": 1},
                                  {"graph__late_free": 0, "pr__deal_global": 0, "pr__deal_snake": 0}])
@pytest.mark.parametrize("k_topics", [1, 16])
def test_build_and_deal_options_agree(ss_ctx, oracle, opts, k_topics):
    # the switches of round 4's set-up path (build temporaries freed late or at once; work items dealt from one global order or
    # chunk by chunk, in alternating direction or least-loaded-first) only move work: same ranks, same iteration counts
    n, e = 60000, 400000
    ptr, dst = synth.rmat_graph(n, e, seed=321)
    n_topic = synth.topic_sizes(n, k_topics)
    with ss_ctx.options(**opts):
        rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, n, ptr, dst, n_topic, 1e-10)
    assert iters.tolist() == ref_iters.tolist()
    np.testing.assert_allclose(rank, ref, rtol=1e-12)


# ---- the two-vector form (option "pr.affine"): every topic of the reference's recurrence from two vectors ---------------------
@pytest.mark.parametrize("k_topics", [1, 3, 16, 40])
def test_two_vector_form_matches_the_oracle(ss_ctx, oracle, k_topics):
    """x_k = (p*u_k + q) / (r*u_k + s) is closed under pagerank.go:104-119, so two vectors carry all topics.  Not the reference's
    operation order: ranks to 1e-12 (gate 1e-6), the inherited part too; iteration counts equal here (no stop decision of these
    runs sits at a rounding tie)."""
    n, e = 30000, 160000
    ptr, dst = synth.rmat_graph(n, e, seed=500 + k_topics)
    n_topic = synth.topic_sizes(n, k_topics)
    for eps, max_iter in ((1e-9, 0), (1e-30, 6)):
        with ss_ctx.options(pr__affine=1):
            rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, n, ptr, dst, n_topic, eps, max_iter=max_iter)
        assert iters.tolist() == ref_iters.tolist()
        np.testing.assert_allclose(rank, ref, rtol=1e-12)


def test_two_vector_form_on_hard_graphs(ss_ctx, oracle):
    with ss_ctx.options(pr__affine=1):
        # the hand-worked graph of test_kat_graph: first iteration exact, then to convergence with very different topic sizes
        ptr, dst = csr(5, [(0, 1), (0, 2), (1, 2), (2, 0), (2, 3), (4, 4)])
        rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 5, ptr, dst, [4], 0.0, max_iter=1)
        np.testing.assert_allclose(rank[0], [19 / 58, 19 / 58, 25 / 58, 19 / 58, 22 / 58], rtol=1e-15)
        rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 5, ptr, dst, [4, 5, 1000, 1], 1e-12)
        assert iters.tolist() == ref_iters.tolist()
        np.testing.assert_allclose(rank, ref, rtol=1e-13)
        # no edges at all, a single self loop, a hub with 60k in-edges and many edge-less rows
        ptr = np.zeros(11, dtype=np.uint64)
        rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 10, ptr, np.zeros(0, np.uint32), [10, 3], 1e-12)
        assert iters.tolist() == ref_iters.tolist()
        np.testing.assert_allclose(rank, ref, rtol=1e-14)
        ptr, dst = csr(1, [(0, 0)])
        rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 1, ptr, dst, [1], 1e-12)
        assert iters.tolist() == ref_iters.tolist()
        np.testing.assert_allclose(rank, ref, rtol=1e-14)
        n, ptr, dst = _skewed_graph()
        rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, n, ptr, dst, synth.topic_sizes(n, 16), 1e-10)
        assert iters.tolist() == ref_iters.tolist()
        np.testing.assert_allclose(rank, ref, rtol=1e-12)
        # damping 1 (no teleport: the form's s vanishes, every topic is p / r) and a small damping
        ptr, dst = synth.rmat_graph(5000, 40000, seed=9)
        for dd in (1.0, 0.15):
            rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 5000, ptr, dst, [5000, 40, 1], 1e-9, max_iter=60, d=dd)
            assert iters.tolist() == ref_iters.tolist()
            np.testing.assert_allclose(rank, ref, rtol=1e-11)


def test_edge_cases(ss_ctx, oracle):
    # no edges at all: every node dangling (pagerank.go:131-134)
    ptr = np.zeros(11, dtype=np.uint64)
    rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 10, ptr, np.zeros(0, np.uint32), [10, 3], 1e-12)
    assert iters.tolist() == ref_iters.tolist()
    np.testing.assert_allclose(rank, ref, rtol=1e-14)
    # single node with a self loop
    ptr, dst = csr(1, [(0, 0)])
    rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 1, ptr, dst, [1], 1e-12)
    assert iters.tolist() == ref_iters.tolist()
    np.testing.assert_allclose(rank, ref, rtol=1e-14)
    # complete graph on 40 nodes
    ptr, dst = csr(40, [(a, b) for a in range(40) for b in range(40)])
    rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 40, ptr, dst, [40], 1e-13)
    assert iters.tolist() == ref_iters.tolist()
    np.testing.assert_allclose(rank, ref, rtol=1e-13)
    # max_iter cut
    ptr, dst = synth.rmat_graph(1000, 5000, seed=1)
    rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 1000, ptr, dst, [1000, 10], 1e-30, max_iter=3)
    assert iters.tolist() == [3, 3] == ref_iters.tolist()
    np.testing.assert_allclose(rank, ref, rtol=1e-13)


# ---- the three kernels of K <= 2 -----------------------------------------------------------------------------------------
# default (round 4): k_pr_sweep_n<1/2>, wave-owned items, a lane per row / per edge (classes V_SEG with the multi-piece ticket
# path, V_ROWW, V_QUAD, V_DEG by lane, V_ZERO); "pr.narrow_wave" = 0: the choice before it (small graphs: k_pr_sweep<8> with padded
# topics); "pr.force_narrow" = 1: the block-item kernel k_pr_step<1/2> (W_SEG, W_WAVE, W_GROUP, W_ZERO).  Every one of them meets
# the oracle directly.
NARROW_VARIANTS = {"wave_items": {}, "padded_8_wide": {"pr__narrow_wave": 0}, "block_items": {"pr__force_narrow": 1}}
def _skewed_graph():
    rng = np.random.default_rng(3)
    n = 70000
    edges = {(int(s), 0) for s in range(1, 60001)}                      # hub: 60k in-edges = many W_SEG segments
    edges |= {(int(s), 1) for s in rng.choice(n, 3000, replace=False)}  # several segments at K=1 (128*64 edges each)
    edges |= {(int(s), 2) for s in rng.choice(n, 300, replace=False)}
    edges |= {(int(a), int(b)) for a, b in rng.integers(0, n, size=(50000, 2))}
    return (n,) + csr(n, list(edges))


@pytest.mark.parametrize("variant", list(NARROW_VARIANTS))
@pytest.mark.parametrize("k_topics", [1, 2])
def test_narrow_kernel_skewed_rows(ss_ctx, oracle, k_topics, variant):
    n, ptr, dst = _skewed_graph()
    n_topic = [n, 7][:k_topics]
    with ss_ctx.options(**NARROW_VARIANTS[variant]):
        rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, n, ptr, dst, n_topic, 1e-10)
        rank2, iters2, _, _ = run_both(ss_ctx, oracle, n, ptr, dst, n_topic, 1e-10)
    assert iters.tolist() == ref_iters.tolist()
    np.testing.assert_allclose(rank, ref, rtol=1e-12)
    assert rank.tobytes() == rank2.tobytes() and iters.tolist() == iters2.tolist()     # run-to-run bit-identical
    # and the kernel the default picks gives the same ranks
    wide, wide_it, _, _ = run_both(ss_ctx, oracle, n, ptr, dst, n_topic, 1e-10)
    assert wide_it.tolist() == iters.tolist()
    np.testing.assert_allclose(wide, rank, rtol=1e-13)


@pytest.mark.parametrize("variant", list(NARROW_VARIANTS))
@pytest.mark.parametrize("k_topics", [1, 2])
def test_narrow_kernel_rmat_and_edge_cases(ss_ctx, oracle, k_topics, variant):
    with ss_ctx.options(**NARROW_VARIANTS[variant]):
        n, e = 20000, 100000
        ptr, dst = synth.rmat_graph(n, e, seed=100 + k_topics)
        n_topic = synth.topic_sizes(n, k_topics)
        rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, n, ptr, dst, n_topic, 1e-9)
        assert iters.tolist() == ref_iters.tolist()
        np.testing.assert_allclose(rank, ref, rtol=1e-12)
        nt = [10, 3][:k_topics]
        # no edges at all: every node dangling (pagerank.go:131-134)
        ptr = np.zeros(11, dtype=np.uint64)
        rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 10, ptr, np.zeros(0, np.uint32), nt, 1e-12)
        assert iters.tolist() == ref_iters.tolist()
        np.testing.assert_allclose(rank, ref, rtol=1e-14)
        # single node with a self loop
        ptr, dst = csr(1, [(0, 0)])
        rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 1, ptr, dst, [1] * k_topics, 1e-12)
        assert iters.tolist() == ref_iters.tolist()
        np.testing.assert_allclose(rank, ref, rtol=1e-14)
        # complete graph on 40 nodes
        ptr, dst = csr(40, [(a, b) for a in range(40) for b in range(40)])
        rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 40, ptr, dst, [40, 9][:k_topics], 1e-13)
        assert iters.tolist() == ref_iters.tolist()
        np.testing.assert_allclose(rank, ref, rtol=1e-13)
        # max_iter cut
        ptr, dst = synth.rmat_graph(1000, 5000, seed=1)
        rank, iters, ref, ref_iters = run_both(ss_ctx, oracle, 1000, ptr, dst, [1000, 10][:k_topics], 1e-30, max_iter=3)
        assert iters.tolist() == [3] * k_topics == ref_iters.tolist()
        np.testing.assert_allclose(rank, ref, rtol=1e-13)
        # the reference's own stop threshold (start_crawl.go:175): to the floating-point fixed point
        ptr, dst = synth.rmat_graph(30000, 160000, seed=3)
        nt = [15000, 7][:k_topics]
        ref, ref_it = oracle.pagerank(30000, ptr, dst, 0.75, 1e-20, nt, max_iter=300)
        from spaghettisearch_amd import engine
        g = engine.Graph(ss_ctx, 30000, ptr, dst)
        rank, it = g.pagerank(0.75, 1e-20, nt, max_iter=300)
        g.close()
        assert ref_it.max() < 300 and np.abs(it - ref_it).max() <= 1
        np.testing.assert_allclose(rank, ref, rtol=1e-12)


def test_option_names_are_checked(ss_ctx):
    from spaghettisearch_amd import SpaghettiError
    with pytest.raises(SpaghettiError):
        ss_ctx.set_option("pr.no_such_switch", 1)
    ss_ctx.set_option("pr.force_narrow", 1)
    ss_ctx.set_option("pr.force_narrow", None)


def test_bad_input_is_rejected(ss_ctx):
    from spaghettisearch_amd import SpaghettiError, engine
    ptr = np.array([0, 1, 2], dtype=np.uint64)
    with pytest.raises(SpaghettiError):
        engine.Graph(ss_ctx, 2, ptr, np.array([0, 5], dtype=np.uint32))      # child id out of range
    with pytest.raises(SpaghettiError):
        engine.Graph(ss_ctx, 2, np.array([0, 2, 1], dtype=np.uint64), np.array([0, 1], dtype=np.uint32))


def test_sharded_layout_single_process(ss_ctx, oracle):
    """world=2/4 shards driven from ONE process: the host plays the all-gather.
    Checks the doc-range sharding + exchange protocol of SURVEY.md §8e on the real kernels."""
    import torch
    from spaghettisearch_amd import engine
    from spaghettisearch_amd.sharding import LocalExchange, run_sharded
    n, e = 30000, 160000
    ptr, dst = synth.rmat_graph(n, e, seed=77)
    stream = torch.cuda.Stream()          # library kernels and the torch copies share one stream
    ss_ctx.set_stream(stream.cuda_stream)
    try:
        with torch.cuda.stream(stream):
            # (K = 5: the 8-wide sweep; K = 2, 1: the wave-item kernel of one or two topics, whose shards also end in the two tail rows)
            for world, kt in ((2, 5), (4, 5), (2, 2), (3, 1)):
                n_topic = synth.topic_sizes(n, kt)
                ref, ref_iters = oracle.pagerank(n, ptr, dst, D, 1e-9, n_topic)
                graphs = [engine.Graph(ss_ctx, n, ptr, dst, rank=r, world=world) for r in range(world)]
                infos = [g.info() for g in graphs]
                assert sum(i.n_rows_local for i in infos) == n
                assert sum(i.n_edges_local for i in infos) == e
                assert max(i.n_edges_local for i in infos) < 1.2 * e / world   # edge-balanced shards
                states = [engine.PageRankState(g, D, 1e-9, n_topic) for g in graphs]
                rank, iters = run_sharded(states, LocalExchange(states, torch.device("cuda:0")))
                for s in states:
                    s.close()
                for g in graphs:
                    g.close()
                assert iters.tolist() == ref_iters.tolist()
                np.testing.assert_allclose(rank, ref, rtol=1e-12)
    finally:
        torch.cuda.synchronize()
        ss_ctx.set_stream(None)


@pytest.mark.parametrize("world", [2, 4])
def test_in_library_pipelined_sharded_run(ss_ctx, oracle, world):
    """ss_pagerank_run_group: the library's own sharded loop (topic blocks, exchanges on the context's second stream, events
    between the streams) with all shards in this process and device copies standing in for the RCCL all-gather."""
    from spaghettisearch_amd import engine
    n, e = 30000, 160000
    ptr, dst = synth.rmat_graph(n, e, seed=77)
    graphs = [engine.Graph(ss_ctx, n, ptr, dst, rank=r, world=world) for r in range(world)]
    try:
        for k_topics, blocks in ((16, None), (16, 4), (5, None), (9, 3), (2, 2), (1, None)):
            n_topic = synth.topic_sizes(n, k_topics)
            ref, ref_iters = oracle.pagerank(n, ptr, dst, D, 1e-9, n_topic)
            with ss_ctx.options(pr__topic_blocks=blocks):
                rank, iters = engine.Graph.pagerank_group(graphs, D, 1e-9, n_topic)
                rank2, iters2 = engine.Graph.pagerank_group(graphs, D, 1e-9, n_topic)
            assert iters.tolist() == ref_iters.tolist(), (k_topics, blocks)
            np.testing.assert_allclose(rank, ref, rtol=1e-12)
            assert rank.tobytes() == rank2.tobytes() and iters.tolist() == iters2.tolist()     # run to run
            # a block's results are bit for bit those of its topics run on their own (topics are independent: pagerank.go:54-63)
            b = blocks if blocks else (2 if k_topics >= 8 else 1)
            b = min(b, k_topics)
            bounds = [k_topics * i // b for i in range(b + 1)]
            with ss_ctx.options(pr__topic_blocks=1):
                for lo, hi in zip(bounds[:-1], bounds[1:]):
                    alone, it_alone = engine.Graph.pagerank_group(graphs, D, 1e-9, n_topic[lo:hi])
                    assert alone.tobytes() == rank[lo:hi].tobytes() and it_alone.tolist() == iters[lo:hi].tolist()
        # fixed iteration count (max_iter) and the reference's eps = 1e-20
        n_topic = synth.topic_sizes(n, 6)
        ref, ref_iters = oracle.pagerank(n, ptr, dst, D, 1e-30, n_topic, max_iter=5)
        rank, iters = engine.Graph.pagerank_group(graphs, D, 1e-30, n_topic, max_iter=5)
        assert iters.tolist() == [5] * 6 == ref_iters.tolist()
        np.testing.assert_allclose(rank, ref, rtol=1e-13)
    finally:
        for g in graphs:
            g.close()


@pytest.mark.parametrize("world,lag", [(2, 1), (4, 1), (3, 0)])
def test_two_vector_form_on_a_sharded_graph(ss_ctx, oracle, world, lag):
    """Option "pr.affine" on doc-range shards (ss_pagerank_run_group: all shards in this process, device copies standing in for
    the RCCL all-gathers): ONE K = 2 state per shard, a two-column exchange per iteration whatever the topic count, the topics'
    L1 changes summed over the shards in rank order, every shard writing the ranks of its own rows.  Against the oracle
    (rtol 1e-12, equal iteration counts), equal to the single-GPU two-vector run to 1e-13, and bit-identical run to run."""
    from spaghettisearch_amd import engine
    n, e = 30000, 160000
    ptr, dst = synth.rmat_graph(n, e, seed=78)
    graphs = [engine.Graph(ss_ctx, n, ptr, dst, rank=r, world=world) for r in range(world)]
    one = engine.Graph(ss_ctx, n, ptr, dst)
    try:
        # lag = 1 (default): ONE exchange per iteration — the per-topic sums of iteration i ride in the spare tail rows of iteration
        # i + 1's slice, decisions one exchange late, a flush after the last sweep of a max_iter run; 0: round 4's second small exchange
        with ss_ctx.options(pr__affine=1, pr__affine_lag=lag):
            for k_topics in (1, 16, 40, 64, 100):          # (100: more sums than the spare rows carry — the sums keep their own collective)
                n_topic = synth.topic_sizes(n, k_topics)
                ref, ref_iters = oracle.pagerank(n, ptr, dst, D, 1e-9, n_topic)
                rank, iters = engine.Graph.pagerank_group(graphs, D, 1e-9, n_topic)
                rank2, iters2 = engine.Graph.pagerank_group(graphs, D, 1e-9, n_topic)
                assert iters.tolist() == ref_iters.tolist(), k_topics
                np.testing.assert_allclose(rank, ref, rtol=1e-12)
                assert rank.tobytes() == rank2.tobytes() and iters.tolist() == iters2.tolist()
                if k_topics <= 64:                 # (ss_pagerank_run takes SS_MAX_TOPICS = 64; the sharded calls take up to 256 in this form)
                    single, it1 = one.pagerank(D, 1e-9, n_topic)
                    assert it1.tolist() == iters.tolist()
                    np.testing.assert_allclose(rank, single, rtol=1e-13)
            n_topic = synth.topic_sizes(n, 6)
            ref, ref_iters = oracle.pagerank(n, ptr, dst, D, 1e-30, n_topic, max_iter=5)
            rank, iters = engine.Graph.pagerank_group(graphs, D, 1e-30, n_topic, max_iter=5)
            assert iters.tolist() == [5] * 6 == ref_iters.tolist()
            np.testing.assert_allclose(rank, ref, rtol=1e-13)
            # both opt-ins at once: two columns on the wire, as float32 (inside the 1e-6 gate, iteration counts within one)
            n_topic = synth.topic_sizes(n, 16)
            ref, ref_iters = oracle.pagerank(n, ptr, dst, D, 1e-7, n_topic)
            with ss_ctx.options(pr__wire_f32=1):
                rank, iters = engine.Graph.pagerank_group(graphs, D, 1e-7, n_topic)
            assert np.max(np.abs(iters.astype(int) - ref_iters.astype(int))) <= 1
            np.testing.assert_allclose(rank, ref, rtol=1e-6)
    finally:
        one.close()
        for g in graphs:
            g.close()


@pytest.mark.parametrize("world", [2, 3])
def test_float32_wire_exchange_stays_inside_the_gate(ss_ctx, oracle, world):
    """VERDICT r3 #6b, option "pr.wire_f32" (opt-in): the sharded sweep's contribution slices cross the links as float32, the tail
    rows (partial sums) as (hi, lo) float pairs.  Not the reference's float64 arithmetic, so not the default — but inside the
    parity gate: ranks within 1e-6 relative of the oracle, iteration counts within one; and deterministic run to run."""
    from spaghettisearch_amd import engine
    n, e = 60000, 400000
    ptr, dst = synth.rmat_graph(n, e, seed=91)
    graphs = [engine.Graph(ss_ctx, n, ptr, dst, rank=r, world=world) for r in range(world)]
    try:
        for k_topics in (16, 5, 2, 1):
            n_topic = synth.topic_sizes(n, k_topics)
            ref, ref_iters = oracle.pagerank(n, ptr, dst, D, 1e-6, n_topic)
            exact, exact_iters = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic)
            with ss_ctx.options(pr__wire_f32=1):
                rank, iters = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic)
                rank2, iters2 = engine.Graph.pagerank_group(graphs, D, 1e-6, n_topic)
            assert np.abs(iters - ref_iters).max() <= 1, (k_topics, iters, ref_iters)
            same_it = iters == ref_iters
            if same_it.any():
                np.testing.assert_allclose(rank[same_it], ref[same_it], rtol=1e-6)
            assert rank.tobytes() == rank2.tobytes() and iters.tolist() == iters2.tolist()
            assert rank.tobytes() != exact.tobytes()                     # the option took effect ...
            assert exact_iters.tolist() == ref_iters.tolist()            # ... and the default is untouched
            np.testing.assert_allclose(exact, ref, rtol=1e-12)
    finally:
        for g in graphs:
            g.close()


def _mp_worker(rank, world, port, n, e, n_topic, out_path):
    """One process per shard, all on cuda:0: the production driver (sharding.iterate + DistExchange +
    gather_ranks) with the real HIP states; gloo + host-staged exchange stand in for RCCL."""
    import os
    import torch
    import torch.distributed as dist
    from spaghettisearch_amd import engine, sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ctx = engine.Context(0)
        stream = torch.cuda.Stream()
        ctx.set_stream(stream.cuda_stream)
        with torch.cuda.stream(stream):
            ptr, dst = synth.rmat_graph(n, e, seed=31)
            g = engine.Graph(ctx, n, ptr, dst, rank=rank, world=world)
            st = engine.PageRankState(g, D, 1e-9, n_topic)
            ex = sharding.DistExchange(st, torch.device("cuda:0"), host_staged=True)
            final = sharding.iterate([st], ex, batch=3)
            ranks = sharding.gather_ranks(st)
            st.close()
            g.close()
        torch.cuda.synchronize()
        ctx.set_stream(None)
        ctx.close()
        if rank == 0:
            np.savez(out_path, rank=ranks, iters=final["iters"])
    finally:
        dist.destroy_process_group()


def test_two_processes_one_gpu_rehearsal(tmp_path, oracle):
    import socket
    import torch.multiprocessing as mp
    n, e = 20000, 110000
    n_topic = synth.topic_sizes(n, 16)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "mp.npz")
    mp.spawn(_mp_worker, args=(2, port, n, e, n_topic, out), nprocs=2, join=True)
    got = np.load(out)
    ptr, dst = synth.rmat_graph(n, e, seed=31)
    ref, ref_iters = oracle.pagerank(n, ptr, dst, D, 1e-9, n_topic)
    assert got["iters"].tolist() == ref_iters.tolist()
    np.testing.assert_allclose(got["rank"], ref, rtol=1e-12)


@pytest.mark.parametrize("n,e,seed", [(1000, 20000, 2), (30000, 160000, 3), (200000, 3000000, 6)])
def test_reference_stop_threshold(ss_ctx, oracle, n, e, seed):
    # start_crawl.go:175 calls UpdateTopicSensitivePagerank(ctx, 0.75, 1e-20, ...): the loop runs to the floating-point
    # fixed point.  Pull and push sum in different orders, so the last iteration may differ by one (SURVEY.md §8d gate).
    from spaghettisearch_amd import engine
    ptr, dst = synth.rmat_graph(n, e, seed=seed)
    n_topic = [n // 2, n, 7, n // 3 + 1]
    for d in (0.75, 0.85):
        ref, ref_it = oracle.pagerank(n, ptr, dst, d, 1e-20, n_topic, max_iter=300)
        g = engine.Graph(ss_ctx, n, ptr, dst)
        rank, it = g.pagerank(d, 1e-20, n_topic, max_iter=300)
        g.close()
        assert ref_it.max() < 300 and np.abs(it - ref_it).max() <= 1, (it.tolist(), ref_it.tolist())
        np.testing.assert_allclose(rank, ref, rtol=1e-12)


def test_run_to_run_bit_identical(ss_ctx):
    # static work deal, fixed-order partial sums, no float atomics: two runs (and two states on one graph) agree bit for bit
    from spaghettisearch_amd import engine
    n, e = 150000, 900000
    ptr, dst = synth.rmat_graph(n, e, seed=77)
    outs = []
    for k_topics in (16, 5, 1):
        n_topic = synth.topic_sizes(n, k_topics)
        g = engine.Graph(ss_ctx, n, ptr, dst)
        r1, i1 = g.pagerank(0.75, 1e-12, n_topic)
        r2, i2 = g.pagerank(0.75, 1e-12, n_topic)
        g.close()
        g = engine.Graph(ss_ctx, n, ptr, dst)
        r3, i3 = g.pagerank(0.75, 1e-12, n_topic)
        g.close()
        assert i1.tolist() == i2.tolist() == i3.tolist()
        assert r1.tobytes() == r2.tobytes() == r3.tobytes()
        outs.append(r1)


@pytest.mark.parametrize("k_topics", [1, 2])
def test_sweeps_inside_one_launch_are_bit_identical(ss_ctx, oracle, k_topics):
    """k_pr_multi_n (round 5; option "pr.persistent", on by default for K <= 2 on graphs whose contribution table is cache-resident):
    ss_pr_step's sweeps run inside one launch, the blocks waiting for each other between two sweeps.  Same arithmetic in the same
    order as one launch per sweep: ranks and iteration counts bit for bit, also against the oracle, for eps stops inside a batch of
    sweeps, max_iter cuts, a hub whose row is cut into pieces (tickets reused sweep after sweep), and graphs without edges."""
    from spaghettisearch_amd import engine
    cases = [synth.rmat_graph(30000, 160000, seed=77), synth.rmat_graph(1 << 18, 1_500_000, seed=5)]
    # a 70k-edge hub (V_SEG pieces) beside short rows
    rng = np.random.default_rng(4)
    n = 90000
    src = np.concatenate([rng.integers(0, n, 70000), rng.integers(0, n, 200000)])
    dst = np.concatenate([np.full(70000, 17), rng.integers(0, n, 200000)])
    pairs = np.unique(np.stack([src, dst], 1), axis=0)
    ptr = np.zeros(n + 1, dtype=np.uint64)
    np.add.at(ptr, pairs[:, 0] + 1, 1)
    cases.append((np.cumsum(ptr).astype(np.uint64), pairs[:, 1].astype(np.uint32)))
    cases.append((np.zeros(1001, dtype=np.uint64), np.zeros(0, dtype=np.uint32)))          # no edges at all
    for ptr, dst in cases:
        n = len(ptr) - 1
        n_topic = synth.topic_sizes(n, k_topics)
        for eps, max_iter in ((1e-9, 0), (1e-20, 300), (-1.0, 11)):
            ref, ref_it = oracle.pagerank(n, ptr, dst, D, eps, n_topic, max_iter=max_iter)
            got = {}
            for mode in (0, 1, 2):                # one launch per sweep; write-through hand-offs; release / acquire fences
                with ss_ctx.options(pr__persistent=mode):
                    g = engine.Graph(ss_ctx, n, ptr, dst)
                    got[mode] = g.pagerank(D, eps, n_topic, max_iter=max_iter)
                    g.close()
            for mode in (1, 2):
                assert got[0][1].tolist() == got[mode][1].tolist(), (mode, eps)
                assert got[0][0].tobytes() == got[mode][0].tobytes(), (mode, eps)
            assert np.abs(got[1][1].astype(int) - ref_it.astype(int)).max() <= (1 if eps == 1e-20 else 0)
            np.testing.assert_allclose(got[1][0], ref, rtol=1e-12)
    # stepping by hand: 3 + 1 + 5 sweeps inside launches == 9 launches
    ptr, dst = cases[0]
    n = len(ptr) - 1
    n_topic = synth.topic_sizes(n, k_topics)
    xs = {}
    for mode in (0, 1, 2):
        with ss_ctx.options(pr__persistent=mode):
            g = engine.Graph(ss_ctx, n, ptr, dst)
            st = engine.PageRankState(g, D, -1.0, n_topic, max_iter=0)
            st.begin()
            for m in (3, 1, 5):
                st.step(m)
            assert st.status()["sweeps"] == 9
            xs[mode] = st.read()
            st.close()
            g.close()
    assert xs[0].tobytes() == xs[1].tobytes() == xs[2].tobytes()
