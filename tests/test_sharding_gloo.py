"""N>1 path on CPU: two processes, torch.distributed/gloo, spaghettisearch_amd.sharding driving a
CPU shard model that follows the library's layout and exchange protocol (tests/shard_model.py).
The same driver + DistExchange run on the GPU box with backend "nccl" (= RCCL over xGMI)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pyoracle
from spaghettisearch_amd import sharding, synth
from tests.shard_model import NumpyAffineShard, NumpyShardState

D, EPS = 0.75, 1e-10


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, e, n_topic, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ptr, dst = synth.rmat_graph(n, e, seed=21)
        st = NumpyShardState(n, ptr, dst, D, EPS, n_topic, rank, world)
        ex = sharding.DistExchange(st, torch.device("cpu"))
        final = sharding.iterate([st], ex, batch=3)
        ranks = sharding.gather_ranks(st)
        if rank == 0:
            np.savez(out_path, rank=ranks, iters=final["iters"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_two_process_gloo_matches_oracle(tmp_path, world):
    n, e = 4000, 22000
    n_topic = synth.topic_sizes(n, 4)
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(world, _free_port(), n, e, n_topic, out), nprocs=world, join=True)
    got = np.load(out)
    ptr, dst = synth.rmat_graph(n, e, seed=21)
    ref, ref_iters = pyoracle.pagerank(n, ptr, dst, D, EPS, n_topic)
    assert got["iters"].tolist() == ref_iters.tolist()
    np.testing.assert_allclose(got["rank"], ref, rtol=1e-12)


def test_single_process_local_exchange_model():
    # all shards in one process through LocalExchange (the GPU test uses the same driver with HIP states)
    n, e = 3000, 15000
    ptr, dst = synth.rmat_graph(n, e, seed=5)
    n_topic = synth.topic_sizes(n, 3)
    ref, ref_iters = pyoracle.pagerank(n, ptr, dst, D, EPS, n_topic)
    for world in (1, 2, 5):
        states = [NumpyShardState(n, ptr, dst, D, EPS, n_topic, r, world) for r in range(world)]
        if world == 1:
            class _NoExchange:
                def __call__(self):
                    pass
            ex = _NoExchange()
        else:
            ex = sharding.LocalExchange(states, torch.device("cpu"))
        rank, iters = sharding.run_sharded(states, ex, batch=4)
        assert iters.tolist() == ref_iters.tolist()
        np.testing.assert_allclose(rank, ref, rtol=1e-12)
    # shards are edge-balanced by the round-robin deal of degree-sorted rows
    sizes = [len(s.e_row) for s in states]
    assert max(sizes) < 1.25 * e / 5


def _worker_pipelined(rank, world, port, n, e, n_topic, sweeps, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ptr, dst = synth.rmat_graph(n, e, seed=21)
        half = len(n_topic) // 2
        blocks = [n_topic[:half], n_topic[half:]]
        states = [NumpyShardState(n, ptr, dst, D, -1.0, b, rank, world) for b in blocks]     # eps < 0: fixed number of sweeps
        exs = [sharding.DistExchange(s, torch.device("cpu")) for s in states]
        handles = sharding.prime_pipelined(states, exs)
        sharding.sweep_pipelined(states, exs, handles, sweeps)
        sharding.drain_pipelined(states, exs, handles)
        ranks = np.concatenate([sharding.gather_ranks(s) for s in states])
        if rank == 0:
            np.savez(out_path, rank=ranks)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_pipelined_topic_blocks_gloo_match_oracle(tmp_path, world):
    # two topic blocks per rank, the all-gather of one block in flight while the other block is swept
    n, e, sweeps = 3000, 16000, 6
    n_topic = synth.topic_sizes(n, 4)
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker_pipelined, args=(world, _free_port(), n, e, n_topic, sweeps, out), nprocs=world, join=True)
    got = np.load(out)
    ptr, dst = synth.rmat_graph(n, e, seed=21)
    ref, _ = pyoracle.pagerank(n, ptr, dst, D, -1.0, n_topic, max_iter=sweeps)
    np.testing.assert_allclose(got["rank"], ref, rtol=1e-12)


# ---- the two-vector form on doc-range shards (library option "pr.affine", csrc/pagerank.hip run_affine_sharded) ---------------------
def _worker_affine(rank, world, port, n, e, n_topic, eps, max_iter, out_path, lag=True):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ptr, dst = synth.rmat_graph(n, e, seed=31)
        st = NumpyAffineShard(n, ptr, dst, D, eps, n_topic, rank, world, max_iter=max_iter)
        ids, ranks, iters = st.run(lag=lag)
        # assemble by original id on rank 0
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([len(ids)], dtype=torch.int64))
        m = int(max(int(x) for x in sizes))
        pad_ids = torch.full((m,), -1, dtype=torch.int64)
        pad_ids[:len(ids)] = torch.from_numpy(ids.astype(np.int64))
        pad_rk = torch.zeros((len(n_topic), m), dtype=torch.float64)
        pad_rk[:, :len(ids)] = torch.from_numpy(ranks)
        all_ids = [torch.empty_like(pad_ids) for _ in range(world)]
        all_rk = [torch.empty_like(pad_rk) for _ in range(world)]
        dist.all_gather(all_ids, pad_ids)
        dist.all_gather(all_rk, pad_rk)
        if rank == 0:
            full = np.zeros((len(n_topic), n))
            for i_, r_ in zip(all_ids, all_rk):
                i_ = i_.numpy()
                ok = i_ >= 0
                full[:, i_[ok]] = r_.numpy()[:, ok]
            np.savez(out_path, rank=full, iters=iters)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,lag", [(2, True), (3, True), (8, True), (2, False), (8, False)])
def test_two_vector_form_on_gloo_shards_matches_oracle(tmp_path, world, lag):
    """The sharded two-vector protocol between real processes: lag = ONE exchange per iteration (the ranks' K local L1 sums of
    iteration i ride in the spare tail rows of iteration i + 1's 2-column slice; stop decisions one exchange late, from the sums
    added in rank order; a small exchange of their own only after the last sweep of a max_iter run), or the round-4 form with a
    second, small all-gather per iteration.  Every topic's ranks and iteration count as the oracle's."""
    n, e = 4000, 22000
    for n_topic, eps, max_iter in ((synth.topic_sizes(n, 9), EPS, 0), ([n, 7, 123], 1e-30, 4), (synth.topic_sizes(n, 64), 1e-8, 0)):
        out = str(tmp_path / f"aff{len(n_topic)}.npz")
        mp.spawn(_worker_affine, args=(world, _free_port(), n, e, list(n_topic), eps, max_iter, out, lag), nprocs=world, join=True)
        got = np.load(out)
        ptr, dst = synth.rmat_graph(n, e, seed=31)
        ref, ref_iters = pyoracle.pagerank(n, ptr, dst, D, eps, n_topic, max_iter=max_iter)
        assert got["iters"].tolist() == ref_iters.tolist()
        np.testing.assert_allclose(got["rank"], ref, rtol=1e-12)


def test_topic_group_layout_at_world_8():
    """bench.py's 2-D variants at 8 ranks (ss_comm_split(color, key)): 2 x 4 and 4 x 2 — every (group, shard) pair exactly once,
    a group's ranks `groups` apart, the groups' topic ranges a partition of the 16 topics."""
    for G in (1, 2, 4, 8):
        lay = [sharding.topic_group_layout(r, 8, G, 16) for r in range(8)]
        assert sorted((c, k) for c, k, *_ in lay) == [(c, k) for c in range(G) for k in range(8 // G)]
        assert all(s_ == 8 // G and hi - lo == 16 // G for _, _, s_, lo, hi in lay)
        for c in range(G):
            members = [r for r in range(8) if lay[r][0] == c]
            assert members == list(range(c, 8, G)) and [lay[r][1] for r in members] == list(range(8 // G))
        assert sorted({(lo, hi) for *_, lo, hi in lay}) == [(i * 16 // G, (i + 1) * 16 // G) for i in range(G)]
    for bad in ((8, 3, 16), (8, 2, 15), (6, 4, 16)):
        with pytest.raises(ValueError):
            sharding.topic_group_layout(0, *bad)
