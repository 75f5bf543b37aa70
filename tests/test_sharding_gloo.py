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
from tests.shard_model import NumpyShardState

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


@pytest.mark.parametrize("world", [2, 3])
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


@pytest.mark.parametrize("world", [2, 3])
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
