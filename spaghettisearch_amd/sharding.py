"""Multi-GPU host logic for the doc-range-sharded PageRank.

One process per GPU.  Rank r owns the destination rows of shard r (the library
deals degree-sorted rows round-robin to the ranks, so shards are edge-balanced);
every sweep each rank computes its rows' new ranks and next-sweep contributions
(ss_pr_step), then ONE collective — an all-gather of the contribution slices of
the non-dangling nodes, each slice carrying its rank's partial sums (normaliser
and L1 delta) in two trailing rows — rebuilds the full contribution table on
every rank; ss_pr_finalize then combines the partial sums in rank order and
applies the stop rule (pagerank.go:93) identically on every rank.

The collective is torch.distributed's all_gather_into_tensor (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests); the compute engine is
whatever object implements the small state protocol below — in the product that
is engine.PageRankState (HIP kernels).  This module never computes ranks itself.

State protocol: begin(), step(n), finalize(), status() -> dict(n_active, iters,
sweeps, ...), read_local() -> (ids, rank[K][rows]), exchange_tensors() ->
(send, recv) torch tensors, attributes k (topics) and n_nodes.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np


class _DevMem:
    """Expose a raw device pointer through __cuda_array_interface__ (float64 view)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def tensor_from_ptr(ptr: int, nbytes: int, device):
    import torch
    return torch.as_tensor(_DevMem(ptr, nbytes), device=device)


def exchange_tensors(state, device):
    """(send, recv) float64 torch tensors aliasing the library's exchange buffers of `state`."""
    if hasattr(state, "exchange_tensors"):
        return state.exchange_tensors()
    sp, sb, rp, rb = state.exchange_buffers()
    return tensor_from_ptr(sp, sb, device), tensor_from_ptr(rp, rb, device)


class DistExchange:
    """One process per GPU: all-gather this rank's slice into the full table (RCCL / gloo).

    host_staged=True copies through pinned host buffers around the collective: only for
    rehearsing the multi-process path with the gloo backend (several ranks on one GPU); the
    production path hands the device buffers straight to RCCL."""

    def __init__(self, state, device, group=None, host_staged: bool = False):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.send, self.recv = exchange_tensors(state, device)
        world = dist.get_world_size(group)
        if self.recv.numel() != world * self.send.numel():
            raise ValueError(f"exchange buffers do not match world size {world}: "
                             f"send {self.send.numel()} recv {self.recv.numel()}")
        self.host_staged = host_staged and self.send.is_cuda
        if self.host_staged:
            self.h_send = torch.empty(self.send.shape, dtype=self.send.dtype, pin_memory=True)
            self.h_recv = torch.empty(self.recv.shape, dtype=self.recv.dtype, pin_memory=True)

    def __call__(self) -> None:
        if self.host_staged:
            import torch
            self.h_send.copy_(self.send, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            self.dist.all_gather_into_tensor(self.h_recv, self.h_send, group=self.group)
            self.recv.copy_(self.h_recv, non_blocking=True)
            return
        self.dist.all_gather_into_tensor(self.recv, self.send, group=self.group)


class LocalExchange:
    """All shards live in ONE process (tests on a single GPU / CPU): plays the all-gather by copies."""

    def __init__(self, states: Sequence, device):
        self.pairs = [exchange_tensors(s, device) for s in states]

    def __call__(self) -> None:
        import torch
        full = torch.cat([send for send, _ in self.pairs])
        for _, recv in self.pairs:
            recv.copy_(full)


def iterate(states: Sequence, exchange, batch: int = 8, max_sweeps: Optional[int] = None) -> dict:
    """Run the sharded power iteration to convergence (or max_sweeps).  `states` are the shard
    states living in THIS process (one in production).  Returns the final status of states[0]."""
    for s in states:
        s.begin()
    exchange()
    for s in states:
        s.finalize()
    done = 0
    while True:
        todo = batch if max_sweeps is None else min(batch, max_sweeps - done)
        if todo <= 0:
            break
        for _ in range(todo):
            for s in states:
                s.step(1)
            exchange()
            for s in states:
                s.finalize()
        done += todo
        st = states[0].status()      # host looks at the device-side stop rule once per batch
        if st["n_active"] == 0:
            return st
    return states[0].status()


def assemble(parts: Sequence, n_nodes: int, k: int) -> np.ndarray:
    """parts = [(ids, rank[K][rows]), ...] from every shard -> rank [K][n_nodes] in original ids."""
    out = np.full((k, n_nodes), np.nan, dtype=np.float64)
    for ids, r in parts:
        out[:, np.asarray(ids, dtype=np.int64)] = r
    if np.isnan(out).any():
        raise RuntimeError("sharded result does not cover every node")
    return out


def run_sharded(states: Sequence, exchange, batch: int = 8):
    """Single-process driver used by tests: all shards in `states`.  -> (rank [K][N], iters [K])."""
    st = iterate(states, exchange, batch=batch)
    parts = [s.read_local() for s in states]
    n_nodes = states[0].n_nodes if hasattr(states[0], "n_nodes") else states[0].g.n
    return assemble(parts, n_nodes, states[0].k), np.asarray(st["iters"], dtype=np.int32)


def gather_ranks(state, group=None):
    """One process per GPU: every rank contributes its rows; all ranks get rank [K][N]."""
    import torch
    import torch.distributed as dist
    ids, r = state.read_local()
    world = dist.get_world_size(group)
    n_nodes = state.n_nodes if hasattr(state, "n_nodes") else state.g.n
    objs: List = [None] * world
    dist.all_gather_object(objs, (np.asarray(ids), np.asarray(r)), group=group)
    return assemble(objs, n_nodes, state.k)
