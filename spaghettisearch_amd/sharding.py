"""Multi-GPU host logic: doc-range-sharded PageRank, and the doc-range-sharded index
(TF-IDF build + scoring) further down.

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

    def start(self):
        """Begin the all-gather without waiting for it: work queued afterwards on the current stream (another
        state's sweep) overlaps with the collective.  -> handle for finish().  The host-staged rehearsal path has
        nothing to overlap and completes here."""
        if self.host_staged:
            self()
            return None
        return self.dist.all_gather_into_tensor(self.recv, self.send, group=self.group, async_op=True)

    @staticmethod
    def finish(handle) -> None:
        """Make the current stream (RCCL) or the host (gloo) wait for a started all-gather."""
        if handle is not None:
            handle.wait()


class LibExchange:
    """The production exchange: RCCL inside the library (ss_pr_exchange on the context's stream) — no Python, torch or
    second communication stack on the data path.  `state` is an engine.PageRankState whose context has a communicator
    (init_lib_comm).  allreduce=True runs the all-reduce form the north star names instead of the all-gather."""

    def __init__(self, state, allreduce: bool = False):
        self.state, self.allreduce = state, allreduce

    def __call__(self) -> None:
        self.state.exchange(self.allreduce)

    def start(self):
        self.state.exchange(self.allreduce)     # enqueued on the context's stream: later work on that stream is ordered behind it
        return None

    @staticmethod
    def finish(handle) -> None:
        return None


def init_lib_comm(ctx, rank: int, world: int, group=None) -> None:
    """Give `ctx` (engine.Context) its in-library communicator: rank 0 makes the id, torch.distributed only carries the
    128 bytes to the other ranks (any channel would do — the Go shim uses a file)."""
    import torch.distributed as dist
    box = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    ctx.comm_init(box[0], rank, world)


def topic_group_layout(rank: int, world: int, groups: int, k_topics: int):
    """The 2-D decomposition of the sharded PageRank (include/spaghetti_rank.h: ss_comm_split): `groups` topic groups x
    world/groups doc shards.  Rank r belongs to topic group `color` = r % groups and is doc shard `key` = r // groups of it
    (so that the shards of one group sit `groups` ranks apart: with 8 ranks and 2 groups, group 0 = ranks 0, 2, 4, 6);
    its group runs topics [lo, hi) on a graph sharded world/groups ways.
    -> (color, key, shards, lo, hi).  bench.py --gpus N and tests/test_gpu_world8.py take the layout from here."""
    if groups < 1 or world % groups or k_topics % groups:
        raise ValueError(f"{groups} topic groups do not divide world {world} / {k_topics} topics")
    color, key, shards = rank % groups, rank // groups, world // groups
    per = k_topics // groups
    return color, key, shards, color * per, (color + 1) * per


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


def sweep_pipelined(states: Sequence, exchanges: Sequence, handles: list, n_sweeps: int) -> None:
    """Topic-block pipelining of the doc-range-sharded sweep: `states` are this rank's shard states of disjoint
    topic blocks over the same graph; while block b's contribution slices travel, block b+1 is finalized and swept.
    `handles` holds one in-flight all-gather per block (see prime_pipelined) and is updated in place."""
    for _ in range(n_sweeps):
        for b, (st, ex) in enumerate(zip(states, exchanges)):
            ex.finish(handles[b])
            st.finalize()
            st.step(1)
            handles[b] = ex.start()


def prime_pipelined(states: Sequence, exchanges: Sequence) -> list:
    """begin() every block and leave its first all-gather in flight."""
    handles = []
    for st, ex in zip(states, exchanges):
        st.begin()
        handles.append(ex.start())
    return handles


def drain_pipelined(states: Sequence, exchanges: Sequence, handles: list) -> None:
    """Complete the in-flight all-gathers and apply them (the states are then consistent, as after iterate())."""
    for b, (st, ex) in enumerate(zip(states, exchanges)):
        ex.finish(handles[b])
        handles[b] = None
        st.finalize()


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


# --------------------------------------------------------------------------------------------------
# Doc-range-sharded inverted index: TF-IDF build and scoring.
#
# Rank r holds, for EVERY term, the slice of the posting list whose docs fall into its contiguous doc range
# [lo_r, hi_r), with local doc ids (doc - lo_r), plus its slice of the magnitudes and of the PageRank prior.
#   * build: idf needs the length of a term's WHOLE list (term_weighting.go:37) -> one all-reduce(sum) of the
#     local list lengths (int64[T]) -> ss_index_set_doc_freq; weights and magnitudes are then shard-local
#     (all postings of a doc live in its shard), no further exchange;
#   * query: the batch is replicated, every rank returns its own top-k (ss_score_topk), ONE all-gather of the
#     n_q*k hits (+ the counts) and ss_merge_hits on every rank give the corpus top-k — identical to the
#     unsharded result, because the k best of the union of per-shard top-k lists are the k best overall.
# The alternative for an index that fits one GPU — query-split replicas, no collective — is what bench.py's
# primary top-k number uses; both are reported.


def doc_range(n_docs: int, rank: int, world: int):
    """Contiguous doc range [lo, hi) of shard `rank`."""
    return rank * n_docs // world, (rank + 1) * n_docs // world


def _xp(a):
    if type(a).__module__.startswith("torch"):
        import torch
        return torch
    return np


def shard_index_by_docs(term_ptr, post_doc, post_tf, lo: int, hi: int, pos_ptr=None, pos=None):
    """Restrict a term-major CSR table to the docs [lo, hi): -> (term_ptr, post_doc - lo, post_tf[, pos_ptr, pos]).
    numpy arrays, or torch tensors (int64/int32 bit-views of the u64/u32 arrays) which stay on their device.
    Pure data movement (select + prefix sum); lists stay ascending by doc."""
    xp = _xp(post_doc)
    if xp is np:
        doc = np.asarray(post_doc).astype(np.int64)
        mask = (doc >= lo) & (doc < hi)
        csum = np.concatenate([[0], np.cumsum(mask, dtype=np.int64)])
        new_ptr = csum[np.asarray(term_ptr).astype(np.int64)].astype(np.uint64)
        out = [new_ptr, (doc[mask] - lo).astype(np.uint32), np.asarray(post_tf)[mask]]
        if pos_ptr is not None:
            pp = np.asarray(pos_ptr).astype(np.int64)
            lens = np.diff(pp)
            out += [np.concatenate([[0], np.cumsum(lens[mask])]).astype(np.uint64), np.asarray(pos)[np.repeat(mask, lens)]]
        return tuple(out)
    import torch
    if hi > 2 ** 31:
        raise ValueError("torch path holds doc ids in int32 views: n_docs must stay below 2^31")
    mask = (post_doc >= lo) & (post_doc < hi)
    csum = torch.cumsum(mask, 0, dtype=torch.int64)
    tp = term_ptr.to(torch.int64)
    new_ptr = torch.where(tp > 0, csum[(tp - 1).clamp(min=0)], torch.zeros_like(tp)) if csum.numel() else torch.zeros_like(tp)
    del csum
    out = [new_ptr, (post_doc[mask] - lo).to(torch.int32), post_tf[mask]]
    if pos_ptr is not None:
        lens = pos_ptr[1:] - pos_ptr[:-1]
        zero = torch.zeros(1, dtype=torch.int64, device=lens.device)
        out += [torch.cat([zero, torch.cumsum(lens[mask], 0)]), pos[torch.repeat_interleave(mask, lens)]]
    return tuple(out)


def local_doc_freq(term_ptr):
    """Local list lengths (int64[T]) of a shard's term_ptr."""
    xp = _xp(term_ptr)
    if xp is np:
        return np.diff(np.asarray(term_ptr).astype(np.int64))
    return (term_ptr[1:] - term_ptr[:-1]).to(xp.int64)


def global_doc_freq(term_ptr, group=None):
    """Whole-corpus document frequencies: all-reduce(sum) of the shards' local list lengths.
    -> uint64 numpy array, or an int64 torch tensor on the input's device."""
    import torch
    import torch.distributed as dist
    df = local_doc_freq(term_ptr)
    if _xp(df) is np:
        t = torch.from_numpy(np.ascontiguousarray(df))
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy().astype(np.uint64)
    if df.is_cuda and dist.get_backend(group) != "nccl":         # rehearsal: device data, gloo collective
        h = df.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        return h.to(df.device)
    dist.all_reduce(df, op=dist.ReduceOp.SUM, group=group)
    return df


HIT_BYTES = 40      # sizeof(ss_hit)


class DocShardedScorer:
    """One doc-range shard per process: replicate the query batch, score locally, all-gather, merge.

    `scorer`  local shard scorer: score_topk(q_ptr, q_terms, k, query_len=, topic_probs=[, out=]) with LOCAL doc ids
              (engine.Scorer in the product);
    `merge`   callable(parts, n_hits [world][n_q], k, doc_base) -> (hits, n_hits) (engine.Context.merge_hits);
    `device`  torch device of the exchange buffers (None: host buffers, the gloo tests);
    host_staged: device results, gloo collective (single-GPU rehearsal only).
    """

    def __init__(self, scorer, merge, n_docs: int, rank: int, world: int, device=None, group=None, host_staged: bool = False):
        self.scorer, self.merge = scorer, merge
        self.rank, self.world, self.group = rank, world, group
        self.device, self.host_staged = device, host_staged
        self.doc_base = np.asarray([doc_range(n_docs, r, world)[0] for r in range(world)], dtype=np.uint32)
        self._bufs = None

    def _buffers(self, n_q: int, k: int):
        import torch
        key = (n_q, k)
        if self._bufs is None or self._bufs[0] != key:
            dev = self.device if self.device is not None else "cpu"
            mk = lambda n, dt: torch.empty(n, dtype=dt, device=dev)
            self._bufs = (key, mk(n_q * k * HIT_BYTES, torch.uint8), mk(n_q, torch.int32),
                          mk(self.world * n_q * k * HIT_BYTES, torch.uint8), mk(self.world * n_q, torch.int32))
        return self._bufs[1:]

    def score_topk(self, q_ptr, q_terms, k: int, query_len=None, topic_probs=None, out=None):
        """-> (hits [n_q][k], n_hits [n_q]) over the whole corpus, identical on every rank."""
        import torch
        import torch.distributed as dist
        n_q = int(q_ptr.shape[0]) - 1
        loc_h, loc_n, all_h, all_n = self._buffers(n_q, k)
        if self.device is not None:
            self.scorer.score_topk(q_ptr, q_terms, k, query_len=query_len, topic_probs=topic_probs, out=(loc_h, loc_n))
        else:
            h, n = self.scorer.score_topk(q_ptr, q_terms, k, query_len=query_len, topic_probs=topic_probs)
            loc_h.copy_(torch.from_numpy(np.ascontiguousarray(h).view(np.uint8).reshape(-1)))
            loc_n.copy_(torch.from_numpy(np.ascontiguousarray(n, dtype=np.int32)))
        if self.host_staged:
            hh, hn = loc_h.cpu(), loc_n.cpu()
            ah, an = torch.empty(all_h.shape, dtype=torch.uint8), torch.empty(all_n.shape, dtype=torch.int32)
            dist.all_gather_into_tensor(ah, hh, group=self.group)
            dist.all_gather_into_tensor(an, hn, group=self.group)
            all_h.copy_(ah)
            all_n.copy_(an)
        else:
            dist.all_gather_into_tensor(all_h, loc_h, group=self.group)
            dist.all_gather_into_tensor(all_n, loc_n, group=self.group)
        if self.device is not None:
            return self.merge(all_h, all_n.view(self.world, n_q), k, self.doc_base, out=out)
        parts = all_h.numpy().view(_hit_dtype()).reshape(self.world, n_q, k)
        return self.merge(parts, all_n.numpy().reshape(self.world, n_q), k, self.doc_base)


def _hit_dtype():
    return np.dtype([("doc", "<u4"), ("_pad", "<u4"), ("title", "<f8"), ("body", "<f8"), ("pagerank", "<f8"), ("final", "<f8")])
