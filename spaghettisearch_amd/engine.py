"""Thin object wrappers over the C ABI (include/spaghetti_rank.h).

Arrays may be numpy arrays (host) or torch tensors (host or device): the library
copies with hipMemcpyDefault, so both kinds of pointer are accepted.  Every call
goes through libspaghetti_rank.so — there is no Python/CPU implementation here.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import SsGraphInfo, SsHit, check

HIT_DTYPE = np.dtype([("doc", "<u4"), ("_pad", "<u4"), ("title", "<f8"), ("body", "<f8"),
                      ("pagerank", "<f8"), ("final", "<f8")])

_NP2T = {"uint64": "int64", "uint32": "int32"}


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _as(x, dtype: str):
    """Contiguous array of the wanted dtype (numpy or torch), no copy when already right.
    torch has no uint32/uint64 storage: int32/int64 tensors of the same bits are accepted."""
    if x is None:
        return None
    if _is_torch(x):
        import torch
        want = getattr(torch, _NP2T.get(dtype, dtype))
        if x.dtype != want:
            raise TypeError(f"torch tensor has dtype {x.dtype}, expected {want}")
        return x.contiguous()
    return np.ascontiguousarray(x, dtype=np.dtype(dtype))


def _ptr(x) -> Optional[int]:
    if x is None:
        return None
    if _is_torch(x):
        return x.data_ptr()
    return x.ctypes.data


class Context:
    """One per GPU per process (ss_init)."""

    def __init__(self, device_id: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.ss_init(device_id, C.byref(h)))
        self.h = h
        self.device_id = device_id
        self._options = {}           # name -> value of the options set through this object (options() restores from it)

    def set_stream(self, hip_stream: Optional[int]) -> None:
        check(self.lib.ss_set_stream(self.h, C.c_void_p(hip_stream)), self.h)
        self._shared_stream = bool(hip_stream)

    def ready(self, *arrays) -> None:
        """Device arrays handed to the library must hold their final contents with respect to the context's stream.
        While the context runs on its own (non-blocking) stream, wait for torch's current stream if any argument is a
        CUDA tensor; with a shared stream (set_stream) ordering is the stream's and nothing is done."""
        if getattr(self, "_shared_stream", False):
            return
        for a in arrays:
            if a is not None and _is_torch(a) and a.is_cuda:
                import torch
                torch.cuda.current_stream(a.device).synchronize()
                return

    def merge_hits(self, parts, n_hits, k: int, doc_base=None, out=None):
        """ss_merge_hits: parts [n_parts][n_q][k] hits (numpy HIT_DTYPE, or a torch uint8 tensor of the same bytes),
        n_hits [n_parts][n_q] int32, doc_base uint32[n_parts] | None -> (hits [n_q][k], n_hits [n_q]).
        out=(hits_buf, n_hits_buf) keeps the result in the given (device) buffers."""
        n_hits = _as(n_hits, "int32")
        n_parts, n_q = int(n_hits.shape[0]), int(n_hits.shape[1])
        doc_base = _as(doc_base, "uint32")
        if _is_torch(parts):
            parts = parts.contiguous()
            have = parts.numel() * parts.element_size()
        else:
            parts = np.ascontiguousarray(parts, dtype=HIT_DTYPE)
            have = parts.nbytes
        if have != n_parts * n_q * k * HIT_DTYPE.itemsize:
            raise ValueError("parts does not hold n_parts*n_q*k hits")
        if out is not None:
            hits, n_out = out
            if hits.numel() * hits.element_size() < n_q * k * HIT_DTYPE.itemsize or n_out.numel() < n_q:
                raise ValueError("output buffers too small")
        else:
            hits = np.zeros((n_q, k), dtype=HIT_DTYPE)
            n_out = np.zeros(n_q, dtype=np.int32)
        self.ready(parts, n_hits, doc_base)
        check(self.lib.ss_merge_hits(self.h, n_q, n_parts, k, _ptr(parts), _ptr(n_hits), _ptr(doc_base), _ptr(hits), _ptr(n_out)), self.h)
        return hits, n_out

    def synchronize(self) -> None:
        check(self.lib.ss_synchronize(self.h), self.h)

    def set_option(self, name: str, value: Optional[int]) -> None:
        """ss_set_option: tuning / diagnostic switch of this context; None restores the default."""
        check(self.lib.ss_set_option(self.h, name.encode(), -(1 << 63) if value is None else int(value)), self.h)
        if value is None:
            self._options.pop(name, None)
        else:
            self._options[name] = int(value)

    def options(self, **kv):
        """Context manager: set options (dots written as double underscores: pr__force_narrow=1); on exit every option gets back
        the value it had when the scope was entered (the default if it had none), so scopes nest."""
        ctx = self

        class _Scope:
            def __enter__(self_inner):
                self_inner.before = {k.replace("__", "."): ctx._options.get(k.replace("__", ".")) for k in kv}
                for k, v in kv.items():
                    ctx.set_option(k.replace("__", "."), v)

            def __exit__(self_inner, *exc):
                for name, old in self_inner.before.items():
                    ctx.set_option(name, old)
                return False
        return _Scope()

    # ---- in-library collectives (RCCL over xGMI): one context per rank
    @staticmethod
    def comm_unique_id() -> bytes:
        """ss_comm_unique_id: rank 0 creates the 128-byte id, the host hands it to every rank."""
        buf = C.create_string_buffer(_lib.SS_COMM_ID_BYTES)
        check(_lib.load().ss_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid: bytes, rank: int, world: int) -> None:
        if len(uid) != _lib.SS_COMM_ID_BYTES:
            raise ValueError("communicator id must be SS_COMM_ID_BYTES long")
        check(self.lib.ss_comm_init(self.h, C.c_char_p(uid), rank, world), self.h)

    def comm_split(self, color: int, key: int) -> None:
        """ss_comm_split: topic groups x doc shards; the context's communicator becomes its group's."""
        check(self.lib.ss_comm_split(self.h, color, key), self.h)

    def comm_destroy(self) -> None:
        check(self.lib.ss_comm_destroy(self.h), self.h)

    def comm_info(self):
        r, w = C.c_int32(), C.c_int32()
        check(self.lib.ss_comm_info(self.h, C.byref(r), C.byref(w)), self.h)
        return r.value, w.value

    def comm_allreduce_u64(self, buf) -> None:
        """In-place all-reduce(sum) of a uint64 array (numpy host array or torch int64 device tensor)."""
        buf = _as(buf, "uint64")
        self.ready(buf)
        check(self.lib.ss_comm_allreduce_u64(self.h, _ptr(buf), int(buf.shape[0])), self.h)

    def comm_allgather(self, send, recv, bytes_per_rank: int) -> None:
        self.ready(send, recv)
        check(self.lib.ss_comm_allgather(self.h, _ptr(send), _ptr(recv), int(bytes_per_rank)), self.h)

    def last_kernel_ms(self, kind: int) -> float:
        ms = C.c_float(0)
        check(self.lib.ss_last_kernel_ms(self.h, kind, C.byref(ms)), self.h)
        return float(ms.value)

    def close(self) -> None:
        if self.h:
            self.lib.ss_shutdown(self.h)
            self.h = None


class Graph:
    """Link graph (ranking/pagerank.go:17-44) in the library's HBM layout."""

    def __init__(self, ctx: Context, n_nodes: int, out_ptr, out_dst, rank: int = 0, world: int = 1):
        self.ctx = ctx
        self.n = int(n_nodes)
        out_ptr = _as(out_ptr, "uint64")
        out_dst = _as(out_dst, "uint32")
        self.e = int(out_dst.shape[0]) if out_dst is not None else 0
        if out_ptr.shape[0] != self.n + 1:
            raise ValueError("out_ptr must have n_nodes+1 entries")
        ctx.ready(out_ptr, out_dst)
        h = C.c_void_p()
        check(ctx.lib.ss_graph_create(ctx.h, self.n, self.e, _ptr(out_ptr), _ptr(out_dst), rank, world, C.byref(h)), ctx.h)
        self.h = h
        self.rank, self.world = rank, world

    def apply_delta(self, n_nodes_new: int, changed, new_ptr, new_children) -> None:
        """ss_graph_apply_delta: replace the child lists of the `changed` parents (and admit nodes up to n_nodes_new)."""
        changed = _as(changed, "uint32")
        new_ptr = _as(new_ptr, "uint64")
        new_children = _as(new_children, "uint32")
        self.ctx.ready(changed, new_ptr, new_children)
        check(self.ctx.lib.ss_graph_apply_delta(self.h, int(n_nodes_new), int(changed.shape[0]), _ptr(changed), _ptr(new_ptr), _ptr(new_children)),
              self.ctx.h)
        self.n = int(n_nodes_new)

    def info(self) -> SsGraphInfo:
        gi = SsGraphInfo()
        check(self.ctx.lib.ss_graph_get_info(self.h, C.byref(gi)), self.ctx.h)
        return gi

    def pagerank_sharded(self, damping: float, eps: float, n_topic: Sequence[int], max_iter: int = 0, allreduce: bool = False, out=None):
        """ss_pagerank_run_sharded (this rank's part of the doc-range-sharded loop, collectives inside the library):
        -> (ids uint32[rows], rank [K][rows] float64, iters [K] int32).  out = (ids, rank) device tensors keeps the result
        in HBM (torch int32 [rows], float64 [K][rows])."""
        n_topic = np.ascontiguousarray(np.atleast_1d(n_topic), dtype=np.int32)
        K = len(n_topic)
        rows = int(self.info().n_rows_local)
        if out is not None:
            ids, rank = _as(out[0], "uint32"), _as(out[1], "float64")
            count = lambda a: a.numel() if _is_torch(a) else a.size
            if count(ids) < rows or count(rank) < K * rows:
                raise ValueError("out arrays too small")
        else:
            ids = np.zeros(rows, dtype=np.uint32)
            rank = np.zeros((K, rows), dtype=np.float64)
        iters = np.zeros(K, dtype=np.int32)
        check(self.ctx.lib.ss_pagerank_run_sharded(self.h, damping, eps, max_iter, K, _ptr(n_topic), 1 if allreduce else 0,
                                                   _ptr(ids), _ptr(rank), _ptr(iters)), self.ctx.h)
        return ids, rank, iters

    def pagerank(self, damping: float, eps: float, n_topic: Sequence[int], max_iter: int = 0):
        """ss_pagerank_run: -> (rank [K][N] float64, iters [K] int32)."""
        n_topic = np.ascontiguousarray(np.atleast_1d(n_topic), dtype=np.int32)
        K = len(n_topic)
        rank = np.zeros((K, self.n), dtype=np.float64)
        iters = np.zeros(K, dtype=np.int32)
        check(self.ctx.lib.ss_pagerank_run(self.h, damping, eps, max_iter, K, _ptr(n_topic), _ptr(rank), _ptr(iters)),
              self.ctx.h)
        return rank, iters

    @staticmethod
    def pagerank_group(shards, damping: float, eps: float, n_topic: Sequence[int], max_iter: int = 0):
        """ss_pagerank_run_group: all `world` shards of one graph on one context, the library's pipelined (topic-blocked,
        two-stream) sharded loop with in-process copies as the exchange -> (rank [K][N], iters [K])."""
        n_topic = np.ascontiguousarray(np.atleast_1d(n_topic), dtype=np.int32)
        K = len(n_topic)
        ctx = shards[0].ctx
        hs = (C.c_void_p * len(shards))(*[g.h for g in shards])
        rank = np.zeros((K, shards[0].n), dtype=np.float64)
        iters = np.zeros(K, dtype=np.int32)
        check(ctx.lib.ss_pagerank_run_group(hs, len(shards), damping, eps, max_iter, K, _ptr(n_topic), _ptr(rank), _ptr(iters)), ctx.h)
        return rank, iters

    def pagerank_dev(self, damping: float, eps: float, n_topic: Sequence[int], rank_out, max_iter: int = 0):
        """ss_pagerank_run with the ranks left in device memory (`rank_out`: torch float64 [K][N] on the GPU) -> iters [K]."""
        n_topic = np.ascontiguousarray(np.atleast_1d(n_topic), dtype=np.int32)
        K = len(n_topic)
        rank_out = _as(rank_out, "float64")
        if rank_out.numel() < K * self.n:
            raise ValueError("rank_out too small")
        iters = np.zeros(K, dtype=np.int32)
        check(self.ctx.lib.ss_pagerank_run(self.h, damping, eps, max_iter, K, _ptr(n_topic), _ptr(rank_out), _ptr(iters)),
              self.ctx.h)
        return iters

    def close(self) -> None:
        if self.h:
            self.ctx.lib.ss_graph_destroy(self.h)
            self.h = None


class PageRankState:
    """Step-wise power iteration (ss_pr_*), used for timing and by the multi-GPU host."""

    def __init__(self, graph: Graph, damping: float, eps: float, n_topic: Sequence[int], max_iter: int = 0):
        self.g = graph
        self.ctx = graph.ctx
        n_topic = np.ascontiguousarray(np.atleast_1d(n_topic), dtype=np.int32)
        self.k = len(n_topic)
        h = C.c_void_p()
        check(self.ctx.lib.ss_pr_create(graph.h, damping, eps, max_iter, self.k, _ptr(n_topic), C.byref(h)), self.ctx.h)
        self.h = h

    def set_teleport(self, sets) -> None:
        """ss_pr_set_teleport: `sets` = one array of DISTINCT node ids per topic (empty = uniform teleport for that topic),
        or None to restore the reference's uniform teleport.  Opt-in true topic-sensitive PageRank."""
        if sets is None:
            check(self.ctx.lib.ss_pr_set_teleport(self.h, None, None), self.ctx.h)
            return
        if len(sets) != self.k:
            raise ValueError("one teleport set per topic")
        ptr = np.zeros(self.k + 1, dtype=np.uint64)
        ptr[1:] = np.cumsum([len(x) for x in sets])
        nodes = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.uint32) for x in sets]) if int(ptr[-1]) else np.zeros(0, np.uint32))
        check(self.ctx.lib.ss_pr_set_teleport(self.h, _ptr(ptr), _ptr(nodes)), self.ctx.h)

    def begin(self) -> None:
        check(self.ctx.lib.ss_pr_begin(self.h), self.ctx.h)

    def step(self, n_steps: int = 1) -> None:
        check(self.ctx.lib.ss_pr_step(self.h, n_steps), self.ctx.h)

    def finalize(self) -> None:
        check(self.ctx.lib.ss_pr_finalize(self.h), self.ctx.h)

    def exchange(self, allreduce: bool = False) -> None:
        """ss_pr_exchange: the per-sweep collective inside the library (needs Context.comm_init)."""
        check(self.ctx.lib.ss_pr_exchange(self.h, 1 if allreduce else 0), self.ctx.h)

    def exchange_buffers(self):
        """-> (send_ptr, send_bytes, recv_ptr, recv_bytes) device pointers."""
        sp, rp = C.c_void_p(), C.c_void_p()
        sb, rb = C.c_uint64(), C.c_uint64()
        check(self.ctx.lib.ss_pr_exchange_buffers(self.h, C.byref(sp), C.byref(sb), C.byref(rp), C.byref(rb)), self.ctx.h)
        return sp.value, sb.value, rp.value, rb.value

    def status(self):
        """-> dict(iters, n_active, sweeps, delta, total); waits for the stream."""
        iters = np.zeros(self.k, dtype=np.int32)
        delta = np.zeros(self.k, dtype=np.float64)
        total = np.zeros(self.k, dtype=np.float64)
        na, sw = C.c_int32(), C.c_int32()
        check(self.ctx.lib.ss_pr_status(self.h, _ptr(iters), C.byref(na), C.byref(sw), _ptr(delta), _ptr(total)), self.ctx.h)
        return {"iters": iters, "n_active": na.value, "sweeps": sw.value, "delta": delta, "total": total}

    def probe(self, mode: int = 0, n_reps: int = 5) -> float:
        """ss_pr_probe: ms per gather-only pass over this graph's index stream (diagnostic)."""
        ms = C.c_float(0)
        check(self.ctx.lib.ss_pr_probe(self.h, mode, n_reps, C.byref(ms)), self.ctx.h)
        return float(ms.value)

    def read(self) -> np.ndarray:
        out = np.zeros((self.k, self.g.n), dtype=np.float64)
        check(self.ctx.lib.ss_pr_read(self.h, _ptr(out)), self.ctx.h)
        return out

    def read_local(self):
        n_rows = int(self.g.info().n_rows_local)
        ids = np.zeros(n_rows, dtype=np.uint32)
        out = np.zeros((self.k, n_rows), dtype=np.float64)
        check(self.ctx.lib.ss_pr_read_local(self.h, _ptr(ids), _ptr(out)), self.ctx.h)
        return ids, out

    def close(self) -> None:
        if self.h:
            self.ctx.lib.ss_pr_destroy(self.h)
            self.h = None


class InvertedIndex:
    """One inverted table (inv[0] title / inv[1] body) — ss_index_*."""

    def __init__(self, ctx: Context, n_docs: int, term_ptr, post_doc, post_tf):
        self.ctx = ctx
        self.n_docs = int(n_docs)
        term_ptr = _as(term_ptr, "uint64")
        post_doc = _as(post_doc, "uint32")
        post_tf = _as(post_tf, "float32")
        self.n_terms = int(term_ptr.shape[0]) - 1
        self.n_post = int(post_doc.shape[0])
        ctx.ready(term_ptr, post_doc, post_tf)
        h = C.c_void_p()
        check(ctx.lib.ss_index_create(ctx.h, self.n_docs, self.n_terms, _ptr(term_ptr), _ptr(post_doc), _ptr(post_tf),
                                      C.byref(h)), ctx.h)
        self.h = h

    def tfidf_build(self, total_docs: int, want_w: bool = True, want_mag: bool = True, want_idf: bool = True):
        """ss_tfidf_build (ranking.UpdateTermWeights): -> (w f32[P] | None, mag f64[N] | None, idf f32[T] | None)."""
        w = np.zeros(self.n_post, dtype=np.float32) if want_w else None
        mag = np.zeros(self.n_docs, dtype=np.float64) if want_mag else None
        idf = np.zeros(self.n_terms, dtype=np.float32) if want_idf else None
        check(self.ctx.lib.ss_tfidf_build(self.h, int(total_docs), _ptr(w), _ptr(mag), _ptr(idf)), self.ctx.h)
        return w, mag, idf

    def apply_delta(self, del_docs=None, del_pairs=None, add=None, add_pos=None) -> None:
        """ss_index_apply_delta(_pos): del_docs uint32[], del_pairs = (terms uint32[], docs uint32[]), add = (terms, docs, weights f32[]),
        add_pos = (pos_ptr uint64[n_add+1], pos float32[]) positional postings of the new postings (tables with positions)."""
        dd = _as(del_docs if del_docs is not None else np.zeros(0, np.uint32), "uint32")
        dt, dp = (_as(del_pairs[0], "uint32"), _as(del_pairs[1], "uint32")) if del_pairs is not None else (np.zeros(0, np.uint32),) * 2
        at, ad, aw = (_as(add[0], "uint32"), _as(add[1], "uint32"), _as(add[2], "float32")) if add is not None else \
            (np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32))
        pp, pv = (_as(add_pos[0], "uint64"), _as(add_pos[1], "float32")) if add_pos is not None else (None, None)
        self.ctx.ready(dd, dt, dp, at, ad, aw, pp, pv)
        check(self.ctx.lib.ss_index_apply_delta_pos(self.h, int(dd.shape[0]), _ptr(dd), int(dt.shape[0]), _ptr(dt), _ptr(dp),
                                                    int(at.shape[0]), _ptr(at), _ptr(ad), _ptr(aw), _ptr(pp), _ptr(pv)), self.ctx.h)
        n_post = C.c_uint64()
        check(self.ctx.lib.ss_index_get_info(self.h, None, None, C.byref(n_post)), self.ctx.h)
        self.n_post = int(n_post.value)

    def resize(self, n_docs: int, n_terms: int) -> None:
        """ss_index_resize: grow the doc and / or term space (new words, new child pages of a re-indexed page)."""
        check(self.ctx.lib.ss_index_resize(self.h, int(n_docs), int(n_terms)), self.ctx.h)
        self.n_docs, self.n_terms = int(n_docs), int(n_terms)

    def read_magnitudes(self, docs) -> np.ndarray:
        docs = _as(docs, "uint32")
        out = np.zeros(int(docs.shape[0]), dtype=np.float64)
        check(self.ctx.lib.ss_index_read_magnitudes(self.h, int(docs.shape[0]), _ptr(docs), _ptr(out)), self.ctx.h)
        return out

    def read_positions(self):
        """-> (pos_ptr uint64[P+1], pos float32[]) as they stand (tables with positional postings)."""
        pp = np.zeros(self.n_post + 1, dtype=np.uint64)
        check(self.ctx.lib.ss_index_read_positions(self.h, _ptr(pp), None), self.ctx.h)
        pv = np.zeros(int(pp[-1]), dtype=np.float32)
        check(self.ctx.lib.ss_index_read_positions(self.h, None, _ptr(pv)), self.ctx.h)
        return pp, pv

    def refresh_magnitudes(self) -> np.ndarray:
        mag = np.zeros(self.n_docs, dtype=np.float64)
        check(self.ctx.lib.ss_index_refresh_magnitudes(self.h, _ptr(mag)), self.ctx.h)
        return mag

    def read(self):
        """-> (term_ptr uint64[T+1], post_doc uint32[P], post_w float32[P]) as the table stands."""
        tp = np.zeros(self.n_terms + 1, dtype=np.uint64)
        pd = np.zeros(self.n_post, dtype=np.uint32)
        pw = np.zeros(self.n_post, dtype=np.float32)
        check(self.ctx.lib.ss_index_read(self.h, _ptr(tp), _ptr(pd), _ptr(pw)), self.ctx.h)
        return tp, pd, pw

    def set_doc_freq(self, df) -> None:
        """ss_index_set_doc_freq: whole-corpus document frequencies when this table is one doc-range shard
        (uint64[n_terms]; None = local list lengths).  Call before tfidf_build."""
        df = _as(df, "uint64")
        if df is not None and int(df.shape[0]) != self.n_terms:
            raise ValueError("df must hold one entry per term")
        self.ctx.ready(df)
        check(self.ctx.lib.ss_index_set_doc_freq(self.h, _ptr(df)), self.ctx.h)

    def set_positions(self, pos_ptr, pos) -> None:
        """Positional postings (listPos[1:] per posting) for phrase search."""
        pos_ptr = _as(pos_ptr, "uint64")
        pos = _as(pos, "float32")
        self.ctx.ready(pos_ptr, pos)
        check(self.ctx.lib.ss_index_set_positions(self.h, _ptr(pos_ptr), _ptr(pos)), self.ctx.h)

    def set_weighted(self, mag) -> None:
        mag = _as(mag, "float64")
        self.ctx.ready(mag)
        check(self.ctx.lib.ss_index_set_weighted(self.h, _ptr(mag)), self.ctx.h)

    def close(self) -> None:
        if self.h:
            check(self.ctx.lib.ss_index_destroy(self.h), self.ctx.h)
            self.h = None


class Scorer:
    """Batched OR-query cosine scorer + PageRank blend + top-k (ss_score_topk)."""

    def __init__(self, ctx: Context, title: InvertedIndex, body: InvertedIndex):
        self.ctx = ctx
        self.title, self.body = title, body
        h = C.c_void_p()
        check(ctx.lib.ss_scorer_create(ctx.h, title.h, body.h, C.byref(h)), ctx.h)
        self.h = h
        self.k_topics = 0

    def set_prior(self, rank) -> None:
        """rank [K][n_docs] topic-major (ss_pagerank_run layout) or None to clear."""
        if rank is None:
            check(self.ctx.lib.ss_scorer_set_prior(self.h, 0, None), self.ctx.h)
            self.k_topics = 0
            return
        rank = _as(rank, "float64")
        self.k_topics = int(rank.shape[0])
        self.ctx.ready(rank)
        check(self.ctx.lib.ss_scorer_set_prior(self.h, self.k_topics, _ptr(rank)), self.ctx.h)

    def score_topk_phrase(self, q_ptr, q_terms, p_ptr, p_terms, k: int, query_len=None, topic_probs=None):
        """ss_score_topk_phrase: OR terms + one concatenated quoted phrase per query."""
        q_ptr = _as(q_ptr, "uint32")
        q_terms = _as(q_terms, "uint32")
        p_ptr = _as(p_ptr, "uint32")
        p_terms = _as(p_terms, "uint32")
        query_len = _as(query_len, "int32")
        topic_probs = _as(topic_probs, "float64")
        n_q = int(q_ptr.shape[0]) - 1
        self.ctx.ready(q_ptr, q_terms, p_ptr, p_terms, query_len, topic_probs)
        hits = np.zeros((n_q, k), dtype=HIT_DTYPE)
        n_hits = np.zeros(n_q, dtype=np.int32)
        check(self.ctx.lib.ss_score_topk_phrase(self.h, n_q, _ptr(q_ptr), _ptr(q_terms), _ptr(p_ptr), _ptr(p_terms),
                                                _ptr(query_len), _ptr(topic_probs), k, hits.ctypes.data, _ptr(n_hits)), self.ctx.h)
        return hits, n_hits

    def score_topk(self, q_ptr, q_terms, k: int, query_len=None, topic_probs=None, out=None):
        """-> (hits [n_q][k] HIT_DTYPE, n_hits [n_q] int32).
        out=(hits_buf, n_hits_buf): optional device-resident outputs (torch uint8 tensor of
        n_q*k*40 bytes and int32 tensor of n_q) — results stay in HBM and `out` is returned."""
        q_ptr = _as(q_ptr, "uint32")
        q_terms = _as(q_terms, "uint32")
        query_len = _as(query_len, "int32")
        topic_probs = _as(topic_probs, "float64")
        n_q = int(q_ptr.shape[0]) - 1
        self.ctx.ready(q_ptr, q_terms, query_len, topic_probs)
        if out is not None:
            hits, n_hits = out
            if hits.numel() * hits.element_size() < n_q * k * HIT_DTYPE.itemsize or n_hits.numel() < n_q:
                raise ValueError("output buffers too small")
            check(self.ctx.lib.ss_score_topk(self.h, n_q, _ptr(q_ptr), _ptr(q_terms), _ptr(query_len), _ptr(topic_probs), k,
                                             _ptr(hits), _ptr(n_hits)), self.ctx.h)
            # device outputs: the library only enqueues (results are ordered on the context's stream).  On a stream shared
            # with the caller that is all that is needed; on the context's own stream wait here.
            if not getattr(self.ctx, "_shared_stream", False):
                self.ctx.synchronize()
            return hits, n_hits
        hits = np.zeros((n_q, k), dtype=HIT_DTYPE)
        n_hits = np.zeros(n_q, dtype=np.int32)
        check(self.ctx.lib.ss_score_topk(self.h, n_q, _ptr(q_ptr), _ptr(q_terms), _ptr(query_len), _ptr(topic_probs), k,
                                         hits.ctypes.data, _ptr(n_hits)), self.ctx.h)
        return hits, n_hits

    def submit(self, q_ptr, q_terms, k: int, query_len=None, topic_probs=None, p_ptr=None, p_terms=None):
        """ss_score_topk_submit: enqueue a batch whose hits go to host memory; -> ticket for collect().
        p_ptr / p_terms: the queries' quoted phrases as in score_topk_phrase (None: plain OR queries)."""
        q_ptr = _as(q_ptr, "uint32")
        q_terms = _as(q_terms, "uint32")
        p_ptr = _as(p_ptr, "uint32")
        p_terms = _as(p_terms, "uint32")
        query_len = _as(query_len, "int32")
        topic_probs = _as(topic_probs, "float64")
        n_q = int(q_ptr.shape[0]) - 1
        self.ctx.ready(q_ptr, q_terms, p_ptr, p_terms, query_len, topic_probs)
        ticket = C.c_uint64(0)
        check(self.ctx.lib.ss_score_topk_submit(self.h, n_q, _ptr(q_ptr), _ptr(q_terms), _ptr(p_ptr), _ptr(p_terms), _ptr(query_len),
                                                _ptr(topic_probs), k, C.addressof(ticket)), self.ctx.h)
        return (int(ticket.value), n_q, k)

    def collect(self, ticket, out=None):
        """ss_score_topk_collect: wait for the batch of `ticket` (from submit) -> (hits [n_q][k], n_hits [n_q]) in host memory."""
        t, n_q, k = ticket
        hits, n_hits = out if out is not None else (np.zeros((n_q, k), dtype=HIT_DTYPE), np.zeros(n_q, dtype=np.int32))
        check(self.ctx.lib.ss_score_topk_collect(self.h, t, hits.ctypes.data, n_hits.ctypes.data), self.ctx.h)
        return hits, n_hits

    def close(self) -> None:
        if self.h:
            self.ctx.lib.ss_scorer_destroy(self.h)
            self.h = None


__all__ = ["Context", "Graph", "PageRankState", "InvertedIndex", "Scorer", "HIT_DTYPE", "SsHit"]
