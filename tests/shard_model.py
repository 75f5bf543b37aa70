"""CPU model of ONE shard of the doc-range-sharded PageRank (test infrastructure).

It follows the same layout and exchange protocol as the HIP library
(spaghettisearch_amd/csrc/graph.hip + pagerank.hip): nodes ordered
[non-dangling | dangling], in-degree descending inside a class, dealt round-robin
to the ranks; every rank's all-gather piece = its non-dangling contribution rows +
two tail rows (contribution sum, L1 delta).  Used by the gloo world_size-2 test to
drive spaghettisearch_amd.sharding exactly the way the GPU path does.
"""
import numpy as np
import torch


TAIL_SUM_ROWS = 32


class NumpyShardState:
    def __init__(self, n_nodes, out_ptr, out_dst, d, eps, n_topic, rank, world, max_iter=0):
        self.n_nodes, self.rank, self.world = int(n_nodes), rank, world
        self.d, self.eps, self.max_iter = d, eps, max_iter
        self.x0 = 1.0 / np.asarray(n_topic, dtype=np.float64)
        self.k = len(self.x0)
        out_ptr = np.asarray(out_ptr, dtype=np.int64)
        out_dst = np.asarray(out_dst, dtype=np.int64)
        N = self.n_nodes
        outdeg = np.diff(out_ptr)
        indeg = np.bincount(out_dst, minlength=N)
        cls = (outdeg == 0).astype(np.int64)
        order = np.lexsort((np.arange(N), -indeg, cls))            # class, in-degree desc, id asc
        n_nd = int((outdeg > 0).sum())
        # two tail rows (contribution sum, L1 delta) behind TAIL_SUM_ROWS spare rows (csrc/graph.hpp: the two-vector form's per-topic sums)
        tails = 2 + TAIL_SUM_ROWS if world > 1 else 0
        self.sl_nd = -(-n_nd // world) + tails
        nd_sorted, d_sorted = order[:n_nd], order[n_nd:]
        # internal (table) id of every non-dangling node
        tab_id = np.full(N, -1, dtype=np.int64)
        i = np.arange(n_nd)
        tab_id[nd_sorted] = (i % world) * self.sl_nd + i // world
        self.own_nd = nd_sorted[rank::world]
        self.own_d = d_sorted[rank::world]
        self.own = np.concatenate([self.own_nd, self.own_d])
        self.outdeg_nd = outdeg[self.own_nd].astype(np.float64)
        # in-edges of the own rows, sources as table ids
        src = np.repeat(np.arange(N), outdeg)
        pos = np.full(N, -1, dtype=np.int64)
        pos[self.own] = np.arange(len(self.own))
        m = pos[out_dst] >= 0
        self.e_row = pos[out_dst[m]]
        self.e_src = tab_id[src[m]]
        assert (self.e_src >= 0).all()
        self.send = np.zeros((self.sl_nd, self.k))
        self.table = np.zeros((world * self.sl_nd, self.k))
        self.x = np.zeros((len(self.own), self.k))
        self.S = np.ones(self.k)
        self.active = np.ones(self.k, dtype=bool)
        self.iters = np.zeros(self.k, dtype=np.int32)
        self.delta = np.zeros(self.k)
        self.sweep = 0
        self._begin_pending = False

    # --- state protocol -------------------------------------------------------
    def exchange_tensors(self):
        return torch.from_numpy(self.send.reshape(-1)), torch.from_numpy(self.table.reshape(-1))

    def begin(self):
        self.x[:] = self.x0
        c = self.d * self.x[:len(self.own_nd)] / self.outdeg_nd[:, None]
        self._publish(c, np.zeros(self.k))
        self._begin_pending = True

    def step(self, n=1):
        assert n == 1
        if not self.active.any():
            return
        y = np.zeros_like(self.x)
        np.add.at(y, self.e_row, self.table[self.e_src])
        if self.sweep == 0:
            y += self.x0                                            # Q4
        xn = (y + (1.0 - self.d)) / self.S
        xn[:, ~self.active] = self.x[:, ~self.active]
        dl = np.abs(xn - self.x).sum(axis=0)
        dl[~self.active] = 0.0
        self.x = xn
        c = self.d * xn[:len(self.own_nd)] / self.outdeg_nd[:, None]
        self._publish(c, dl)

    def _publish(self, c, dl):
        self.send[:len(self.own_nd)] = c
        if self.world > 1:
            self.send[self.sl_nd - 2] = c.sum(axis=0)
            self.send[self.sl_nd - 1] = dl
        else:
            self.table[:] = self.send
            self._local = (c.sum(axis=0), dl)

    def finalize(self):
        if self.world > 1:
            t = self.table.reshape(self.world, self.sl_nd, self.k)
            cs, dl = t[:, -2].sum(axis=0), t[:, -1].sum(axis=0)
        else:
            cs, dl = self._local
        tele_n = (1.0 - self.d) * self.n_nodes
        if self._begin_pending:
            self.S = cs + tele_n
            self._begin_pending = False
            return
        if not self.active.any():
            return
        it = self.sweep + 1
        for k in range(self.k):
            if self.active[k]:
                self.iters[k] = it
                self.delta[k] = dl[k]
                cont = dl[k] > self.eps
                if self.max_iter > 0 and it >= self.max_iter:
                    cont = False
                self.active[k] = cont
                self.S[k] = cs[k] + tele_n
        self.sweep = it

    def status(self):
        return {"iters": self.iters.copy(), "n_active": int(self.active.sum()), "sweeps": self.sweep,
                "delta": self.delta.copy(), "total": self.S.copy()}

    def read_local(self):
        return self.own.astype(np.uint32), np.ascontiguousarray(self.x.T)


# ------------------------------------------------------------------------------------------------
# CPU model of ONE shard of the TWO-VECTOR form on doc-range shards (library option "pr.affine", csrc/pagerank.hip
# run_affine_sharded): every topic's ranks are (p*u_k + q) / (r*u_k + s) with u_k = 1/n_k; the shards exchange the 2-wide
# contribution slices and, per iteration, their K local L1 sums.  Same layout as NumpyShardState (k = 2, start vector (1, 0)).

class NumpyAffineShard(NumpyShardState):
    def __init__(self, n_nodes, out_ptr, out_dst, d, eps, n_topic, rank, world, max_iter=0):
        super().__init__(n_nodes, out_ptr, out_dst, d, -1.0, [1, 1], rank, world, max_iter=0)
        self.x0 = np.array([1.0, 0.0])
        self.u = 1.0 / np.asarray(n_topic, dtype=np.float64)
        self.K = len(self.u)
        self.eps_t, self.max_iter_t = eps, max_iter
        self.t_active = np.ones(self.K, dtype=bool)
        self.t_iters = np.zeros(self.K, dtype=np.int32)
        self.out = np.zeros((self.K, len(self.own)))
        self.tele = np.array([0.0, 1.0 - d])
        self.r_x, self.s_x, self.r_next, self.s_next = 0.0, 1.0, 0.0, 1.0

    def _gather(self, send, recv):
        import torch.distributed as dist
        parts = [torch.empty_like(send) for _ in range(self.world)]
        dist.all_gather(parts, send)
        recv.copy_(torch.cat(parts))

    def _exchange_table(self):
        snd, tab = self.exchange_tensors()
        self._gather(snd, tab)

    def _sums(self):
        t = self.table.reshape(self.world, self.sl_nd, self.k)
        return t[:, -2].sum(axis=0)

    def run(self, lag=True):
        """lag (the library's default, option pr.affine_lag): ONE exchange per iteration — the per-topic L1 sums of iteration i ride in
        the spare tail rows of iteration i + 1's slice, the stop decisions of iteration i are taken one exchange late and the ranks
        of a topic that stops are taken from the vectors of the iteration it stopped in; after the last sweep of a max_iter run the
        sums get a small exchange of their own.  lag=False: the round-4 protocol (a second, small exchange per iteration)."""
        tele_n = (1.0 - self.d) * self.n_nodes
        tau = 1.0 - self.d
        assert self.world == 1 or self.K <= 2 * TAIL_SUM_ROWS
        sums_lo = self.sl_nd - 2 - TAIL_SUM_ROWS

        def decide(tot, it, new):
            for k in range(self.K):
                if self.t_active[k]:
                    self.t_iters[k] = it
                    cont = tot[k] > self.eps_t
                    if self.max_iter_t > 0 and it >= self.max_iter_t:
                        cont = False
                    if not cont:
                        self.t_active[k] = False
                        self.out[k] = new[:, k]

        def gather_small(loc):
            allv = torch.empty(self.world * self.K, dtype=torch.float64)
            self._gather(torch.from_numpy(loc), allv)
            dl = allv.reshape(self.world, self.K).numpy()
            tot = np.zeros(self.K)
            for r in range(self.world):
                tot += dl[r]
            return tot

        # begin
        self.x[:] = self.x0
        c = self.d * self.x[:len(self.own_nd)] / self.outdeg_nd[:, None]
        self._publish(c, np.zeros(self.k))
        self._exchange_table()
        cs = self._sums()
        r1, s1 = cs[0], cs[1] + tele_n
        sigma = r1 + s1
        self.r_next, self.s_next = r1 / sigma, s1 / sigma
        it = 0
        pending = None          # (ranks of the vectors of the iteration whose sums are on their way)
        while self.t_active.any():
            y = np.zeros_like(self.x)
            np.add.at(y, self.e_row, self.table[self.e_src])
            if it == 0:
                y += self.x0
            xo = self.x
            xn = (y + self.tele) / sigma
            self.x = xn
            c = self.d * xn[:len(self.own_nd)] / self.outdeg_nd[:, None]
            self._publish(c, np.zeros(self.k))
            self._exchange_table()                                # (lag: carries the previous iteration's per-topic sums)
            cs = self._sums()
            r_prev, s_prev = self.r_x, self.s_x
            self.r_x, self.s_x = self.r_next, self.s_next
            r1, s1 = cs[0] + tele_n * self.r_x, cs[1] + tele_n * self.s_x
            sigma = r1 + s1
            self.r_next, self.s_next = r1 / sigma, s1 / sigma
            self.tele = np.array([tau * self.r_x, tau * self.s_x])
            new = (xn[:, :1] * self.u + xn[:, 1:2]) / (self.r_x * self.u + self.s_x)
            old = (xo[:, :1] * self.u + xo[:, 1:2]) / (r_prev * self.u + s_prev)
            loc = np.abs(new - old).sum(axis=0)
            if lag and self.world > 1:
                if pending is not None:
                    t = self.table.reshape(self.world, self.sl_nd * self.k)
                    tot = np.zeros(self.K)
                    for r in range(self.world):                   # rank order
                        tot += t[r, sums_lo * self.k: sums_lo * self.k + self.K]
                    decide(tot, it, pending)
                flat = self.send.reshape(-1)
                flat[sums_lo * self.k: sums_lo * self.k + self.K] = loc      # rides with the next slice
                pending = new
                it += 1
                if self.max_iter_t > 0 and it >= self.max_iter_t and self.t_active.any():
                    decide(gather_small(loc), it, new)            # the flush after the last sweep
            else:
                it += 1
                decide(gather_small(loc), it, new)
        return self.own.astype(np.uint32), self.out, self.t_iters


# ------------------------------------------------------------------------------------------------
# CPU model of ONE doc-range shard of the inverted index (test infrastructure): the oracle's arithmetic
# applied to the shard's slice, with the idf taken from whole-corpus document frequencies — what
# ss_index_set_doc_freq + ss_tfidf_build + ss_score_topk do on the GPU.

class CpuIndexShard:
    def __init__(self, n_docs_local, title, body, df_title, df_body, total_docs):
        from oracle import pyoracle
        self.n = int(n_docs_local)
        self.tables = []
        self.mags = []
        for (ptr, doc, tf), df in ((title, df_title), (body, df_body)):
            ptr = np.asarray(ptr, dtype=np.uint64)
            idf = np.array([np.float32(pyoracle.go_log2(float(total_docs) / float(x))) if x else np.float32(np.inf)
                            for x in np.asarray(df, dtype=np.uint64)], dtype=np.float32)          # term_weighting.go:37
            w = (np.asarray(tf, dtype=np.float32) * np.repeat(idf, np.diff(ptr.astype(np.int64)))).astype(np.float32)   # :42
            sq = (w * w).astype(np.float32).astype(np.float64)                                   # :44
            mag2 = np.zeros(self.n)
            np.add.at(mag2, np.asarray(doc, dtype=np.int64), sq)                                 # posting order, like the Go loop
            self.tables.append((ptr, np.asarray(doc, dtype=np.uint32), w))
            self.mags.append(np.sqrt(mag2))
        self.prior = None

    def set_prior(self, rank_local):
        """rank_local [K][n_local] topic-major -> node-major like forw[3] rows."""
        self.prior = None if rank_local is None else np.ascontiguousarray(np.asarray(rank_local).T)

    def score_topk(self, q_ptr, q_terms, k, query_len=None, topic_probs=None):
        from oracle import pyoracle
        return pyoracle.score_topk_batch(self.n, self.tables[0], self.tables[1], self.mags[0], self.mags[1], q_ptr, q_terms, k,
                                         prior=self.prior, topic_probs=topic_probs, query_len=query_len)


def merge_hits_model(parts, n_hits, k, doc_base):
    """numpy restatement of the merge order: final descending, equal finals by ascending corpus doc id, NaN last."""
    n_parts, n_q = n_hits.shape
    out = np.zeros((n_q, k), dtype=parts.dtype)
    n_out = np.zeros(n_q, dtype=np.int32)
    for q in range(n_q):
        rows = []
        for p in range(n_parts):
            r = parts[p, q, :n_hits[p, q]].copy()
            r["doc"] += np.uint32(doc_base[p])
            rows.append(r)
        allr = np.concatenate(rows)
        nan = np.isnan(allr["final"])
        order = np.lexsort((allr["doc"], -np.where(nan, 0.0, allr["final"]), nan))
        allr = allr[order][:k]
        out[q, :len(allr)] = allr
        n_out[q] = len(allr)
    return out, n_out
