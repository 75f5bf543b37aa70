"""Deterministic synthetic inputs for the ranking hot path.

Nothing here is on the product path: these generators feed tests and bench.py.
numpy versions for CPU-sized cases, torch versions (suffix ``_torch``) generate
the 1M-10M-doc configurations directly in HBM.

  * link graph: R-MAT (a,b,c,d = 0.57,0.19,0.19,0.05), unique (src,dst) pairs
    (the crawler de-duplicates children, crawler/crawler.go:163-170), self-loops
    kept, node ids randomly permuted; ~60 % of nodes end up with out-degree 0
    (the analogue of uncrawled frontier pages, pagerank.go:24-44).
  * inverted index: Zipf-Mandelbrot document frequencies df_r ~ 1/(r+q),
    each term's docs distinct, sorted ascending; tf = float32(c)/float32(m)
    (normalised tf = count/maxFreq, indexer/indexer.go:362).
"""
from __future__ import annotations

import math

import numpy as np

RMAT = (0.57, 0.19, 0.19, 0.05)


# ----------------------------------------------------------------------------- graph (numpy)
def _rmat_pairs(scale: int, m: int, rng: np.random.Generator, abcd=RMAT):
    a, b, c, _ = abcd
    src = np.zeros(m, dtype=np.int64)
    dst = np.zeros(m, dtype=np.int64)
    for bit in range(scale):
        r = rng.random(m)
        sbit = r >= a + b                      # quadrants c, d
        dbit = ((r >= a) & (r < a + b)) | (r >= a + b + c)   # quadrants b, d
        src |= sbit.astype(np.int64) << bit
        dst |= dbit.astype(np.int64) << bit
    return src, dst


def rmat_graph(n_nodes: int, n_edges: int, seed: int = 42, abcd=RMAT, permute: bool = True):
    """-> (out_ptr uint64[n+1], out_dst uint32[e]) with exactly n_edges unique edges."""
    rng = np.random.default_rng(seed)
    scale = max(1, math.ceil(math.log2(max(n_nodes, 2))))
    keys = np.zeros(0, dtype=np.int64)
    max_e = n_nodes * n_nodes
    if n_edges > max_e:
        raise ValueError("more edges than node pairs")
    guard = 0
    while len(keys) < n_edges:
        need = n_edges - len(keys)
        s, d = _rmat_pairs(scale, int(need * 1.5) + 64, rng, abcd)
        ok = (s < n_nodes) & (d < n_nodes)
        keys = np.unique(np.concatenate([keys, s[ok] * n_nodes + d[ok]]))
        guard += 1
        if guard > 200:   # tiny dense graphs: top up uniformly
            extra = rng.integers(0, max_e, size=need * 4)
            keys = np.unique(np.concatenate([keys, extra]))
    if len(keys) > n_edges:
        keys = np.sort(rng.permutation(keys)[:n_edges])
    src, dst = keys // n_nodes, keys % n_nodes
    if permute:
        perm = np.random.default_rng(seed + 1).permutation(n_nodes)
        src, dst = perm[src], perm[dst]
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    out_ptr = np.zeros(n_nodes + 1, dtype=np.uint64)
    np.cumsum(np.bincount(src, minlength=n_nodes), out=out_ptr[1:])
    return out_ptr, dst.astype(np.uint32)


def topic_sizes(n_nodes: int, k_topics: int) -> np.ndarray:
    """n_topic[k] = ceil(N/(k+2)) — any distinct values do (Q2)."""
    return np.array([math.ceil(n_nodes / (k + 2)) for k in range(k_topics)], dtype=np.int32)


# ----------------------------------------------------------------------------- index (numpy)
def zipf_df(n_terms: int, n_post: int, n_docs: int, q: float = 10.0, clip_frac: float = 0.25) -> np.ndarray:
    """Document frequencies df_r ~ c/(r+q), r=1..T, clipped to clip_frac*n_docs, summing to ~n_post."""
    r = np.arange(1, n_terms + 1, dtype=np.float64)
    clip = max(1, int(n_docs * clip_frac))
    lo, hi = 0.0, float(n_post) * (n_terms + q)
    for _ in range(80):
        c = 0.5 * (lo + hi)
        tot = np.minimum(np.rint(c / (r + q)), clip).sum()
        if tot < n_post:
            lo = c
        else:
            hi = c
    return np.minimum(np.rint(hi / (r + q)), clip).astype(np.int64)


def make_tf(n: int, rng: np.random.Generator) -> np.ndarray:
    m = rng.integers(1, 17, size=n)
    c = (rng.random(n) * m).astype(np.int64) + 1
    return (c.astype(np.float32) / m.astype(np.float32)).astype(np.float32)


def zipf_index(n_docs: int, n_terms: int, n_post: int, seed: int = 44, q: float = 10.0, clip_frac: float = 0.25):
    """-> (term_ptr uint64[T+1], post_doc uint32[P], post_tf float32[P]); term id = frequency rank-1."""
    rng = np.random.default_rng(seed)
    df = zipf_df(n_terms, n_post, n_docs, q, clip_frac)
    docs = []
    for t in range(n_terms):
        k = int(df[t])
        if k == 0:
            docs.append(np.zeros(0, dtype=np.uint32))
        elif k * 4 > n_docs:
            docs.append(np.sort(rng.permutation(n_docs)[:k]).astype(np.uint32))
        else:
            docs.append(np.sort(rng.choice(n_docs, size=k, replace=False)).astype(np.uint32))
    term_ptr = np.zeros(n_terms + 1, dtype=np.uint64)
    np.cumsum(df, out=term_ptr[1:])
    post_doc = np.concatenate(docs) if docs else np.zeros(0, np.uint32)
    return term_ptr, post_doc.astype(np.uint32), make_tf(len(post_doc), rng)


def make_queries(n_q: int, terms_per_q: int, max_rank: int, seed: int = 45):
    """n_q queries of `terms_per_q` DISTINCT term ids drawn uniformly from the max_rank most frequent terms.
    -> (q_ptr uint32[n_q+1], q_terms uint32[n_q*terms_per_q])"""
    rng = np.random.default_rng(seed)
    out = np.zeros((n_q, terms_per_q), dtype=np.uint32)
    for i in range(n_q):
        out[i] = rng.choice(max_rank, size=terms_per_q, replace=False)
    q_ptr = (np.arange(n_q + 1) * terms_per_q).astype(np.uint32)
    return q_ptr, out.reshape(-1)


# ----------------------------------------------------------------------------- torch (device) versions
def rmat_graph_torch(n_nodes: int, n_edges: int, seed: int = 42, device="cuda", abcd=RMAT):
    """Same construction on the GPU.  -> (out_ptr int64[n+1], out_dst int32[e]) device tensors
    (bit patterns of uint64/uint32)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    a, b, c, _ = abcd
    scale = max(1, math.ceil(math.log2(max(n_nodes, 2))))
    keys = torch.zeros(0, dtype=torch.int64, device=device)
    while keys.numel() < n_edges:
        need = n_edges - keys.numel()
        m = int(need * 1.5) + 1024
        src = torch.zeros(m, dtype=torch.int64, device=device)
        dst = torch.zeros(m, dtype=torch.int64, device=device)
        for bit in range(scale):
            r = torch.rand(m, generator=g, device=device, dtype=torch.float64)
            src |= (r >= a + b).to(torch.int64) << bit
            dst |= (((r >= a) & (r < a + b)) | (r >= a + b + c)).to(torch.int64) << bit
        ok = (src < n_nodes) & (dst < n_nodes)
        keys = torch.unique(torch.cat([keys, src[ok] * n_nodes + dst[ok]]))
        del src, dst, r, ok
    if keys.numel() > n_edges:
        sel = torch.randperm(keys.numel(), generator=g, device=device)[:n_edges]
        keys = keys[sel]
    src, dst = keys // n_nodes, keys % n_nodes
    del keys
    g2 = torch.Generator(device=device)
    g2.manual_seed(seed + 1)
    perm = torch.randperm(n_nodes, generator=g2, device=device)
    src, dst = perm[src], perm[dst]
    key2 = torch.sort(src * n_nodes + dst).values
    src, dst = key2 // n_nodes, key2 % n_nodes
    out_ptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=device)
    out_ptr[1:] = torch.cumsum(torch.bincount(src, minlength=n_nodes), dim=0)
    return out_ptr, dst.to(torch.int32)


def zipf_index_torch(n_docs: int, n_terms: int, n_post: int, seed: int = 44, device="cuda", q: float = 10.0,
                     clip_frac: float = 0.25, chunk_posts: int = 1 << 27):
    """Device-side index generator for the 10M-doc configuration.
    -> (term_ptr int64[T+1], post_doc int32[P], post_tf float32[P]) device tensors.
    For each term, m = -N ln(1-df/N) uniform draws are de-duplicated, so the expected number of
    distinct docs is df; P therefore matches n_post only approximately."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    df = zipf_df(n_terms, n_post, n_docs, q, clip_frac).astype(np.float64)
    draws = np.ceil(-n_docs * np.log1p(-np.minimum(df / n_docs, 0.999))).astype(np.int64)
    draws[df == 0] = 0
    cum = np.concatenate([[0], np.cumsum(draws)])
    keys_all = []
    t0 = 0
    while t0 < n_terms:
        # terms [t0, t1) whose draws fit one chunk
        t1 = int(np.searchsorted(cum, cum[t0] + chunk_posts, side="right")) - 1
        t1 = max(t1, t0 + 1)
        t1 = min(t1, n_terms)
        cnt = torch.from_numpy(draws[t0:t1]).to(device)
        m = int(cnt.sum().item())
        if m:
            term = torch.repeat_interleave(torch.arange(t0, t1, device=device, dtype=torch.int64), cnt)
            doc = torch.randint(0, n_docs, (m,), generator=g, device=device, dtype=torch.int64)
            keys_all.append(torch.unique(term * n_docs + doc))
            del term, doc
        t0 = t1
    keys = torch.cat(keys_all) if keys_all else torch.zeros(0, dtype=torch.int64, device=device)
    del keys_all
    term = keys // n_docs
    post_doc = (keys % n_docs).to(torch.int32)
    del keys
    term_ptr = torch.zeros(n_terms + 1, dtype=torch.int64, device=device)
    term_ptr[1:] = torch.cumsum(torch.bincount(term, minlength=n_terms), dim=0)
    del term
    P = post_doc.numel()
    mm = torch.randint(1, 17, (P,), generator=g, device=device, dtype=torch.int32)
    cc = (torch.rand(P, generator=g, device=device) * mm).to(torch.int32) + 1
    cc = torch.minimum(cc, mm)
    post_tf = cc.to(torch.float32) / mm.to(torch.float32)
    return term_ptr, post_doc, post_tf
