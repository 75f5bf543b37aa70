"""k_score_small's auto routing (score.small = 2: a call of at most score.small_max_batch queries that are all small): host-to-host ms
of calls of 1 .. 64 small tail queries (term ids 200k .. 1M: ~250 postings per list) with the routing off / auto at batch limit 64,
hits compared."""
import os, statistics, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
nd, nt = 10_000_000, 1_000_000
b = synth.zipf_index_torch(nd, nt, 640_000_000, seed=44, device=dev)
t = synth.zipf_index_torch(nd, nt, 40_000_000, seed=144, device=dev)
h_b = b[0].cpu().numpy().view(np.uint64); h_t = t[0].cpu().numpy().view(np.uint64)
bi = engine.InvertedIndex(ctx, nd, *b); ti = engine.InvertedIndex(ctx, nd, *t)
del b, t
ti.tfidf_build(nd, False, False, False); bi.tfidf_build(nd, False, False, False)
sc = engine.Scorer(ctx, ti, bi)
rng = np.random.default_rng(5)
for lo_rank, label in ((200_000, "terms 200k..1M"), (20_000, "terms 20k..1M")):
    for nq in (1, 2, 4, 8, 16, 32, 64):
        terms = np.stack([rng.choice(np.arange(lo_rank, nt), size=3, replace=False) for _ in range(nq)]).astype(np.uint32)
        tot = (h_b[terms + 1] - h_b[terms] + h_t[terms + 1] - h_t[terms]).sum(axis=1)
        qp = (np.arange(nq + 1) * 3).astype(np.uint32); qt = terms.reshape(-1)
        res = {}
        for mode in (0, 2):
            ctx.set_option("score.small", mode); ctx.set_option("score.small_max_batch", 64)
            for _ in range(10): sc.score_topk(qp, qt, 50)
            ls = []
            for _ in range(200):
                t0 = time.perf_counter(); h = sc.score_topk(qp, qt, 50); ls.append(time.perf_counter() - t0)
            res[mode] = (statistics.median(ls) * 1e3, h[0].tobytes(), h[1].tobytes())
        print(f"{label}: {nq:3d} queries (postings per query: median {int(np.median(tot))}, max {int(tot.max())}): off {res[0][0]:.4f} ms, auto {res[2][0]:.4f} ms, same hits {res[0][1:] == res[2][1:]}", flush=True)
sc.close(); ti.close(); bi.close(); ctx.close()
