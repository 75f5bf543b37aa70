"""one query at a time, host in / host out (the reference's call shape, k = 50): wall ms by term rank range, k_score_small on / off"""
import os, sys, time, statistics
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
nd, nt = 10_000_000, 1_000_000
b = synth.zipf_index_torch(nd, nt, 640_000_000, seed=44, device=dev)
t = synth.zipf_index_torch(nd, nt, 40_000_000, seed=144, device=dev)
bi = engine.InvertedIndex(ctx, nd, *b); ti = engine.InvertedIndex(ctx, nd, *t)
del b, t
ti.tfidf_build(nd, False, False, False); bi.tfidf_build(nd, False, False, False)
sc = engine.Scorer(ctx, ti, bi)
for name, rmax, seed in (("tail U[1,1M]", 1_000_000, 47), ("mixed U[1,100k]", 100_000, 46), ("head U[1,10k]", 10_000, 45)):
    qp, qt = synth.make_queries(256, 3, rmax, seed=seed)
    for small in (1, 0):
        ctx.set_option("score.small", small)
        for nq in (1, 8, 64):
            lat = []
            for i in range(60):
                a = (i * nq) % (256 - nq + 1)
                p1 = (qp[a:a + nq + 1] - qp[a]).astype(np.uint32); t1 = qt[qp[a]:qp[a + nq]]
                t0 = time.perf_counter(); sc.score_topk(p1, t1, 50); lat.append((time.perf_counter() - t0) * 1e3)
            lat = lat[10:]
            print(f"{name:18s} small={small} batch of {nq:3d}: wall ms median {statistics.median(lat):.4f} min {min(lat):.4f} p90 {sorted(lat)[int(len(lat)*0.9)]:.4f}", flush=True)
sc.close(); ti.close(); bi.close(); ctx.close()
