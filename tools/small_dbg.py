"""tail / mixed batches: device ms per call (ss_last_kernel_ms(1), one batch at a time: score.pipeline = 0) with and without k_score_small"""
import os, sys, time
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
k, nq = 100, 1024
d_hits = torch.empty(nq * k * 40, dtype=torch.uint8, device=dev); d_n = torch.empty(nq, dtype=torch.int32, device=dev)
for name, rmax, seed in (("tail", 1_000_000, 47),):
    qp, qt = synth.make_queries(nq, 3, rmax, seed=seed)
    for small in (1, 0):
        ctx.set_option("score.small", small)
        ctx.set_option("score.pipeline", 0)
        ms = []
        for i in range(int(os.environ.get("REPS", "12"))):
            sc.score_topk(qp, qt, k, out=(d_hits, d_n)); ctx.synchronize()
            ms.append(ctx.last_kernel_ms(1))
        print(f"{name} small={small}: device ms per batch (one at a time) median {sorted(ms)[len(ms)//2]:.4f} min {min(ms):.4f}", flush=True)
sc.close(); ti.close(); bi.close(); ctx.close()
