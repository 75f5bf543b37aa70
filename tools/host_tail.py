"""host cost of a scoring call by batch kind (tail / mixed / head): host-only call time, back-to-back wall, and the library's phase trace"""
import os, statistics, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
stream = torch.cuda.Stream(device=dev); ctx.set_stream(stream.cuda_stream)
torch.cuda.set_stream(stream)
nd, nt = 10_000_000, 1_000_000
b = synth.zipf_index_torch(nd, nt, 640_000_000, seed=44, device=dev)
t = synth.zipf_index_torch(nd, nt, 40_000_000, seed=144, device=dev)
bi = engine.InvertedIndex(ctx, nd, *b); ti = engine.InvertedIndex(ctx, nd, *t)
del b, t
ti.tfidf_build(nd, False, False, False); bi.tfidf_build(nd, False, False, False)
sc = engine.Scorer(ctx, ti, bi)
k, nq = 100, 1024
d_hits = torch.empty(nq * k * 40, dtype=torch.uint8, device=dev); d_n = torch.empty(nq, dtype=torch.int32, device=dev)
ctx.set_option("score.timing", 0)
kinds = os.environ.get("KINDS", "tail,mixed,head").split(",")
for name, rmax, seed in (("tail", 1_000_000, 47), ("mixed", 100_000, 46), ("head", 10_000, 45)):
    if name not in kinds: continue
    qp, qt = synth.make_queries(nq, 3, rmax, seed=seed)
    for _ in range(10): sc.score_topk(qp, qt, k, out=(d_hits, d_n))
    ctx.synchronize()
    host = []
    for _ in range(30):
        t0 = time.perf_counter(); sc.score_topk(qp, qt, k, out=(d_hits, d_n)); host.append(time.perf_counter() - t0); ctx.synchronize()
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(100): sc.score_topk(qp, qt, k, out=(d_hits, d_n))
    ctx.synchronize(); wall = (time.perf_counter() - t0) / 100
    print(f"{name}: host call median {statistics.median(host)*1e3:.3f} ms (min {min(host)*1e3:.3f}); back-to-back {wall*1e3:.3f} ms/batch", flush=True)
    ctx.set_option('score.trace', 1)
    for _ in range(2): sc.score_topk(qp, qt, k, out=(d_hits, d_n)); ctx.synchronize()
    ctx.set_option('score.trace', None)
sc.close(); ti.close(); bi.close(); ctx.set_stream(None); ctx.close()
