"""mixed / tail / half-half batches against "score.slice_target" (postings per k_score_slices workgroup; default: from the batch) and "score.grade_slices":
back-to-back ms per batch, device outputs."""
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
def rate(qp, qt):
    for _ in range(30): sc.score_topk(qp, qt, k, out=(d_hits, d_n))
    ctx.synchronize(); ws = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(100): sc.score_topk(qp, qt, k, out=(d_hits, d_n))
        ctx.synchronize(); ws.append((time.perf_counter() - t0) / 100)
    return statistics.median(ws) * 1e3
for name, rmax, seed in (("mixed", 100_000, 46), ("tail", 1_000_000, 47)):
    qp, qt = synth.make_queries(nq, 3, rmax, seed=seed)
    for tgt in (None, 4096, 8192, 12288, 16384, 24576, 32768, 65536):
        for grade in (0, 1):
            ctx.set_option("score.slice_target", tgt); ctx.set_option("score.grade_slices", grade)
            print(f"{name}: slice_target {tgt} grade_slices {grade}: {rate(qp, qt):.4f} ms per batch", flush=True)
ctx.set_option("score.slice_target", None); ctx.set_option("score.grade_slices", None)
sc.close(); ti.close(); bi.close(); ctx.set_stream(None); ctx.close()
