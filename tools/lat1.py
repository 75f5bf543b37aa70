import time, numpy as np, torch, sys, statistics
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
q_ptr, q_terms = synth.make_queries(1, 3, 10_000, seed=5)
for _ in range(5): sc.score_topk(q_ptr, q_terms, 50)
ctx.set_option("score.trace", 1)
for _ in range(3):
    t0 = time.perf_counter(); sc.score_topk(q_ptr, q_terms, 50); print("call ms", (time.perf_counter() - t0) * 1e3, "kernels", ctx.last_kernel_ms(1))
