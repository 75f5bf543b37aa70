"""k_score_slices' in-kernel counters for one tail batch (diag build: tools/build_diag.sh; SS_LIB_PATH=spaghettisearch_amd/libspaghetti_rank_diag.so)"""
import os, sys
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
qp, qt = synth.make_queries(nq, 3, int(os.environ.get("RMAX", "1000000")), seed=47)
ctx.set_option("score.small", 0); ctx.set_option("score.pipeline", 0)
for i in range(int(os.environ.get("REPS", "4"))):
    sc.score_topk(qp, qt, k, out=(d_hits, d_n)); ctx.synchronize()
print("device ms", ctx.last_kernel_ms(1), flush=True)
sc.close(); ti.close(); bi.close(); ctx.close()
