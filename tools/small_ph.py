"""phase clocks of k_score_small (variant build -DSSS_PHASES): SS_LIB_PATH=spaghettisearch_amd/libspaghetti_rank_sssph.so python tools/small_ph.py"""
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
k, nq = 100, int(os.environ.get("NQ", "1024"))
d_hits = torch.empty(nq * k * 40, dtype=torch.uint8, device=dev); d_n = torch.empty(nq, dtype=torch.int32, device=dev)
qp, qt = synth.make_queries(nq, 3, 1_000_000, seed=47)
ctx.set_option("score.small", 1); ctx.set_option("score.pipeline", 0)
ms = []
for i in range(12):
    sc.score_topk(qp, qt, k, out=(d_hits, d_n)); ctx.synchronize(); ms.append(ctx.last_kernel_ms(1))
print(f"tail nq={nq} small=1: device ms per batch median {sorted(ms)[len(ms)//2]:.4f} min {min(ms):.4f}", flush=True)
sc.close(); ti.close(); bi.close(); ctx.close()
