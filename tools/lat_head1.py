"""one head query at a time (term ranks U[1,10k], k = 50, host in / host out): wall ms by score.slice_target"""
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
qp, qt = synth.make_queries(256, 3, 10_000, seed=45)
for tgt in [None] + [int(x) for x in os.environ.get("TARGETS", "4096,8192,16384,32768,65536").split(",")]:
    ctx.set_option("score.slice_target", tgt)
    lat = []
    for i in range(80):
        p1 = np.array([0, 3], dtype=np.uint32); t1 = qt[3 * i:3 * i + 3]
        t0 = time.perf_counter(); sc.score_topk(p1, t1, 50); lat.append((time.perf_counter() - t0) * 1e3)
    lat = lat[10:]
    print(f"slice_target {tgt}: wall ms median {statistics.median(lat):.4f} min {min(lat):.4f} p90 {sorted(lat)[int(len(lat)*0.9)]:.4f}", flush=True)
sc.close(); ti.close(); bi.close(); ctx.close()
