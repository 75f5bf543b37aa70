"""Small batches: host-in / host-out latency and kernel ms with the default routing, with k_score_wave forced off and forced on.
    python tools/lat3.py"""
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
for k in (50, 100):
    for nq in (1, 2, 4, 8, 16, 32, 64, 128):
        q_ptr, q_terms = synth.make_queries(nq, 3, 10_000, seed=5)
        row = []
        for mode, opts in (("default", {}), ("slices", {"score__wave": 0})):
            with ctx.options(**opts):
                for _ in range(3): sc.score_topk(q_ptr, q_terms, k)
                lat, km = [], []
                for _ in range(30):
                    t0 = time.perf_counter(); sc.score_topk(q_ptr, q_terms, k); lat.append((time.perf_counter() - t0) * 1e3); km.append(ctx.last_kernel_ms(1))
            row.append(f"{mode}: {statistics.median(lat):.3f} ms (kernels {statistics.median(km):.3f})")
        print(f"k={k} nq={nq:4d}  " + "   ".join(row), flush=True)
sc.close(); ti.close(); bi.close(); ctx.close()
