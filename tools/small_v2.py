"""k_score_small on / off at the config-3 index: back-to-back wall ms per batch (device outputs, default pipelining) for tail, mixed and
half head / half tail batches, 8 tail queries and one tail query host to host; hits compared between the two routings."""
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
qh = synth.make_queries(nq // 2, 3, 10_000, seed=45); qt_ = synth.make_queries(nq // 2, 3, 1_000_000, seed=47)
half = (np.concatenate([qh[0], qh[0][-1] + qt_[0][1:]]).astype(np.uint32), np.concatenate([qh[1], qt_[1]]))
batches = [("tail", synth.make_queries(nq, 3, 1_000_000, seed=47)), ("mixed", synth.make_queries(nq, 3, 100_000, seed=46)), ("half head / half tail", half)]
cfgs = [("routing off", {"score.small": 0}), ("forced on the caller's stream", {"score.small": 1}),
        ("staged, one table size", {"score.small": 2, "score.small_batch": 1}), ("staged, two table sizes", {"score.small": 2, "score.small_batch": 2})]
for name, (qp, qt) in batches:
    ref = None
    for label, opts in cfgs:
        for o in ("score.small", "score.small_batch"): ctx.set_option(o, opts.get(o))
        for _ in range(20): sc.score_topk(qp, qt, k, out=(d_hits, d_n))
        ctx.synchronize()
        ws = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(100): sc.score_topk(qp, qt, k, out=(d_hits, d_n))
            ctx.synchronize(); ws.append((time.perf_counter() - t0) / 100)
        got = (d_hits.cpu().numpy().tobytes(), d_n.cpu().numpy().tobytes())
        if ref is None: ref = got
        print(f"{name}: {label}: {statistics.median(ws)*1e3:.4f} ms per batch (min {min(ws)*1e3:.4f})  hits == routing off: {got == ref}", flush=True)
for o in ("score.small", "score.small_batch"): ctx.set_option(o, None)
for nqs in (1, 8):
    qp, qt = synth.make_queries(nqs, 3, 1_000_000, seed=48)
    for small in (0, 1):
        ctx.set_option("score.small", 2 if small else 0)
        for _ in range(10): sc.score_topk(qp, qt, 50)
        ls = []
        for _ in range(200):
            t0 = time.perf_counter(); sc.score_topk(qp, qt, 50); ls.append(time.perf_counter() - t0)
        print(f"{nqs} tail quer{'y' if nqs == 1 else 'ies'} host to host, score.small={small}: median {statistics.median(ls)*1e3:.4f} ms", flush=True)
sc.close(); ti.close(); bi.close(); ctx.set_stream(None); ctx.close()
