"""ss_score_topk_submit / _collect at config 3 with DEPTH batches in flight (default 3): ms per batch host to host; for a kernel trace
(tools/kt.sh + tools/trace_timeline.py).   DEPTH=3 python tools/score_inflight.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
from spaghettisearch_amd.engine import HIT_DTYPE
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
q_ptr, q_terms = synth.make_queries(nq, 3, 10000, seed=45)
depth = int(os.environ.get("DEPTH", "3"))
if os.environ.get("NOTIMING"): ctx.set_option("score.timing", 0)
for kv in os.environ.get("OPTS", "").split(","):
    if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
outs = [(np.zeros((nq, k), dtype=HIT_DTYPE), np.zeros(nq, dtype=np.int32)) for _ in range(depth)]
def run(n):
    flight = []
    t0 = time.perf_counter()
    for i in range(n):
        if len(flight) == depth:
            tk, o = flight.pop(0); sc.collect(tk, out=o)
        flight.append((sc.submit(q_ptr, q_terms, k), outs[i % depth]))
    for tk, o in flight: sc.collect(tk, out=o)
    return (time.perf_counter() - t0) / n * 1e3
run(10)
print("ms per batch, host to host, %d in flight:" % depth, ["%.3f" % run(40) for _ in range(5)], flush=True)
sc.close(); ti.close(); bi.close(); ctx.close()
