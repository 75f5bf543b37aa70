"""Host in / host out rate of ss_score_topk at config 3 (what a cgo caller pays): 20 calls, fresh numpy outputs per call as engine.score_topk
makes them, and with caller-owned reused / pinned outputs.   python tools/score_hostio.py"""
import sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
from spaghettisearch_amd.engine import _ptr, check, HIT_DTYPE
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
for _ in range(5): sc.score_topk(q_ptr, q_terms, k)
def rate(fn, n=20):
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
print("fresh numpy outputs per call: %.3f ms" % rate(lambda: sc.score_topk(q_ptr, q_terms, k)))
hits = np.zeros((nq, k), dtype=HIT_DTYPE); nh = np.zeros(nq, dtype=np.int32)
def reuse():
    check(ctx.lib.ss_score_topk(sc.h, nq, _ptr(q_ptr), _ptr(q_terms), None, None, k, hits.ctypes.data, _ptr(nh)), ctx.h)
print("caller reuses its (pageable) outputs: %.3f ms" % rate(reuse))
ph = torch.empty(nq * k * 40, dtype=torch.uint8).pin_memory(); pn = torch.empty(nq, dtype=torch.int32).pin_memory()
def pinned():
    check(ctx.lib.ss_score_topk(sc.h, nq, _ptr(q_ptr), _ptr(q_terms), None, None, k, ph.data_ptr(), pn.data_ptr()), ctx.h)
print("caller-pinned outputs: %.3f ms" % rate(pinned))
# batches in flight: ss_score_topk_submit / _collect, the caller reuses its outputs
for depth in (1, 2, 3):
    outs = [(np.zeros((nq, k), dtype=HIT_DTYPE), np.zeros(nq, dtype=np.int32)) for _ in range(depth)]
    tsub = tcol = 0.0
    def run(n=40):
        global tsub, tcol
        flight = []
        tsub = tcol = 0.0
        t0 = time.perf_counter()
        for i in range(n):
            if len(flight) == depth:
                tk, o = flight.pop(0)
                ta = time.perf_counter(); sc.collect(tk, out=o); tcol += time.perf_counter() - ta
            ta = time.perf_counter(); tk = sc.submit(q_ptr, q_terms, k); tsub += time.perf_counter() - ta
            flight.append((tk, outs[i % depth]))
        for tk, o in flight: sc.collect(tk, out=o)
        return (time.perf_counter() - t0) / n * 1e3
    run(10)
    ms = min(run() for _ in range(3))
    print("   per batch in submit %.3f ms, in collect %.3f ms" % (tsub / 40 * 1e3, tcol / 40 * 1e3))
    same = bool(np.array_equal(outs[0][0], hits) and np.array_equal(outs[0][1], nh))
    print("submit/collect, %d in flight: %.3f ms per batch = %.2fM queries/s  (hits == the synchronous call's: %s)" % (depth, ms, nq / ms / 1e3, same))
ctx.set_option("score.trace", 1); reuse(); ctx.set_option("score.trace", None)
sc.close(); ti.close(); bi.close(); ctx.close()
