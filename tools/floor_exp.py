"""What would a good per-query threshold floor buy the mixed batch?  EXPERIMENT, needs the variant library (tools/build_variant.sh floor -DSS_EXP_FLOOR;
SS_LIB_PATH=spaghettisearch_amd/libspaghetti_rank_floor.so): option "score.debug_floor" = 1 makes a host-output call record
every query's k-th best FinalRank (rounded down); the next calls of the same batch start their filters from those — the best floor there can
be.  Mixed / half-half / tail / head batches, back-to-back ms per batch: default routing and every suited query forced into the wave kernel
("score.wave_min_list" = 0), each without and with the floors; hits compared with the default's.
Round 5 (ms per batch, floors off / on): mixed 0.158 / 0.140 (forced into the wave kernel 0.427 / 0.153), half head half tail 0.228 / 0.216, tail 0.089 / 0.085,
head (config 3) 0.328 / 0.301 — even the PERFECT floor buys 4-11 %: the batches' time is per-slice fixed cost, not survivors."""
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
    return statistics.median(ws) * 1e3, (d_hits.cpu().numpy().tobytes(), d_n.cpu().numpy().tobytes())
qh = synth.make_queries(nq // 2, 3, 10_000, seed=45); qt_ = synth.make_queries(nq // 2, 3, 1_000_000, seed=47)
half = (np.concatenate([qh[0], qh[0][-1] + qt_[0][1:]]).astype(np.uint32), np.concatenate([qh[1], qt_[1]]))
for name, (qp, qt) in (("mixed", synth.make_queries(nq, 3, 100_000, seed=46)), ("half head / half tail", half),
                       ("tail", synth.make_queries(nq, 3, 1_000_000, seed=47)), ("head", synth.make_queries(nq, 3, 10_000, seed=45))):
    ref = None
    for wml in (None, 0):
        for floor in (0, 1):
            ctx.set_option("score.wave_min_list", wml); ctx.set_option("score.debug_floor", floor)
            if floor: sc.score_topk(qp, qt, k)           # host-output call: records the floors
            ms, got = rate(qp, qt)
            if ref is None: ref = got
            print(f"{name}: wave_min_list {'default' if wml is None else wml}, floors {'on' if floor else 'off'}: {ms:.4f} ms per batch, hits == default: {got == ref}", flush=True)
    ctx.set_option("score.wave_min_list", None); ctx.set_option("score.debug_floor", None)
sc.close(); ti.close(); bi.close(); ctx.set_stream(None); ctx.close()
