"""Host cost of one ss_score_topk call at config 3 (planning + enqueue, results left in HBM) beside the kernel time:
the batch rate is bounded by the larger of the two.   python tools/score_host.py"""
import os, statistics, sys, time
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
q_ptr, q_terms = synth.make_queries(nq, 3, 10000, seed=45)
dq = (torch.from_numpy(q_ptr.view(np.int32)).to(dev), torch.from_numpy(q_terms.view(np.int32)).to(dev))
d_hits = torch.empty(nq * k * 40, dtype=torch.uint8, device=dev); d_n = torch.empty(nq, dtype=torch.int32, device=dev)
for name, qp, qt in (("device query arrays", dq[0], dq[1]), ("host query arrays", q_ptr, q_terms)):
    for _ in range(5):
        sc.score_topk(qp, qt, k, out=(d_hits, d_n))
    ctx.synchronize()
    # (a) host cost alone: wait for the device after every call, clock only the call
    host = []
    for _ in range(30):
        t0 = time.perf_counter()
        sc.score_topk(qp, qt, k, out=(d_hits, d_n))
        host.append(time.perf_counter() - t0)
        ctx.synchronize()
    # (b) back to back
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        sc.score_topk(qp, qt, k, out=(d_hits, d_n))
    ctx.synchronize(); wall = (time.perf_counter() - t0) / 50
    print(f"{name}: host call median {statistics.median(host) * 1e3:.3f} ms (min {min(host) * 1e3:.3f}), back-to-back {wall * 1e3:.3f} ms/batch, "
          f"kernels {ctx.last_kernel_ms(1):.3f} ms", flush=True)
ctx.set_option('score.trace', 1)
for _ in range(3):
    sc.score_topk(q_ptr, q_terms, k, out=(d_hits, d_n)); ctx.synchronize()
sc.close(); ti.close(); bi.close(); ctx.close()
