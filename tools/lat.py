import time, numpy as np, torch, sys
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
k = 100
for label, maxr in (("head", 10_000), ("tail", nt)):
    for nq in (1, 8, 64, 256, 512, 1024, 2048, 4096):
        q_ptr, q_terms = synth.make_queries(nq, 3, maxr, seed=5)
        dq = (torch.from_numpy(q_ptr.view(np.int32)).to(dev), torch.from_numpy(q_terms.view(np.int32)).to(dev))
        d_hits = torch.empty(nq * k * 40, dtype=torch.uint8, device=dev); d_n = torch.empty(nq, dtype=torch.int32, device=dev)
        for _ in range(3): sc.score_topk(dq[0], dq[1], k, out=(d_hits, d_n))
        torch.cuda.synchronize(); t0 = time.perf_counter(); km = 0
        R = 20
        for _ in range(R):
            sc.score_topk(dq[0], dq[1], k, out=(d_hits, d_n)); km += ctx.last_kernel_ms(1)
        torch.cuda.synchronize(); w = (time.perf_counter() - t0) / R * 1e3
        # host in/out
        t0 = time.perf_counter()
        for _ in range(R): sc.score_topk(q_ptr, q_terms, k)
        wh = (time.perf_counter() - t0) / R * 1e3
        print(f"{label} nq={nq:5d} wall {w:.3f} ms  kernels {km/R:.3f} ms  host-in/out {wh:.3f} ms", flush=True)
