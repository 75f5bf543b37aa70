"""Phrase search at config-3 size (timing only): 10M docs / 682M postings with 2 synthetic positions per posting;
a batch of NQ queries, each ONE 2-term quoted phrase (term ranks U[1, MAXRANK]) and no OR terms.
    NQ=256 MAXRANK=10000 python tools/phrase_exp.py"""
import os, statistics, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
nd, nt = 10_000_000, 1_000_000
idx = []
for P, seed in ((640_000_000, 44), (40_000_000, 144)):
    ptr, doc, tf = synth.zipf_index_torch(nd, nt, P, seed=seed, device=dev)
    ii = engine.InvertedIndex(ctx, nd, ptr, doc, tf)
    n = doc.numel()
    pos_ptr = (torch.arange(n + 1, dtype=torch.int64, device=dev) * 2)
    g = torch.Generator(device=dev); g.manual_seed(seed)
    first = torch.randint(0, 400, (n,), device=dev, generator=g, dtype=torch.int32)
    pos = torch.stack([first, first + torch.randint(1, 50, (n,), device=dev, generator=g, dtype=torch.int32)], dim=1).to(torch.float32).reshape(-1)
    del first
    ii.tfidf_build(nd, False, False, False)
    ii.set_positions(pos_ptr, pos)
    idx.append(ii)
    del ptr, doc, tf
bi, ti = idx
sc = engine.Scorer(ctx, ti, bi)
nq = int(os.environ.get("NQ", "256")); maxrank = int(os.environ.get("MAXRANK", "10000"))
p_ptr, p_terms = synth.make_queries(nq, 2, maxrank, seed=47)
q_ptr = np.zeros(nq + 1, dtype=np.uint32); q_terms = np.zeros(0, dtype=np.uint32)
ms = []
for i in range(6):
    ctx.synchronize(); t0 = time.perf_counter()
    hits, n_hits = sc.score_topk_phrase(q_ptr, q_terms, p_ptr, p_terms, 50)
    ms.append((time.perf_counter() - t0) * 1e3)
print(f"{nq} phrase queries (2 terms, ranks U[1,{maxrank}]): wall ms per batch {['%.2f' % m for m in ms]}; kernels {ctx.last_kernel_ms(1):.3f} ms; hits per query mean {n_hits.mean():.1f}", flush=True)
sc.close(); ti.close(); bi.close(); ctx.close()
