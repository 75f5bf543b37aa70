"""Config-3 batch timing for kernel experiments: prints kernel ms (HIP events on the library's stream), median of R batches.
    [SS_LIB_PATH=...] [SS_SLICE_TARGET=n] NQ=1024 python tools/score_exp.py"""
import os, statistics, sys
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
k = int(os.environ.get("K", "100"))
nq = int(os.environ.get("NQ", "1024"))
q_ptr, q_terms = synth.make_queries(nq, 3, 10_000, seed=45)
dq = (torch.from_numpy(q_ptr.view(np.int32)).to(dev), torch.from_numpy(q_terms.view(np.int32)).to(dev))
d_hits = torch.empty(nq * k * 40, dtype=torch.uint8, device=dev); d_n = torch.empty(nq, dtype=torch.int32, device=dev)
ms = []
for i in range(int(os.environ.get("R", "25"))):
    sc.score_topk(dq[0], dq[1], k, out=(d_hits, d_n))
    if i >= 5: ms.append(ctx.last_kernel_ms(1))
print(f"slice_target={os.environ.get('SS_SLICE_TARGET', 'default')} lib={os.path.basename(os.environ.get('SS_LIB_PATH', 'product'))} nq={nq} k={k}: "
      f"kernels median {statistics.median(ms):.4f} ms  min {min(ms):.4f} ms", flush=True)
sc.close(); ti.close(); bi.close(); ctx.close()
