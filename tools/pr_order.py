"""Does the numbering of the input graph change the sweep time?  (locality experiment for K1)
orders: as generated (random permutation), by out-degree descending, by in-degree descending, by (in+out) desc."""
import time, sys, numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
n, e, kt = 10_000_000, 50_000_000, 16
ptr, dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
deg = (ptr[1:] - ptr[:-1])
src = torch.repeat_interleave(torch.arange(n, device=dev), deg)
dst64 = dst.to(torch.int64)
indeg = torch.bincount(dst64, minlength=n)
n_topic = synth.topic_sizes(n, kt)

def relabel(key):
    if key is None:
        return ptr, dst
    order = torch.argsort(key, descending=True, stable=True)          # old ids in new order
    new_id = torch.empty(n, dtype=torch.int64, device=dev); new_id[order] = torch.arange(n, device=dev)
    s2, d2 = new_id[src], new_id[dst64]
    o = torch.argsort(s2 * n + d2)
    s2, d2 = s2[o], d2[o]
    p2 = torch.zeros(n + 1, dtype=torch.int64, device=dev); p2[1:] = torch.cumsum(torch.bincount(s2, minlength=n), 0)
    return p2, d2.to(torch.int32)

for label, key in (("generated", None), ("out-degree desc", deg), ("in-degree desc", indeg), ("in+out desc", deg + indeg)):
    p2, d2 = relabel(key)
    torch.cuda.synchronize()
    g = engine.Graph(ctx, n, p2, d2)
    pr = engine.PageRankState(g, 0.75, -1.0, n_topic, max_iter=0)
    pr.begin(); pr.step(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); pr.step(20); torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/sweep (kernel {ctx.last_kernel_ms(0) / 20:.3f})", flush=True)
    pr.close(); g.close()
