import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
n, e = 10_000_000, 50_000_000
out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
g = engine.Graph(ctx, n, out_ptr, out_dst)
ctx.synchronize(); t1 = time.perf_counter()
pr = engine.PageRankState(g, 0.75, 1e-6, synth.topic_sizes(n, 16), max_iter=0)
ctx.synchronize(); t2 = time.perf_counter()
pr.begin(); ctx.synchronize(); t3 = time.perf_counter()
r = pr.read(); t4 = time.perf_counter()
print(f"graph_create {1e3*(t1-t0):.1f} ms, pr_create {1e3*(t2-t1):.1f} ms, begin {1e3*(t3-t2):.2f} ms, read(16x10M f64 to host) {1e3*(t4-t3):.1f} ms")
pr.close(); g.close(); ctx.close()
