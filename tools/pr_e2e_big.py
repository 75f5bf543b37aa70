"""Config 4 only, a few one-call updates (graph create + PageRank to eps 1e-6), for a kernel-trace timeline (tools/kt.sh + tools/trace_all.py)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
n, e, k = 10_000_000, 50_000_000, 16
out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
n_topic = synth.topic_sizes(n, k)
rank_dev = torch.empty((k, n), dtype=torch.float64, device=dev)
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = engine.Graph(ctx, n, out_ptr, out_dst)
    g.pagerank_dev(0.75, 1e-6, n_topic, rank_dev, max_iter=500)
    ctx.synchronize(); t1 = time.perf_counter()
    g.close()
    print(f"one call {1e3 * (t1 - t0):.2f} ms", flush=True)
ctx.close()
