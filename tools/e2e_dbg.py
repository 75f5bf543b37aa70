import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
stream = torch.cuda.Stream(device=dev); ctx.set_stream(stream.cuda_stream)
with torch.cuda.stream(stream):
    n, e, k = 10_000_000, 50_000_000, 16
    out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
    n_topic = synth.topic_sizes(n, k)
    g = engine.Graph(ctx, n, out_ptr, out_dst)
    import os
    pr = engine.PageRankState(g, 0.75, -1.0, n_topic, max_iter=0); pr.begin()
    if not os.environ.get('NOSTEP'): pr.step(30)
    if os.environ.get("CLOSE_FIRST"): pr.close(); g.close()
    rank_dev = torch.empty((k, n), dtype=torch.float64, device=dev)
    if os.environ.get('TRACE'): ctx.set_option('pr.trace', 1)
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ge = engine.Graph(ctx, n, out_ptr, out_dst)
        ctx.synchronize(); t1 = time.perf_counter()
        it = ge.pagerank_dev(0.75, 1e-6, n_topic, rank_dev, max_iter=500)
        ctx.synchronize(); t2 = time.perf_counter()
        ge.close()
        print(f"rep {rep}: create {(t1-t0)*1e3:.2f} ms run {(t2-t1)*1e3:.2f} ms", flush=True)
