"""Sweep time against the work-item granularity ("pr.item_turns") at config 2 and config 4.   python tools/pr_items.py"""
import statistics, sys
import torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
for n, e, kt in ((1 << 20, 5_000_000, 1), (10_000_000, 50_000_000, 16)):
    out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
    g = engine.Graph(ctx, n, out_ptr, out_dst)
    for it in (2, 4, 8, 16, 32):
        ctx.set_option("pr.item_turns", it)
        pr = engine.PageRankState(g, 0.75, -1.0, synth.topic_sizes(n, kt), max_iter=0)
        pr.begin(); pr.step(5)
        ms = []
        for _ in range(5):
            pr.step(20); ctx.synchronize(); ms.append(ctx.last_kernel_ms(0) / 20)
        print(f"N={n} K={kt} item_turns={it}: sweep median {statistics.median(ms):.4f} ms", flush=True)
        pr.close()
    g.close()
ctx.close()
