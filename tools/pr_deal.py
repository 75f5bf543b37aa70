"""Sweep time under the two item deals: pr.deal_global = 1 (all items by falling cost, default) against 0 (chunks in table order).
    python tools/pr_deal.py"""
import statistics, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
for n, e, kt in ((1 << 20, 5_000_000, 1), (1 << 20, 5_000_000, 16), (10_000_000, 50_000_000, 1), (10_000_000, 50_000_000, 16)):
    out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
    g = engine.Graph(ctx, n, out_ptr, out_dst)
    ref = None
    for name, opts in (("global", {}), ("chunks", {"pr.deal_global": 0}), ("class-major", {"pr.deal_global": 2}), ("global", {}), ("chunks", {"pr.deal_global": 0}), ("class-major", {"pr.deal_global": 2})):
        for k, v in opts.items(): ctx.set_option(k, v)
        pr = engine.PageRankState(g, 0.75, -1.0, synth.topic_sizes(n, kt), max_iter=0)
        pr.begin(); pr.step(5)
        ms = []
        for _ in range(7):
            pr.step(20); ctx.synchronize(); ms.append(ctx.last_kernel_ms(0) / 20)
        x = pr.read()
        if ref is None: ref = x
        err = float(np.max(np.abs(x - ref) / ref))
        print(f"N={n} E={e} K={kt} {name}: sweep median {statistics.median(ms):.4f} ms  min {min(ms):.4f}  max rel diff to the first {err:.1e}", flush=True)
        pr.close()
        for k in opts: ctx.set_option(k, None)
    g.close()
    del out_ptr, out_dst
ctx.close()
