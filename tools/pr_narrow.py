import os, statistics, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
for n, e, kt in ((1 << 20, 5_000_000, 1), (1 << 20, 5_000_000, 2), (10_000_000, 50_000_000, 1)):
    out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
    g = engine.Graph(ctx, n, out_ptr, out_dst)
    for narrow in (0, 1):
        ctx.set_option("pr.force_narrow", narrow)
        pr = engine.PageRankState(g, 0.75, -1.0, synth.topic_sizes(n, kt), max_iter=0)
        pr.begin(); pr.step(5)
        ms = []
        for _ in range(5):
            pr.step(20); ctx.synchronize(); ms.append(ctx.last_kernel_ms(0) / 20)
        print(f"N={n} E={e} K={kt} force_narrow={narrow}: sweep median {statistics.median(ms):.4f} ms", flush=True)
        pr.close()
    g.close()
ctx.close()
