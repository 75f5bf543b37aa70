"""K <= 2 sweep time by kernel: k_pr_sweep_n (default, round 4), the choice before it (pr.narrow_wave = 0: padded 8-wide sweep on small
graphs, k_pr_step on large ones) and k_pr_step (pr.force_narrow = 1); IT=<pr.item_turns> sweeps the item size of the default.
    python tools/pr_narrow.py"""
import os, statistics, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
its = [int(x) for x in os.environ.get("IT", "0").split(",")]
for n, e, kt in ((1 << 20, 5_000_000, 1), (1 << 20, 5_000_000, 2), (10_000_000, 50_000_000, 1), (10_000_000, 50_000_000, 2)):
    out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
    g = engine.Graph(ctx, n, out_ptr, out_dst)
    ref = None
    for name, opts in [("sweep_n", {})] + [(f"sweep_n it={i}", {"pr.item_turns": i}) for i in its if i] + [("before", {"pr.narrow_wave": 0}), ("k_pr_step", {"pr.force_narrow": 1})]:
        for k, v in opts.items(): ctx.set_option(k, v)
        pr = engine.PageRankState(g, 0.75, -1.0, synth.topic_sizes(n, kt), max_iter=0)
        pr.begin(); pr.step(5)
        ms = []
        for _ in range(5):
            pr.step(20); ctx.synchronize(); ms.append(ctx.last_kernel_ms(0) / 20)
        x = pr.read()
        if ref is None: ref = x
        err = float(np.max(np.abs(x - ref) / ref))
        print(f"N={n} E={e} K={kt} {name}: sweep median {statistics.median(ms):.4f} ms  max rel diff to the first {err:.1e}", flush=True)
        pr.close()
        for k in opts: ctx.set_option(k, None)
    g.close()
    del out_ptr, out_dst
ctx.close()
