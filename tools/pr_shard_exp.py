"""Local sweep time of ONE rank of a doc-range-sharded config-4 graph (no exchange: the table keeps its begin() contents;
fixed-iteration mode) — the compute half of the N-GPU sweep, measured on one GPU.
    W=8 python tools/pr_shard_exp.py"""
import os, statistics, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
n, e, kt = 10_000_000, 50_000_000, 16
out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
for W in [int(x) for x in os.environ.get("W", "1,2,4,8").split(",")]:
    g = engine.Graph(ctx, n, out_ptr, out_dst, rank=0, world=W)
    pr = engine.PageRankState(g, 0.75, -1.0, synth.topic_sizes(n, kt), max_iter=0)
    pr.begin()
    if W > 1:
        pr.finalize()
    ms = []
    for _ in range(12):
        pr.step(1)
        ctx.synchronize()
        ms.append(ctx.last_kernel_ms(0))
        if W > 1:
            pr.finalize()
    gi = g.info()
    print(f"world {W}: rank 0 holds {gi.n_rows_local} rows / {gi.n_edges_local} edges; local sweep median {statistics.median(ms[2:]):.4f} ms (ideal 1/W of world 1)", flush=True)
    pr.close(); g.close()
ctx.close()
