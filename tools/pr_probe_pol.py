"""Gather-only probe of the config-4 graph under different cache policies of the gathers (ss_pr_probe mode + 8*policy)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
n, e = 10_000_000, 50_000_000
out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
g = engine.Graph(ctx, n, out_ptr, out_dst)
pr = engine.PageRankState(g, 0.75, -1.0, synth.topic_sizes(n, 16), max_iter=0)
pr.begin(); pr.step(3)
names = ["default", "all nt", "all sc1", "hot default / cold nt", "hot default / cold sc1"]
for pol in range(5):
    print(f"policy {pol} ({names[pol]}): real index stream {pr.probe(8 * pol, 5):.4f} ms   uniform random {pr.probe(8 * pol + 1, 5):.4f} ms", flush=True)
pr.close(); g.close(); ctx.close()
