"""Sweep time of the PageRank kernel by number of topic vectors (10M nodes / 50M edges).  OPTS="pr.stagger=0" sets options."""
import os, time, sys, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
for kv in os.environ.get("OPTS", "").split(","):
    if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
n, e = 10_000_000, 50_000_000
ptr, dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
torch.cuda.synchronize()
g = engine.Graph(ctx, n, ptr, dst)
for kt in (1, 2, 4, 8, 16):
    pr = engine.PageRankState(g, 0.75, -1.0, synth.topic_sizes(n, kt), max_iter=0)
    pr.begin(); pr.step(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); pr.step(20); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"K={kt:2d}: {dt * 1e3:.3f} ms/sweep  {kt / dt:.0f} topic-iterations/s", flush=True)
    pr.close()
