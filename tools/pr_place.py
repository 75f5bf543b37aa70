"""Does the gather probe (and with it the sweep) depend on WHERE the table lands?  Fresh allocations (pool off) of the same state, several
times in one process: probe ms and sweep ms per allocation."""
import os, statistics, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
n, e, kt = 10_000_000, 50_000_000, 16
out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
g = engine.Graph(ctx, n, out_ptr, out_dst)
nt = synth.topic_sizes(n, kt)
if os.environ.get("NOPOOL", "1") == "1": ctx.set_option("mem.pool_mb", 0)
keep = []
for i in range(int(os.environ.get("TRIES", "8"))):
    pr = engine.PageRankState(g, 0.75, -1.0, nt, max_iter=0)
    pr.begin(); pr.step(5)
    ms = []
    for _ in range(3):
        pr.step(20); ctx.synchronize(); ms.append(ctx.last_kernel_ms(0) / 20)
    probe = pr.probe(0, 5)
    print(f"allocation {i}: sweep {statistics.median(ms):.4f} ms  gather probe {probe:.4f} ms", flush=True)
    if os.environ.get("HOLD") == "1": keep.append(pr)       # keep the blocks: the next state gets other memory
    else: pr.close()
for pr in keep: pr.close()
g.close(); ctx.close()
