"""Does the sweep depend on WHERE its state lands?  The same state allocated several times in one process, sweep ms and gather-probe ms
per allocation.  NOPOOL=1 (default here): the library's pool off, every block a fresh hipMalloc; HOLD=1: earlier states stay alive, so
each new one gets other memory; GAP_MB / PRE_MB: a torch block between the graph's arrays and the state / in front of the graph;
TRACE=1 prints the buffers' addresses.  Round 5: pool on 0.905 x 4; pool off 0.94 every time (same addresses again and again);
pool off + HOLD 0.94 for the first state, 0.906-0.92 for every later one; gaps, a 2 MB minimum block and moving any of the small
buffers change nothing; the gather probe is 0.722-0.727 throughout.  Read: a state whose large arrays are fresh hipMallocs into
the holes the graph build's temporaries left is the slow one; whole reused pool blocks (the default) are not."""
import os, statistics, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
n, e, kt = 10_000_000, 50_000_000, 16
out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
if os.environ.get("PRE_MB"): pre = torch.empty(int(os.environ["PRE_MB"]) << 20, dtype=torch.uint8, device=dev); torch.cuda.synchronize()
g = engine.Graph(ctx, n, out_ptr, out_dst)
nt = synth.topic_sizes(n, kt)
if os.environ.get("NOPOOL", "1") == "1": ctx.set_option("mem.pool_mb", 0)
keep = []
pads = []
if os.environ.get("GAP_MB"): pads.append(torch.empty(int(os.environ["GAP_MB"]) << 20, dtype=torch.uint8, device=dev)); torch.cuda.synchronize()
if os.environ.get("TRACE"): ctx.set_option("pr.trace", 1)
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
