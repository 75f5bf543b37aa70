"""Every start-class vector of the sweep's arrival rounds ("pr.stagger" >= 10: base-6 digits, round r = digit r), config 4:
    B=4 python tools/pr_stagger_all.py     (B = blocks per CU = rounds; prints the 25 best and the default)"""
import itertools, os, statistics, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
n, e, kt = 10_000_000, 50_000_000, int(os.environ.get("K", "16"))
B = int(os.environ.get("B", "4"))
out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
g = engine.Graph(ctx, n, out_ptr, out_dst)
nt = synth.topic_sizes(n, kt)
def run(stag, reps=2):
    ctx.set_option("pr.stagger", stag)
    ctx.set_option("pr.blocks_per_cu", B)
    pr = engine.PageRankState(g, 0.75, -1.0, nt, max_iter=0)
    pr.begin(); pr.step(3)
    ms = []
    for _ in range(reps):
        pr.step(10); ctx.synchronize(); ms.append(ctx.last_kernel_ms(0) / 10)
    pr.close()
    return min(ms)
res = []
first = [int(x) for x in os.environ.get("FIRST", "0,1,2,3,4,5").split(",")]
for v in itertools.product(range(6), repeat=B):
    if v[0] not in first: continue
    code = 10 + sum(d * 6 ** i for i, d in enumerate(v))
    if code == 10: code = 10 + 6 ** B          # all-zero vector: a digit beyond the rounds keeps the code >= 10 and non-zero
    res.append((run(code), v, code))
    if len(res) % 100 == 0: print(len(res), "vectors,", "best so far", min(res), flush=True)
res.sort()
print(f"B={B} K={kt}: default (round r starts at class r) {run(1, 3):.4f} ms, off {run(0, 3):.4f} ms")
for ms, v, code in res[:25]: print(f"  {ms:.4f} ms  start classes {v}  pr.stagger={code}")
print("  worst:", res[-1])
# the ten best again, longer
for ms, v, code in res[:10]: print(f"  again {run(code, 5):.4f} ms  {v} pr.stagger={code}")
g.close(); ctx.close()
