"""Every order of the sweep's six work classes ("pr.class_order"), config 4, with the stagger off; then every start-position
vector ("pr.stagger") for the best few orders:   python tools/pr_order_all.py"""
import itertools, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
n, e, kt = 10_000_000, 50_000_000, int(os.environ.get("K", "16"))
out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
g = engine.Graph(ctx, n, out_ptr, out_dst)
nt = synth.topic_sizes(n, kt)
def run(order, stag, reps=2):
    ctx.set_option("pr.class_order", order)
    ctx.set_option("pr.stagger", stag)
    pr = engine.PageRankState(g, 0.75, -1.0, nt, max_iter=0)
    pr.begin(); pr.step(3)
    ms = []
    for _ in range(reps):
        pr.step(10); ctx.synchronize(); ms.append(ctx.last_kernel_ms(0) / 10)
    pr.close()
    return min(ms)
res = []
for perm in itertools.permutations(range(6)):
    order = int("".join(map(str, perm)))
    res.append((run(order, 0), order))
res.sort()
print("orders, stagger off:")
for ms, o in res[:20]: print(f"  {ms:.4f} ms  order {o:06d}")
print("  identity", [r for r in res if r[1] == 12345], "worst", res[-1])
best = []
for ms, o in res[:int(os.environ.get("TOP", "4"))]:
    r2 = []
    for v in itertools.product(range(6), repeat=4):
        code = 10 + sum(d * 6 ** i for i, d in enumerate(v))
        if code == 10: code = 10 + 6 ** 4
        r2.append((run(o, code), v, code))
    r2.sort()
    print(f"order {o:06d} (no stagger {ms:.4f}): best start vectors")
    for a in r2[:8]: print(f"    {a[0]:.4f} ms {a[1]} pr.stagger={a[2]}")
    best += [(a[0], o, a[1], a[2]) for a in r2[:5]]
best.sort()
print("again, 5 x 20 sweeps:")
for ms, o, v, code in best[:12]: print(f"  {run(o, code, 5):.4f} ms  order {o:06d} start {v} pr.stagger={code}")
g.close(); ctx.close()
