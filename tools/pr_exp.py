"""Config-4 sweep timing for kernel experiments: kernel ms per K-wide sweep (HIP events), median of R blocks of 20 sweeps.
    [SS_LIB_PATH=...] K=16 python tools/pr_exp.py"""
import os, statistics, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
import time
if os.environ.get("PRE_SLEEP"): time.sleep(float(os.environ["PRE_SLEEP"]))     # before anything touches the GPU
dev = torch.device('cuda', 0)
if os.environ.get("RINSE_GB"):      # take (and touch) this much HBM once and give it back before the real allocations
    xs = [torch.empty(8 << 30, dtype=torch.uint8, device=dev) for _ in range(int(os.environ["RINSE_GB"]) // 8)]
    for t in xs: t.fill_(1)
    torch.cuda.synchronize(); del xs, t; torch.cuda.empty_cache(); torch.cuda.synchronize()
ctx = engine.Context(0)
n, e = int(os.environ.get("N", 10_000_000)), int(os.environ.get("E", 50_000_000))
kt = int(os.environ.get("K", "16"))
out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
if os.environ.get("SNAKE"): ctx.set_option("pr.deal_snake", int(os.environ["SNAKE"]))
g = engine.Graph(ctx, n, out_ptr, out_dst)
def run(tag):
    pr = engine.PageRankState(g, 0.75, -1.0, synth.topic_sizes(n, kt), max_iter=0)
    if os.environ.get("TS"):      # opt-in topic-sensitive teleport: topic k teleports to a random 1/16 of the nodes
        rng = np.random.default_rng(3)
        pr.set_teleport([rng.choice(n, size=n // 16, replace=False).astype(np.uint32) for _ in range(kt)])
    pr.begin()
    pr.step(5)
    ms = []
    for _ in range(int(os.environ.get("R", "7"))):
        pr.step(20)
        ctx.synchronize()
        ms.append(ctx.last_kernel_ms(0) / 20)
    probe = [pr.probe(m, 5) for m in (0, 1, 2)] if kt >= 5 and (not tag or os.environ.get("PROBE")) else []
    print(f"lib={os.path.basename(os.environ.get('SS_LIB_PATH', 'product'))} N={n} E={e} K={kt} {tag}: sweep median {statistics.median(ms):.4f} ms  min {min(ms):.4f} ms  probe {probe}", flush=True)
    pr.close()
# OPTSETS="pr.stagger=0;pr.stagger=1,pr.blocks_per_cu=3": one timing per ';'-separated option set (options reset in between)
for oset in os.environ.get("OPTSETS", "").split(";"):
    kv = [x.split("=") for x in oset.split(",") if x]
    for k, v in kv: ctx.set_option(k, int(v))
    run(oset)
    for k, v in kv: ctx.set_option(k, None)
g.close()
for i in range(int(os.environ.get("REBUILD", "0"))):      # the same graph and state allocated again later in the same process
    del out_ptr, out_dst; torch.cuda.empty_cache(); time.sleep(float(os.environ.get("REBUILD_SLEEP", "0")))
    out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
    g = engine.Graph(ctx, n, out_ptr, out_dst); run(f"rebuild {i + 1}"); g.close()
ctx.close()
