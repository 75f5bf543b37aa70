"""The two-vector form (option pr.affine) against the K-wide sweep at config 4: one ss_pagerank_run from a resident graph to
convergence, ranks left on the device; seconds, iterations, largest relative difference of the ranks.   python tools/pr_affine.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
n, e = 10_000_000, 50_000_000
out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
g = engine.Graph(ctx, n, out_ptr, out_dst)
for kt in (16, 64):
    n_topic = synth.topic_sizes(n, kt)
    rank_a = torch.empty((kt, n), dtype=torch.float64, device=dev)
    rank_b = torch.empty((kt, n), dtype=torch.float64, device=dev)
    for eps in (1e-6, 1e-20):
        res = {}
        for name, opt, out in (("k-wide sweep", 0, rank_a), ("two vectors", 1, rank_b)):
            ctx.set_option("pr.affine", opt)
            ts = []
            for r in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                it = g.pagerank_dev(0.75, eps, n_topic, out, max_iter=500)
                ctx.synchronize(); ts.append(time.perf_counter() - t0)
            res[name] = (min(ts[1:]), np.asarray(it))
            ctx.set_option("pr.affine", None)
        diff = float(((rank_a - rank_b).abs() / rank_a).max())
        (ta, ia), (tb, ib) = res["k-wide sweep"], res["two vectors"]
        print(f"K={kt} eps={eps:g}: k-wide {1e3 * ta:.2f} ms ({int(ia.max())} its), two vectors {1e3 * tb:.2f} ms ({int(ib.max())} its), "
              f"iteration counts equal: {bool((ia == ib).all())}, max rel diff {diff:.1e}", flush=True)
    del rank_a, rank_b
g.close(); ctx.close()
