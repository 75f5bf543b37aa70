"""The reference's own stop threshold (start_crawl.go:175: d=0.75, eps=1e-20): iterations to the floating-point fixed
point, GPU vs oracle (summation order differs: pull vs push)."""
import sys, numpy as np
sys.path.insert(0, '.')
from oracle import pyoracle
from spaghettisearch_amd import engine, synth
ctx = engine.Context(0)
bad = 0
for n, e, seed in ((1000, 5000, 1), (1000, 20000, 2), (30000, 160000, 3), (100000, 500000, 4), (1000000, 5000000, 5), (200000, 3000000, 6)):
    ptr, dst = synth.rmat_graph(n, e, seed=seed)
    for d in (0.75, 0.85):
        n_topic = [n // 2, n, 7, n // 3 + 1]
        ref, rit = pyoracle.pagerank(n, ptr, dst, d, 1e-20, n_topic, max_iter=300)
        g = engine.Graph(ctx, n, ptr, dst)
        rank, it = g.pagerank(d, 1e-20, n_topic, max_iter=300)
        g.close()
        err = float(np.max(np.abs(rank - ref) / ref))
        print(n, e, d, "oracle iters", rit.tolist(), "gpu iters", it.tolist(), "max rel err %.2e" % err, flush=True)
        bad += int(np.max(np.abs(it - rit)) > 1) + int(err > 1e-12)
print("BAD" if bad else "OK")
