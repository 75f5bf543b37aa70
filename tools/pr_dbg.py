"""Which rows of a one-sweep PageRank differ from the oracle, by in-degree (kernel debugging aid)."""
import os, sys
import numpy as np
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
from oracle import pyoracle
n, e, k = int(os.environ.get("N", 20000)), int(os.environ.get("E", 100000)), int(os.environ.get("K", 16))
ptr, dst = synth.rmat_graph(n, e, seed=100 + k)
n_topic = synth.topic_sizes(n, k)
ctx = engine.Context(0)
g = engine.Graph(ctx, n, ptr, dst)
for it in (1, 2):
    rank, iters = g.pagerank(0.75, 0.0, n_topic, max_iter=it)
    ref, _ = pyoracle.pagerank(n, ptr, dst, 0.75, 0.0, n_topic, max_iter=it)
    bad = np.abs(rank - ref) > 1e-12 * np.abs(ref)
    indeg = np.bincount(dst, minlength=n)
    rows = np.where(bad.any(axis=0))[0]
    print(f"iters {it}: bad rows {len(rows)}; in-degree histogram of bad rows:", np.unique(indeg[rows], return_counts=True))
    print("   topics affected:", np.where(bad.any(axis=1))[0].tolist()[:20])
    if len(rows):
        r = rows[0]
        print("   first bad row", r, "indeg", indeg[r], "gpu", rank[:3, r], "ref", ref[:3, r])
# implied inherited sums after one sweep
rank, _ = g.pagerank(0.75, 0.0, n_topic, max_iter=1)
ref, _ = pyoracle.pagerank(n, ptr, dst, 0.75, 0.0, n_topic, max_iter=1)
outdeg = np.diff(ptr.astype(np.int64))
src_of = np.repeat(np.arange(n), outdeg)
x0 = 1.0 / n_topic[0]
w = np.where(outdeg > 0, 0.75 * x0 / np.maximum(outdeg, 1), 0.0)
total = w.sum() + 0.25 * n
indeg = np.bincount(dst, minlength=n)
for r in np.where(np.abs(rank[0] - ref[0]) > 1e-12 * ref[0])[0][:5]:
    y_ref = w[src_of[dst == r]].sum()
    y_gpu = rank[0, r] * total - x0 - 0.25
    print(f"row {r} indeg {indeg[r]}: y_ref {y_ref:.6e} y_gpu {y_gpu:.6e} ratio {y_gpu / y_ref:.6f}  extra/meanw {(y_gpu - y_ref) / (y_ref / indeg[r]):.3f}")
