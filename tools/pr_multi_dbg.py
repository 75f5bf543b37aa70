import sys
import numpy as np
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
from oracle import pyoracle
ctx = engine.Context(0)
D = 0.75
ptr, dst = synth.rmat_graph(30000, 160000, seed=77)
n = 30000
nt = synth.topic_sizes(n, 1)
ref, ref_it = pyoracle.pagerank(n, ptr, dst, D, 1e-9, nt)
print("oracle iters", ref_it)
for mode in (0, 1):
    ctx.set_option("pr.persistent", mode)
    g = engine.Graph(ctx, n, ptr, dst)
    r, it = g.pagerank(D, 1e-9, nt)
    print("mode", mode, "iters", it, "max rel err", np.max(np.abs(r - ref) / ref))
    for steps in ((1,) * 6, (6,), (2, 4)):
        st = engine.PageRankState(g, D, -1.0, nt, max_iter=0)
        st.begin()
        for m in steps: st.step(m)
        s = st.status()
        x = st.read()
        refk, _ = pyoracle.pagerank(n, ptr, dst, D, -1.0, nt, max_iter=6)
        print("   steps", steps, "sweeps", s["sweeps"], "delta", s.get("delta"), "err vs oracle@6", np.max(np.abs(x - refk) / refk))
        st.close()
    g.close()
ctx.close()
