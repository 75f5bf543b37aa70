"""End-to-end cost of one UpdateTopicSensitivePagerank call as the caller sees it (start_crawl.go:174-180): device-resident out-edge
CSR -> ss_graph_create -> ss_pagerank_run to convergence, with the phases timed apart (synchronised timers), config 2 and config 4.
    python tools/pr_e2e.py [reps]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
import os
if os.environ.get('TRACE'): ctx.set_option('pr.trace', 1)
if os.environ.get('SNAKE'): ctx.set_option('pr.deal_snake', 1)
if os.environ.get('SHARED_STREAM'):
    _st = torch.cuda.Stream(device=dev); ctx.set_stream(_st.cuda_stream); torch.cuda.set_stream(_st)
for name, n, e, k in (("config2", 1 << 20, 5_000_000, 1), ("config4", 10_000_000, 50_000_000, 16)):
    out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
    n_topic = synth.topic_sizes(n, k)
    torch.cuda.synchronize()
    rank_dev = torch.empty((k, n), dtype=torch.float64, device=dev)
    for eps in (1e-6, 1e-20):
        tc, tp, tr, tt = [], [], [], []
        for r in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            g = engine.Graph(ctx, n, out_ptr, out_dst)
            ctx.synchronize(); t1 = time.perf_counter()
            pr = engine.PageRankState(g, 0.75, eps, n_topic, max_iter=500)
            ctx.synchronize(); t2 = time.perf_counter()
            pr.begin()
            na = k
            while na:
                pr.step(8)
                na = pr.status()["n_active"]
            ctx.synchronize(); t3 = time.perf_counter()
            pr.close(); g.close()
            # the one-call form (what the host mirror and the Go shim use), ranks left on the device
            torch.cuda.synchronize(); t4 = time.perf_counter()
            g = engine.Graph(ctx, n, out_ptr, out_dst)
            check = g.pagerank_dev(0.75, eps, n_topic, rank_dev, max_iter=500) if hasattr(g, "pagerank_dev") else None
            ctx.synchronize(); t5 = time.perf_counter()
            g.close()
            if r:
                tc.append(t1 - t0); tp.append(t2 - t1); tr.append(t3 - t2); tt.append(t5 - t4)
        f = lambda v: f"{1e3 * min(v):.2f}"
        print(f"{name} eps={eps:g}: graph_create {f(tc)} ms, pr_create {f(tp)} ms, begin+sweeps {f(tr)} ms, create+run one call {f(tt)} ms (min of {reps - 1})", flush=True)
    del out_ptr, out_dst, rank_dev
    torch.cuda.empty_cache()
ctx.close()
