"""Blended (config 5) scoring: kernel time vs wall time, host vs device topic_probs."""
import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
nd, nt, kt = 10_000_000, 1_000_000, 16
b = synth.zipf_index_torch(nd, nt, 640_000_000, seed=44, device=dev)
t = synth.zipf_index_torch(nd, nt, 40_000_000, seed=144, device=dev)
bi = engine.InvertedIndex(ctx, nd, *b); ti = engine.InvertedIndex(ctx, nd, *t)
del b, t
ti.tfidf_build(nd, False, False, False); bi.tfidf_build(nd, False, False, False)
sc = engine.Scorer(ctx, ti, bi)
k, nq = 100, 1024
q_ptr, q_terms = synth.make_queries(nq, 3, 10_000, seed=45)
dq = (torch.from_numpy(q_ptr.view(np.int32)).to(dev), torch.from_numpy(q_terms.view(np.int32)).to(dev))
d_hits = torch.empty(nq * k * 40, dtype=torch.uint8, device=dev); d_n = torch.empty(nq, dtype=torch.int32, device=dev)
prior = (torch.rand((kt, nd), dtype=torch.float64, device=dev) * 1e-8 + 1e-6)
probs = np.random.default_rng(46).dirichlet(np.ones(kt), size=nq)
d_probs = torch.from_numpy(probs).to(dev)
def run(label, **kw):
    for _ in range(3): sc.score_topk(dq[0], dq[1], k, out=(d_hits, d_n), **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter(); km = 0; R = 20
    for _ in range(R):
        sc.score_topk(dq[0], dq[1], k, out=(d_hits, d_n), **kw); km += ctx.last_kernel_ms(1)
    torch.cuda.synchronize()
    print(f"{label}: wall {(time.perf_counter()-t0)/R*1e3:.3f} ms kernels {km/R:.3f} ms", flush=True)
run("no prior")
sc.set_prior(prior)
run("prior set, no probs")
run("prior + device probs", topic_probs=d_probs)
run("prior + host probs", topic_probs=probs)
