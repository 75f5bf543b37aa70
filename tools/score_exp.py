"""Config-3 batch timing for kernel experiments: kernel ms (HIP events on the library's stream), median of R batches, for the
wave-per-slice kernel (default) and for k_score_slices alone (option score.wave = 0), plus a bit-for-bit comparison of the hits.
    [SS_LIB_PATH=...] [WT=postings per wave slice] [ST=postings per workgroup slice] NQ=1024 K=100 python tools/score_exp.py"""
import os, statistics, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
nd, nt = 10_000_000, 1_000_000
b = synth.zipf_index_torch(nd, nt, 640_000_000, seed=44, device=dev)
t = synth.zipf_index_torch(nd, nt, 40_000_000, seed=144, device=dev)
bi = engine.InvertedIndex(ctx, nd, *b); ti = engine.InvertedIndex(ctx, nd, *t)
del b, t
ti.tfidf_build(nd, False, False, False); bi.tfidf_build(nd, False, False, False)
sc = engine.Scorer(ctx, ti, bi)
k = int(os.environ.get("K", "100"))
nq = int(os.environ.get("NQ", "1024"))
ranks = int(os.environ.get("RANKS", "10000"))
q_ptr, q_terms = synth.make_queries(nq, 3, ranks, seed=45)
dq = (torch.from_numpy(q_ptr.view(np.int32)).to(dev), torch.from_numpy(q_terms.view(np.int32)).to(dev))
outs = {}
for mode in os.environ.get("MODES", "wave,slices").split(","):
    ctx.set_option("score.wave", 1 if mode == "wave" else 0)
    if os.environ.get("WT"): ctx.set_option("score.wave_slice_target", int(os.environ["WT"]))
    for env, opt in (("GP", "score.wave_big_pct"), ("GB", "score.wave_big_x100"), ("GS", "score.wave_small_x100"), ("GSL", "score.grade_slices")):
        if os.environ.get(env): ctx.set_option(opt, int(os.environ[env]))
    if os.environ.get("ST"): ctx.set_option("score.slice_target", int(os.environ["ST"]))
    if os.environ.get("WML"): ctx.set_option("score.wave_min_list", int(os.environ["WML"]))
    if os.environ.get("PIPE"): ctx.set_option("score.pipeline", 1 if mode == "wave" else 0)
    d_hits = torch.empty(nq * k * 40, dtype=torch.uint8, device=dev); d_n = torch.empty(nq, dtype=torch.int32, device=dev)
    ms = []
    for i in range(int(os.environ.get("R", "25"))):
        sc.score_topk(dq[0], dq[1], k, out=(d_hits, d_n))
        if i >= 5: ms.append(ctx.last_kernel_ms(1))
    outs[mode] = (d_hits.cpu().numpy().copy(), d_n.cpu().numpy().copy())
    print(f"mode={mode} lib={os.path.basename(os.environ.get('SS_LIB_PATH', 'product'))} nq={nq} k={k} ranks={ranks}: "
          f"kernels median {statistics.median(ms):.4f} ms  min {min(ms):.4f} ms  ({nq / statistics.median(ms) / 1e3:.3f} M q/s)", flush=True)
if len(outs) == 2:
    (h1, n1), (h2, n2) = outs.values()
    same = np.array_equal(n1, n2) and np.array_equal(h1, h2)
    print("hits identical:", same, flush=True)
    if not same:
        a = h1.view(engine.HIT_DTYPE).reshape(nq, k); bb = h2.view(engine.HIT_DTYPE).reshape(nq, k)
        bad = [q for q in range(nq) if n1[q] != n2[q] or a[q].tobytes() != bb[q].tobytes()]
        print("queries that differ:", len(bad), bad[:10])
        q = bad[0]
        print("n", n1[q], n2[q]); print(a[q][:5]); print(bb[q][:5])
sc.close(); ti.close(); bi.close(); ctx.close()
