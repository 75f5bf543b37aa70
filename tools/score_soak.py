"""Soak of the two scoring kernels against each other on the config-3 index: random batches (query count, terms per query, term
rank range, k, wave slice size, blend on/off), k_score_wave forced wherever it is allowed ("score.wave_min_list" = 0) against
k_score_slices ("score.wave" = 0): the hits must agree bit for bit.  Prints one line per batch BEFORE it runs (a batch that
never returns names itself) and a summary.      SECONDS=240 python tools/score_soak.py"""
import os, sys, time
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
rng = np.random.default_rng(int(os.environ.get("SEED", "7")))
prior = (torch.rand((4, nd), dtype=torch.float64, device=dev) * 1e-6)
t_end = time.time() + float(os.environ.get("SECONDS", "240"))
n_batches = n_wave = 0
while time.time() < t_end:
    nq = int(rng.choice([1, 7, 64, 300, 1024]))
    ranks = int(rng.choice([100, 300, 1000, 3000, 10000, 30000, 200000]))
    nterm = int(rng.choice([1, 2, 3, 3, 3, 5, 9, 12]))
    k = int(rng.choice([1, 10, 50, 100, 128, 200]))
    wt = rng.choice([None, 2048, 8192, 20000, 49152])
    wml = int(rng.choice([0, 0, 16]))
    blend = bool(rng.integers(0, 2))
    dup = bool(rng.integers(0, 4) == 0)
    lens = np.full(nq, nterm)
    q_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    q_terms = rng.integers(0, ranks, size=int(lens.sum())).astype(np.uint32)
    if dup and nterm > 1:
        q_terms[1::nterm] = q_terms[0::nterm]                      # a duplicate token in every query
    probs = rng.dirichlet(np.ones(4), size=nq) if blend else None
    sc.set_prior(prior if blend else None)
    print(f"batch {n_batches}: nq={nq} terms={nterm} ranks<{ranks} k={k} wave_slice={wt} wave_min_list={wml} blend={blend} dup={dup}" + (" pipelined" if False else ""), flush=True)
    pipe = bool(rng.integers(0, 3) == 0)                               # device outputs, merge on the merge stream ("score.pipeline")
    with ctx.options(score__wave_min_list=wml, score__wave_max_terms=12 if wml == 0 else None, score__wave_slice_target=None if wt is None else int(wt),
                     score__pipeline=1 if pipe else None):
        if pipe:
            d_h = torch.zeros(nq * k * 40, dtype=torch.uint8, device=dev); d_n = torch.zeros(nq, dtype=torch.int32, device=dev)
            sc.score_topk(q_ptr, q_terms, k, topic_probs=probs, out=(d_h, d_n))          # (own stream: the binding synchronises)
            h1 = d_h.cpu().numpy().view(engine.HIT_DTYPE).reshape(nq, k); n1 = d_n.cpu().numpy()
        else:
            h1, n1 = sc.score_topk(q_ptr, q_terms, k, topic_probs=probs)
        ms1 = ctx.last_kernel_ms(1)
    with ctx.options(score__wave=0):
        h0, n0 = sc.score_topk(q_ptr, q_terms, k, topic_probs=probs)
        ms0 = ctx.last_kernel_ms(1)
    same = h1.tobytes() == h0.tobytes() and n1.tolist() == n0.tolist()
    print(f"   wave-allowed {ms1:.3f} ms, slices {ms0:.3f} ms, identical {same}", flush=True)
    if not same:
        bad = [q for q in range(nq) if n1[q] != n0[q] or h1[q].tobytes() != h0[q].tobytes()]
        print("   MISMATCH in queries", bad[:10], "terms", q_terms[q_ptr[bad[0]]:q_ptr[bad[0] + 1]].tolist(), flush=True)
        sys.exit(1)
    n_batches += 1
    n_wave += ms1 != ms0
print(f"soak done: {n_batches} batches, all identical", flush=True)
sc.close(); ti.close(); bi.close(); ctx.close()
