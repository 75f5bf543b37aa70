"""Mixed batches: term ranks U[1,100k], half head / half tail, head only, tail only — wall ms per batch of back-to-back
device-output calls under routing options.  OPTS="a=1,b=2;c=3" = option sets to compare (besides the default and score.wave=0).
    python tools/score_mixed.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
stream = torch.cuda.Stream(device=dev); ctx.set_stream(stream.cuda_stream)
with torch.cuda.stream(stream):
    nd, nt = 10_000_000, 1_000_000
    b = synth.zipf_index_torch(nd, nt, 640_000_000, seed=44, device=dev)
    t = synth.zipf_index_torch(nd, nt, 40_000_000, seed=144, device=dev)
    bi = engine.InvertedIndex(ctx, nd, *b); ti = engine.InvertedIndex(ctx, nd, *t)
    del b, t
    ti.tfidf_build(nd, False, False, False); bi.tfidf_build(nd, False, False, False)
    sc = engine.Scorer(ctx, ti, bi)
    k, nq = 100, 1024
    head = synth.make_queries(nq, 3, 10_000, seed=45)
    mixed = synth.make_queries(nq, 3, 100_000, seed=46)
    tail = synth.make_queries(nq, 3, 1_000_000, seed=47)
    hp, ht = head; tp, tt = tail
    half = (np.concatenate([hp[:nq // 2 + 1], tp[1:nq // 2 + 1] + hp[nq // 2]]).astype(np.uint32), np.concatenate([ht[:hp[nq // 2]], tt[:tp[nq // 2]]]).astype(np.uint32))
    work = {"head U[1,10k]": head, "mixed U[1,100k]": mixed, "half head / half tail": half, "tail U[1,1M]": tail}
    sets = [("default", {}), ("score.wave=0", {"score.wave": 0})]
    for spec in os.environ.get("OPTS", "").split(";"):
        if spec: sets.append((spec, {kv.split("=")[0]: int(kv.split("=")[1]) for kv in spec.split(",")}))
    d_hits = torch.empty(nq * k * 40, dtype=torch.uint8, device=dev); d_n = torch.empty(nq, dtype=torch.int32, device=dev)
    ctx.set_option("score.timing", 0)
    for wname, (qp, qt) in work.items():
        ref = None
        for sname, opts in sets:
            for o, v in opts.items(): ctx.set_option(o, v)
            for _ in range(5): sc.score_topk(qp, qt, k, out=(d_hits, d_n))
            ms = []
            for blk in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20): sc.score_topk(qp, qt, k, out=(d_hits, d_n))
                torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) / 20 * 1e3)
            got = (d_hits.cpu().numpy().copy(), d_n.cpu().numpy().copy())
            if ref is None: ref = got
            same = bool(np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]))
            print(f"{wname:24s} {sname:40s} ms per batch: min {min(ms):.3f} median {sorted(ms)[len(ms) // 2]:.3f}  hits == default: {same}", flush=True)
            for o in opts: ctx.set_option(o, None)
    sc.close(); ti.close(); bi.close()
ctx.set_stream(None); ctx.close()
