"""Wall-clock batch rate at config 3 the way bench.py measures it (library on a torch stream shared with the caller, results in
HBM, blocks of 20 back-to-back calls bracketed by synchronize): ms per batch of every block.   python tools/score_wall.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
stream = torch.cuda.Stream(device=dev)
ctx.set_stream(stream.cuda_stream)
with torch.cuda.stream(stream):
    nd, nt = 10_000_000, 1_000_000
    b = synth.zipf_index_torch(nd, nt, 640_000_000, seed=44, device=dev)
    t = synth.zipf_index_torch(nd, nt, 40_000_000, seed=144, device=dev)
    bi = engine.InvertedIndex(ctx, nd, *b); ti = engine.InvertedIndex(ctx, nd, *t)
    del b, t
    ti.tfidf_build(nd, False, False, False); bi.tfidf_build(nd, False, False, False)
    sc = engine.Scorer(ctx, ti, bi)
    k, nq = 100, 1024
    q_ptr, q_terms = synth.make_queries(nq, 3, 10000, seed=45)
    d_hits = torch.empty(nq * k * 40, dtype=torch.uint8, device=dev); d_n = torch.empty(nq, dtype=torch.int32, device=dev)
    for _ in range(3):
        sc.score_topk(q_ptr, q_terms, k, out=(d_hits, d_n))
    import os
    if os.environ.get('NOTIMING'): ctx.set_option('score.timing', 0)
    for kv in os.environ.get('OPTS', '').split(','):
        if kv: ctx.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    ref_hits = None
    if os.environ.get('PIPE'):
        torch.cuda.synchronize(); ref_hits = (d_hits.cpu().numpy().copy(), d_n.cpu().numpy().copy())
        ctx.set_option('score.pipeline', int(os.environ['PIPE']))
    out = []
    for blk in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            sc.score_topk(q_ptr, q_terms, k, out=(d_hits, d_n))
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 20 * 1e3)
    if ref_hits is not None:
        ctx.synchronize(); torch.cuda.synchronize()
        print("one-stream hits identical:", bool(np.array_equal(ref_hits[0], d_hits.cpu().numpy()) and np.array_equal(ref_hits[1], d_n.cpu().numpy())), flush=True)
    print("ms per batch by block:", ["%.3f" % x for x in out], "kernels", "%.3f" % ctx.last_kernel_ms(1), flush=True)
    sc.close(); ti.close(); bi.close()
ctx.set_stream(None); ctx.close()
