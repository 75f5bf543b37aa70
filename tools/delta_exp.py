"""ss_index_apply_delta on the config-3 body table: wall time of a re-crawl-sized delta (5000 changed docs, 100k new postings)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
nd, nt, P = 10_000_000, 1_000_000, 640_000_000
b_ptr, b_doc, b_tf = synth.zipf_index_torch(nd, nt, P, seed=44, device=dev)
bi = engine.InvertedIndex(ctx, nd, b_ptr, b_doc, b_tf)
del b_ptr, b_doc, b_tf
bi.tfidf_build(nd, False, False, False)
rng = np.random.default_rng(1)
for rnd in range(3):
    changed = rng.choice(nd, size=5000, replace=False).astype(np.uint32)
    at = rng.integers(0, nt, size=100_000).astype(np.uint32)
    ad = changed[rng.integers(0, len(changed), size=100_000)]
    key = np.unique((at.astype(np.uint64) << np.uint64(32)) | ad.astype(np.uint64))
    at, ad = (key >> np.uint64(32)).astype(np.uint32), (key & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    aw = rng.random(len(at), dtype=np.float32) + np.float32(0.01)
    ctx.synchronize(); t0 = time.perf_counter()
    bi.apply_delta(del_docs=changed, add=(at, ad, aw))
    ctx.synchronize(); t1 = time.perf_counter()
    bi.refresh_magnitudes()
    ctx.synchronize(); t2 = time.perf_counter()
    print(f"round {rnd}: table {bi.n_post} postings; apply_delta {1e3 * (t1 - t0):.1f} ms, refresh_magnitudes {1e3 * (t2 - t1):.1f} ms (incl. 80 MB read-back)", flush=True)
bi.close(); ctx.close()
