"""Does the TF-IDF build depend on WHEN its memory was freed?  bench.py builds the body table right after the generator's torch arrays
went back to the driver; tools/tfidf_exp.py (torch arrays alive, nothing freed) measures 5.06-5.13 ms, bench.py 5.34-5.62.
Cases, each on a fresh copy of the same table:  A nothing freed before;  B the way bench.py does it (upload, free the torch arrays,
build);  C as B with a wait of WAIT s between the free and the build;  D 40 GB taken and returned right before the upload."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
nd, nt, P = 10_000_000, 1_000_000, 640_000_000
wait = float(os.environ.get("WAIT", "5"))
def gen(): return synth.zipf_index_torch(nd, nt, P, seed=44, device=dev)
def build(bi):
    bi.tfidf_build(nd, want_w=False, want_mag=False, want_idf=False); ctx.synchronize(); return ctx.last_kernel_ms(2)
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    b_ptr, b_doc, b_tf = gen()
    bi = engine.InvertedIndex(ctx, nd, b_ptr, b_doc, b_tf.clone()); a = build(bi); bi.close()
    bi = engine.InvertedIndex(ctx, nd, b_ptr, b_doc, b_tf); del b_ptr, b_doc, b_tf; torch.cuda.empty_cache(); b = build(bi); bi.close()
    b_ptr, b_doc, b_tf = gen()
    bi = engine.InvertedIndex(ctx, nd, b_ptr, b_doc, b_tf); del b_ptr, b_doc, b_tf; torch.cuda.empty_cache(); time.sleep(wait); c = build(bi); bi.close()
    b_ptr, b_doc, b_tf = gen()
    x = [torch.empty(8 << 30, dtype=torch.uint8, device=dev).fill_(1) for _ in range(5)]; torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
    bi = engine.InvertedIndex(ctx, nd, b_ptr, b_doc, b_tf.clone()); d = build(bi); bi.close()
    del b_ptr, b_doc, b_tf; torch.cuda.empty_cache()
    print(f"round {rnd}: A nothing freed {a:.2f} ms   B freed right before (bench.py) {b:.2f}   C freed, {wait:.0f} s wait {c:.2f}   D 40 GB returned before the upload {d:.2f}", flush=True)
ctx.close()
