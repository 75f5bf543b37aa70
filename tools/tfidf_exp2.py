"""TF-IDF experiments on the config-3 body table: the build (k_weight_count + k_scatter + k_bucket_sum), and the magnitude pass alone by
global atomics (ss_index_refresh_magnitudes with the bucketed pass switched off) — device ms between HIP events.
    python tools/tfidf_exp2.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
nd, nt, P = 10_000_000, 1_000_000, int(os.environ.get("P", 640_000_000))
b_ptr, b_doc, b_tf = synth.zipf_index_torch(nd, nt, P, seed=44, device=dev)
for r in range(3):
    bi = engine.InvertedIndex(ctx, nd, b_ptr, b_doc, b_tf.clone())
    bi.tfidf_build(nd, want_w=False, want_mag=False, want_idf=False)
    ctx.synchronize()
    print(f"build: {ctx.last_kernel_ms(2):.2f} ms", flush=True)
    if r == 2:
        ctx.set_option("tfidf.bucket_min", 1 << 40)         # magnitudes by global atomics
        for _ in range(2):
            bi.refresh_magnitudes()
            ctx.synchronize()
            print(f"refresh_magnitudes, global f64 atomics: {ctx.last_kernel_ms(2):.2f} ms", flush=True)
        ctx.set_option("tfidf.bucket_min", None)
        for _ in range(2):
            bi.refresh_magnitudes()
            ctx.synchronize()
            print(f"refresh_magnitudes, bucketed: {ctx.last_kernel_ms(2):.2f} ms", flush=True)
    bi.close()
ctx.close()
