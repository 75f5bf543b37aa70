"""TF-IDF build timing (config-3 body table): device ms between HIP events, 3 builds on fresh copies of the table, for a list of
(tfidf.blocks, tfidf.bucket_shift[, tfidf.head_min_run]) settings.   CFG="1024:13 2048:13:32" python tools/tfidf_exp.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
nd, nt, P = 10_000_000, 1_000_000, int(os.environ.get("P", 640_000_000))
b_ptr, b_doc, b_tf = synth.zipf_index_torch(nd, nt, P, seed=44, device=dev)
for cfg in os.environ.get("CFG", "1024:13").split():
    blocks, shift, *rest = (int(x) for x in cfg.split(":"))
    ctx.set_option("tfidf.blocks", blocks)
    ctx.set_option("tfidf.bucket_shift", shift)
    ctx.set_option("tfidf.head_min_run", rest[0] if rest else None)
    ms = []
    for r in range(3):
        bi = engine.InvertedIndex(ctx, nd, b_ptr, b_doc, b_tf.clone())
        bi.tfidf_build(nd, want_w=False, want_mag=False, want_idf=False)
        ctx.synchronize()
        ms.append(ctx.last_kernel_ms(2))
        bi.close()
    print(f"blocks={blocks} shift={shift} head_min_run={rest[0] if rest else 'default'} P={b_doc.numel()}: tfidf build ms {['%.2f' % m for m in ms]}", flush=True)
ctx.close()
