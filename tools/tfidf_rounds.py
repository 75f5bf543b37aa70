"""Is k_bucket_sum's time quantised in rounds of resident workgroups (two per CU = 512 slots)?  The same 640M postings over 8 388 608 docs
(1024 buckets of 8192 = 2.0 rounds), 10 000 000 (1221 = 2.38 -> 3 rounds) and 12 582 912 (1536 = 3.0 rounds): build ms, three builds each.
Under rocprofv3 (tools/kt.sh) the kernel trace gives the kernels' own times."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
nt, P = 1_000_000, 640_000_000
for nd in (8_388_608, 10_000_000, 12_582_912):
    b_ptr, b_doc, b_tf = synth.zipf_index_torch(nd, nt, P, seed=44, device=dev)
    ms = []
    for r in range(3):
        bi = engine.InvertedIndex(ctx, nd, b_ptr, b_doc, b_tf.clone())
        bi.tfidf_build(nd, want_w=False, want_mag=False, want_idf=False); ctx.synchronize()
        ms.append(ctx.last_kernel_ms(2)); bi.close()
    print(f"n_docs={nd} buckets={-(-nd // 8192)} rounds={-(-nd // 8192) / 512:.2f} P={b_doc.numel()}: build ms {['%.2f' % m for m in ms]}", flush=True)
    del b_ptr, b_doc, b_tf; torch.cuda.empty_cache()
ctx.close()
