"""Counters of one config-3 batch from the diagnostic build (tools/build_diag.sh):
    SS_LIB_PATH=$PWD/spaghettisearch_amd/libspaghetti_rank_diag.so python tools/score_diag.py [--blend]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from spaghettisearch_amd import engine, synth

nd, nt = 10_000_000, 1_000_000
dev = torch.device("cuda", 0)
ctx = engine.Context(0)
b = synth.zipf_index_torch(nd, nt, 640_000_000, seed=44, device=dev)
t = synth.zipf_index_torch(nd, nt, 40_000_000, seed=144, device=dev)
bi = engine.InvertedIndex(ctx, nd, *b)
ti = engine.InvertedIndex(ctx, nd, *t)
del b, t
ti.tfidf_build(nd, False, False, False)
bi.tfidf_build(nd, False, False, False)
t0 = time.time()
sc = engine.Scorer(ctx, ti, bi)
print(f"scorer created in {time.time() - t0:.3f}s", file=sys.stderr)
nq = int(sys.argv[sys.argv.index("--queries") + 1]) if "--queries" in sys.argv else 1024
q_ptr, q_terms = synth.make_queries(nq, 3, 10_000, seed=45)
hits, n = sc.score_topk(q_ptr, q_terms, 100)
print("kernel ms", ctx.last_kernel_ms(1), file=sys.stderr)
sc.close()
ti.close(); bi.close(); ctx.close()
