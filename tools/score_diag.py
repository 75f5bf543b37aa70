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
import os
q_ptr, q_terms = synth.make_queries(nq, 3, int(os.environ.get("RANKS", "10000")), seed=45)
for _ in range(int(os.environ.get('BATCHES', '1'))):
    hits, n = sc.score_topk(q_ptr, q_terms, 100)
print("kernel ms", ctx.last_kernel_ms(1), file=sys.stderr)
sc.close()
ti.close(); bi.close(); ctx.close()
# timeline of the LAST batch's slices (diagnostic build only)
import csv, os
if os.path.exists("gpurun_out/ss_diag_slices.csv"):
    rows = [dict(r) for r in csv.DictReader(open("gpurun_out/ss_diag_slices.csv"))]
    st = np.array([int(r["start"]) for r in rows]); en = np.array([int(r["end"]) for r in rows])
    rec = np.array([int(r["records"]) for r in rows]); win = np.array([int(r["windows"]) for r in rows])
    t0 = st.min(); dur = (en - st) / 100.0       # us (100 MHz realtime counter)
    print(f"slices {len(rows)}  kernel span {(en.max() - t0) / 100.0:.1f} us  sum of durations {dur.sum():.0f} us", file=sys.stderr)
    order = np.argsort(-dur)[:8]
    for i in order:
        print(f"  launch {rows[i]['launch']:>5} start {(st[i]-t0)/100.0:8.1f} us  dur {dur[i]:7.1f} us  windows {win[i]:4d} records {rec[i]:7d}  us/window {dur[i]/max(win[i],1):.2f}", file=sys.stderr)
    edges = np.linspace(0, (en.max() - t0) / 100.0, 21)
    for a, b in zip(edges[:-1], edges[1:]):
        mid = (a + b) / 2
        conc = int(((st - t0) / 100.0 <= mid).sum() - ((en - t0) / 100.0 <= mid).sum())
        print(f"  t={mid:7.1f} us  running slices {conc}", file=sys.stderr)
    big = rec > 200000
    if big.any():
        print(f"  slices > 200k records: {big.sum()}, mean dur {dur[big].mean():.1f} us, mean us/window {(dur[big]/win[big]).mean():.3f}", file=sys.stderr)
    small = rec < 50000
    print(f"  slices < 50k records: {small.sum()}, mean dur {dur[small].mean():.1f} us, mean us/window {(dur[small]/np.maximum(win[small],1)).mean():.3f}", file=sys.stderr)
