"""Per-wave finish times of the last config-4 sweep against the deal's modelled loads (variant build -DSS_PR_WAVETIME):
    tools/build_variant.sh wt -DSS_PR_WAVETIME; SS_LIB_PATH=$PWD/spaghettisearch_amd/libspaghetti_rank_wt.so python tools/pr_wavetime.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
for kv in os.environ.get("OPTS", "").split(","):
    if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
n, e, kt = 10_000_000, 50_000_000, 16
out_ptr, out_dst = synth.rmat_graph_torch(n, e, seed=42, device=dev)
g = engine.Graph(ctx, n, out_ptr, out_dst)
pr = engine.PageRankState(g, 0.75, -1.0, synth.topic_sizes(n, kt), max_iter=0)
pr.begin(); pr.step(5)
pr.step(20); ctx.synchronize(); print("sweep ms", ctx.last_kernel_ms(0) / 20)
pr.close(); g.close(); ctx.close()
wt = np.loadtxt("gpurun_out/pr_wt.csv", delimiter=",", skiprows=1)
ld = np.loadtxt("gpurun_out/pr_load.csv", delimiter=",", skiprows=1)
end = wt[:, 2]; load = ld[:, 1]
print("waves", len(end), "end us: min %.0f mean %.0f max %.0f" % (end.min(), end.mean(), end.max()))
print("modelled load: min %.1f mean %.1f max %.1f" % (load.min(), load.mean(), load.max()))
print("corr(end, load) %.3f" % np.corrcoef(end, load)[0, 1])
blk = (wt[:, 0] // 4).astype(int)
for name, key in (("block % 8 (XCD)", blk % 8), ("wave in block", (wt[:, 0] % 4).astype(int)), ("block // 8 % 32 (CU in XCD)", blk // 8 % 32), ("block // 256 (round)", blk // 256)):
    print(name, " ".join("%d:%.0f" % (k, end[key == k].mean()) for k in np.unique(key)))
# least squares: end ~ sum_c a_c * class cost
A = np.c_[ld[:, 2:8], np.ones(len(end))]
coef, *_ = np.linalg.lstsq(A, end, rcond=None)
print("us per modelled turn by class (V_SEG+ROWW, QUAD, DEG2, DEG4, DEG8, ZERO), constant:", np.round(coef, 3))
print("residual std %.1f us (end std %.1f)" % ((end - A @ coef).std(), end.std()))
