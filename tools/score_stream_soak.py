"""Soak of the scorer's stream pipelining on the config-3 index: bursts of batches BACK TO BACK on a stream shared with the caller
(device outputs, no synchronize inside a burst; right behind every call a copy of its outputs is enqueued and the output buffer
is scribbled over), every burst a random sequence of batch kinds — head queries (k_score_wave), tail queries and long queries
(k_score_slices), mixed ranks, SPLIT batches (3-term and 8-term queries in one batch: both kernels side by side) — of random
size and k, in the default mode or with "score.pipeline" = 0; every third burst goes through ss_score_topk_submit / _collect with
up to three batches in flight instead.  The copies must equal, bit for bit, what the same batch gives alone, synchronously, from
k_score_slices ("score.wave" = 0, "score.pipeline" = 0) — the kernel the test suite checks against the oracle.
    SECONDS=240 python tools/score_stream_soak.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from spaghettisearch_amd import engine, synth
dev = torch.device('cuda', 0)
ctx = engine.Context(0)
nd, nt = 10_000_000, 1_000_000
b = synth.zipf_index_torch(nd, nt, 640_000_000, seed=44, device=dev)
t = synth.zipf_index_torch(nd, nt, 40_000_000, seed=144, device=dev)
bi = engine.InvertedIndex(ctx, nd, *b); ti = engine.InvertedIndex(ctx, nd, *t)
del b, t
ti.tfidf_build(nd, False, False, False); bi.tfidf_build(nd, False, False, False)
stream = torch.cuda.Stream(device=dev)
ctx.set_stream(stream.cuda_stream)
sc = engine.Scorer(ctx, ti, bi)
rng = np.random.default_rng(int(os.environ.get("SEED", "11")))
prior = torch.rand((4, nd), dtype=torch.float64, device=dev) * 1e-6


def queries(nq, nterm, lo, hi):
    q_ptr = (np.arange(nq + 1) * nterm).astype(np.uint32)
    return q_ptr, rng.integers(lo, hi, size=nq * nterm).astype(np.uint32)


def make_batch():
    kind = str(rng.choice(["head", "tail", "mixed", "split", "long", "tiny"]))
    nq = int(rng.choice([1, 50, 300, 1024]))
    if kind == "head":
        qp, qt = queries(nq, 3, 0, int(rng.choice([300, 3000])))
    elif kind == "tail":
        qp, qt = queries(nq, 3, 100_000, 1_000_000)
    elif kind == "mixed":
        qp, qt = queries(nq, 3, 0, 100_000)
    elif kind == "long":
        qp, qt = queries(max(1, nq // 4), 8, 0, 3000)
    elif kind == "tiny":
        qp, qt = queries(int(rng.integers(1, 9)), int(rng.integers(1, 4)), 0, 1_000_000)
    else:
        ap, at = queries(max(1, nq // 2), 3, 0, 300)
        bp, bt = queries(max(1, nq // 8), 8, 0, 3000)
        qp = np.concatenate([ap, bp[1:] + ap[-1]]).astype(np.uint32); qt = np.concatenate([at, bt])
        order = rng.permutation(len(qp) - 1)                        # the two kinds interleaved, not one behind the other
        lens = np.diff(qp.astype(np.int64))[order]
        qt = np.concatenate([qt[qp[i]:qp[i + 1]] for i in order]).astype(np.uint32)
        qp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    return kind, qp, qt, int(rng.choice([10, 40, 100]))


t_end = time.time() + float(os.environ.get("SECONDS", "240"))
n_bursts = n_batches = 0
while time.time() < t_end:
    blend = bool(rng.integers(0, 3) == 0)
    sc.set_prior(prior if blend else None)
    burst = [make_batch() for _ in range(int(rng.integers(2, 9)))]
    probs = [rng.dirichlet(np.ones(4), size=len(qp) - 1) if blend else None for _, qp, _, _ in burst]
    mode = int(rng.choice([0, 1, 1, 2]))                                 # 0: score.pipeline off, 1: default, 2: submit / collect
    # (round 5) k_score_small: the default routing (short all-small calls), with the small queries of longer calls staged on an internal
    # stream (one or two table sizes), forced for every query that fits, or off
    small, small_batch = [(2, 0), (2, 0), (2, 1), (2, 2), (1, 0), (0, 0)][int(rng.integers(0, 6))]
    ctx.set_option("score.small", small); ctx.set_option("score.small_batch", small_batch)
    print(f"burst {n_bursts}: mode={mode} small={small}/{small_batch} blend={blend} " + " ".join(f"{kind}:{len(qp) - 1}q/k{k}" for kind, qp, _, k in burst), flush=True)
    got = []
    if mode == 2:
        flight = []
        for i, (_, qp, qt, k) in enumerate(burst):
            if len(flight) == 3:
                j, tk = flight.pop(int(rng.integers(0, 3)))
                got.append((j, *sc.collect(tk)))
            flight.append((i, sc.submit(qp, qt, k, topic_probs=probs[i])))
        for j, tk in flight:
            got.append((j, *sc.collect(tk)))
        got = [(h, n) for _, h, n in sorted(got, key=lambda e: e[0])]
    else:
        with torch.cuda.stream(stream), ctx.options(score__pipeline=None if mode else 0):
            snaps = []
            for i, (_, qp, qt, k) in enumerate(burst):
                nq = len(qp) - 1
                d_h = torch.zeros(nq * k * 40, dtype=torch.uint8, device=dev); d_n = torch.zeros(nq, dtype=torch.int32, device=dev)
                sc.score_topk(qp, qt, k, topic_probs=probs[i], out=(d_h, d_n))
                snaps.append((d_h.clone(), d_n.clone()))
                d_h.fill_(0xEE); d_n.fill_(-7)
            stream.synchronize()
        for (_, qp, _, k), (dh, dn) in zip(burst, snaps):
            nq = len(qp) - 1
            got.append((dh.cpu().numpy().view(engine.HIT_DTYPE).reshape(nq, k), dn.cpu().numpy()))
    ctx.set_option("score.small", None); ctx.set_option("score.small_batch", None)
    with ctx.options(score__wave=0, score__pipeline=0, score__small=0):
        for i, ((kind, qp, qt, k), (h1, n1)) in enumerate(zip(burst, got)):
            h0, n0 = sc.score_topk(qp, qt, k, topic_probs=probs[i])
            if h1.tobytes() != h0.tobytes() or n1.tolist() != n0.tolist():
                bad = [q for q in range(len(qp) - 1) if n1[q] != n0[q] or h1[q].tobytes() != h0[q].tobytes()]
                print(f"   MISMATCH in batch {i} ({kind}), queries {bad[:10]}", flush=True)
                sys.exit(1)
    n_bursts += 1
    n_batches += len(burst)
print(f"stream soak done: {n_bursts} bursts, {n_batches} batches, all identical", flush=True)
sc.close(); ti.close(); bi.close()
ctx.set_stream(None); ctx.close()
