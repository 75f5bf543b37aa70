"""A plain-C program (what cgo compiles for go/spaghetti) drives the whole hot path through the C ABI on the GPU:
ss_graph_create -> ss_pagerank_run, ss_index_create -> ss_tfidf_build, ss_scorer_create -> ss_score_topk, with host
buffers only — and its outputs are compared with the oracle.  No Python, torch or ctypes in the calling process."""
import os
import subprocess

import numpy as np
import pytest

from spaghettisearch_amd import _lib, engine, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_caller_round_trip(tmp_path, oracle):
    n, e, kt = 30_000, 150_000, 3
    nd, nt, k, nq = n, 2_000, 10, 64
    out_ptr, out_dst = synth.rmat_graph(n, e, seed=21)
    n_topic = synth.topic_sizes(n, kt)
    t = synth.zipf_index(nd, nt, 60_000, seed=22)
    b = synth.zipf_index(nd, nt, 900_000, seed=23)
    q_ptr, q_terms = synth.make_queries(nq, 3, 500, seed=24)
    q_terms[5] = _lib.SS_UNKNOWN_TERM                 # an unknown word (main_retrieve.go:193) and a duplicate token (Q8)
    q_terms[7] = q_terms[6]
    d, eps = 0.75, 1e-9
    src = os.path.join(ROOT, "tests", "c_caller", "roundtrip.c")
    exe = tmp_path / "roundtrip"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", str(exe),
                    "-L", libdir, "-lspaghetti_rank", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        np.array([n, e, kt, nd, nt, len(t[1]), len(b[1]), nq, len(q_terms), k], dtype=np.uint64).tofile(f)
        np.array([d, eps], dtype=np.float64).tofile(f)
        n_topic.astype(np.int32).tofile(f)
        for a in (out_ptr, out_dst, t[0], t[1], t[2], b[0], b[1], b[2], q_ptr, q_terms):
            a.tofile(f)
    r = subprocess.run([str(exe), str(fin), str(fout)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "roundtrip: ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
    with open(fout, "rb") as f:
        rank = np.fromfile(f, np.float64, kt * n).reshape(kt, n)
        iters = np.fromfile(f, np.int32, kt)
        t_w = np.fromfile(f, np.float32, len(t[1]))
        t_mag = np.fromfile(f, np.float64, nd)
        b_w = np.fromfile(f, np.float32, len(b[1]))
        b_mag = np.fromfile(f, np.float64, nd)
        hits = np.fromfile(f, engine.HIT_DTYPE, nq * k).reshape(nq, k)
        n_hits = np.fromfile(f, np.int32, nq)
    ref, ref_it = oracle.pagerank(n, out_ptr, out_dst, d, eps, n_topic)
    assert iters.tolist() == ref_it.tolist()
    np.testing.assert_allclose(rank, ref, rtol=1e-12)
    rtw, rtm, _ = oracle.tfidf(*t, n, nd)
    rbw, rbm, _ = oracle.tfidf(*b, n, nd)
    assert np.array_equal(t_w, rtw) and np.array_equal(b_w, rbw)
    assert np.array_equal(t_mag, rtm) and np.array_equal(b_mag, rbm)
    rh, rn = oracle.score_topk_batch(nd, (t[0], t[1], rtw), (b[0], b[1], rbw), rtm, rbm, q_ptr, q_terms, k)
    assert n_hits.tolist() == rn.tolist()
    assert hits.tobytes() == rh.tobytes()
