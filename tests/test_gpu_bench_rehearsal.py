"""bench.py --gpus N as the driver launches it (torch.distributed.run, one process per rank), rehearsed on ONE GPU:
SS_BENCH_REHEARSAL=1 puts every rank on cuda:0 with gloo and a host-staged exchange in place of RCCL.  The numbers mean nothing;
the test checks that the multi-rank flow of every section runs to the end and that rank 0's JSON line has the shape the driver
parses.  Four ranks, not eight: the GPU boxes admit at most 6 processes on the card at once (this pytest process is one of
them); the 8-rank arithmetic itself is covered in-process by tests/test_gpu_world8.py and on the CPU by the gloo tests."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_four_rank_rehearsal_prints_the_drivers_line():
    world = 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SS_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--nodes", "200000", "--edges", "1000000", "--topics", "16", "--docs", "200000", "--terms", "20000",
           "--body-postings", "2000000", "--title-postings", "200000", "--queries", "64", "--k", "10", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads(lines[0])
    # the contract's keys, in the driver's sense
    assert out["metric"] == "pagerank_iters_per_sec" and out["unit"] == "topic-iterations/s"
    assert out["n_gpus"] == world and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "strong"
    assert out["value"] > 0 and out["ms_per_step"] > 0 and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["dtype"] == "f64" and out["data"] == "synthetic" and "workload" in out["config"]
    assert out.get("valid", True) is True, out.get("invalid_because")
    # the doc-range-sharded sweep with one exchange per sweep is the headline; the other decompositions sit beside it
    dec = out["decompositions"]
    assert dec["doc_range_shards"]["value"] > 0 and dec["topic_shards"]["value"] > 0
    assert dec["doc_range_shards_pipelined"]["matches_unpipelined"] is True
    assert out["exchange"]["rccl_world"] is None and "rehearsal" in out["lib_comm_error"]      # no RCCL in a rehearsal, and the line says so
    assert out["exchange"]["exchange_bytes_per_rank"] > 0
    assert len(out["config"]["to_convergence_eps1e-6"]["iters"]) == 16
    # the retrieval half: query-split replicas (weak scaling) and doc-range shards that reproduce the replica's hits
    tk = out["topk"]
    assert tk["value"] > 0 and tk["scaling"] == "weak" and tk["config"]["parallelism"] == f"query-split replicas x{world}"
    assert tk["doc_sharded"]["matches_unsharded_replica"] is True
    assert tk["blended_config5"]["value"] > 0
    # the digest is the LAST key of the line (the driver keeps the line's tail)
    assert list(out)[-1] == "summary" and lines[0].rstrip().endswith("}}")
    sm = out["summary"]
    assert sm["topk"]["queries_per_s"] == pytest.approx(tk["value"], rel=1e-3) and sm["valid"] is True
    assert len(json.dumps(sm)) < 1900
