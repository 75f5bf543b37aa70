"""bench.py --gpus N started WITHOUT a launcher must start its own ranks (as child processes, before anything touches the
GPU) and pass their exit status on; started by a launcher (WORLD_SIZE set) it must not."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    monkeypatch.syspath_prepend(ROOT)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    return importlib.import_module("bench")


def test_spawn_command(bench, monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    assert bench.spawn_ranks(4) == 0
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_spawns_before_touching_the_gpu_and_relays_the_status(bench, monkeypatch):
    calls = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda n: calls.append(n) or 7)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    mods_before = set(sys.modules)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 7 and calls == [2]
    assert "spaghettisearch_amd.engine" not in set(sys.modules) - mods_before   # the parent never loaded the HIP library


def test_main_does_not_spawn_under_a_launcher(bench, monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(bench, "spawn_ranks", lambda n: pytest.fail("spawned although WORLD_SIZE is set"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--help"])
    with pytest.raises(SystemExit) as ei:        # --help exits from argparse, before any device work
        bench.main()
    assert ei.value.code == 0


def test_profiled_traffic_picks_the_kernel_and_the_dispatch_it_is_asked_for(bench, tmp_path, monkeypatch):
    """bench.py reads a kernel's HBM bytes from the committed PMC passes.  Two selection errors of round 4 must not come back: `k_scatter`
    matched the graph build's `k_scatter_runs` as a substring, and the TF-IDF kernels — two dispatches per run, the 41M-posting title
    table and the 641M-posting body table — were given the MEDIAN of the two, which put the build's traffic below its algorithmic bytes."""
    import json
    rows = []
    for kern, disp, med, mx in (("k_scatter_runs", 27, 10.0, 12.0), ("k_scatter<true, 2>", 2, 1000.0, 1900.0), ("k_pr_sweep<16, false>", 300, 2700.0, 2705.0),
                                ("k_pr_sweep<16, true>", 3, 2900.0, 2900.0), ("k_merge_flat", 1000, 60.0, 70.0)):
        for counter, scale in (("FETCH_SIZE", 1.0), ("WRITE_SIZE", 0.5)):
            rows.append({"counter": counter, "kernel": kern, "dispatches": disp, "median_KB": med * scale, "max_KB": mx * scale})
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r99_pmc_hbm_bytes.json").write_text(json.dumps(rows))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    body = bench.profiled_traffic("k_scatter", pick="max")
    assert body["fetch_size_raw_bytes"] == 1900.0 * 1024 and body["write_size_bytes"] == 950.0 * 1024
    assert body["bytes"] == 2 * 1900.0 * 1024 + 950.0 * 1024           # the read side doubled (gfx950 counts 128-byte requests as 64)
    assert bench.profiled_traffic("k_scatter")["fetch_size_raw_bytes"] == 1000.0 * 1024          # the median when asked for it — of k_scatter, not k_scatter_runs
    assert bench.profiled_traffic("k_pr_sweep<16")["fetch_size_raw_bytes"] == 2700.0 * 1024       # first instance listed: the reference's path
    assert bench.profiled_traffic("k_merge_flat")["dispatch"].startswith("median")
    assert bench.profiled_traffic("k_no_such_kernel") is None
