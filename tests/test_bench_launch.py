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
