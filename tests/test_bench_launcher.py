"""`python bench.py --gpus N` must start its own N ranks when WORLD_SIZE is not set (VERDICT r2 missing #5).  Here, without a GPU: the
launcher (bench.spawn_ranks -> torch.distributed.run on 127.0.0.1) starts two ranks of tests/bench_rank_cpu.py -- bench_tp.main_tp, the
N > 1 bench's own round protocol, on the reference CPU backend with gloo -- and relays rank 0's JSON line."""
import json
import os
import subprocess
import sys
import pytest

from conftest import have_ref, ROOT

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")


def test_bench_starts_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "tiny-gqa",
           "--rank-script", os.path.join(ROOT, "tests", "bench_rank_cpu.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks"] == 2 and res["communicator_size"] == 2
    assert "started 2 ranks itself" in res["launcher"]
    assert res["steps"] == 3 and res["value"] > 0 and res["tokens_per_round"] >= 1.0
    assert res["allreduces"] > 0 and res["scaling"] == "strong"


def test_bench_under_an_external_launcher_does_not_spawn(monkeypatch):
    """with WORLD_SIZE in the environment (the driver's torch.distributed.run) bench.py must go straight to its rank code"""
    sys.path.insert(0, ROOT)
    import bench
    called = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda a, v: called.append(1) or 0)
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "0")
    import bench_tp
    seen = []
    monkeypatch.setattr(bench_tp, "main_tp", lambda args, rank, world, local, platform=None: seen.append((rank, world, local)))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    bench.main()
    assert not called and seen == [(0, 2, 0)]
