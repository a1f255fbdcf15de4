"""The tensor-parallel code path on ONE GPU (promoted from round 1's gpurun_out/ rehearsal): EH_FORCE_TP=1 makes bench.py take the
tensor-parallel decode with a 1-rank RCCL communicator -- the 2 all-reduces per layer enqueued on the plugin's stream from inside
graph_compute (node hooks -> host/tp.cpp), draft / accept broadcast -- exactly what `bench.py --gpus N` runs per rank.  Multi-GPU numbers are the driver's to measure."""
import json
import os
import subprocess
import sys
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(extra_env):
    env = dict(os.environ, EH_FORCE_TP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_one_rank_rccl_rehearsal_of_the_tp_path():
    d = _run({})
    assert d["n_gpus"] == 1 and d["value"] > 100 and d["unit"] == "tokens/s"
    assert d["allreduces"] >= 64 * 8                                     # 2 per layer x 32 layers per target forward, >= 8 forwards
    assert d["communicator_size"] == 1 and d["ranks"] == 1               # as RCCL reports it (ncclCommCount)
    assert "inside graph_compute" in d["allreduce_submission"]           # the plugin's node hooks enqueue ncclAllReduce: one submission per forward
    # the segment loop (one graph_compute per all-reduce, round 2's path) must draft / accept exactly the same
    s = _run({"EH_TP_SEGMENTS": "1"})
    assert "segments" in s["allreduce_submission"]
    assert (s["tokens_per_round"], s["accept_rate"], s["allreduces"]) == (d["tokens_per_round"], d["accept_rate"], d["allreduces"])
