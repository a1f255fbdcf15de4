"""Tensor-parallel path (one process per shard) on CPU: world_size 2 over gloo, compute on the reference CPU
backend, checked against the unsharded model.  Shards are slices of the same global tensors (heads / n_ff in whole
quantised blocks), two all-reduces per layer -- exactly the code path bench.py --gpus N runs with RCCL."""
import os
import subprocess
import sys
import numpy as np
import pytest

from conftest import have_ref, ROOT
import refapi

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")


@pytest.mark.parametrize("world", [2])
def test_tp_matches_unsharded(ea, tmp_path, world):
    out = str(tmp_path / "tp.npz")
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tp_worker.py"), str(r), str(world), port, out]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    z = np.load(out)
    be = refapi.reference_cpu(ea, threads=2)
    m = ea.Model(be, "tiny-gqa", "q4_k_m", n_ctx=128, seed=9, predictable=False)
    lg, hid = m.decode(list(range(7, 19)), list(range(12)))
    lg1, hid1 = m.decode([40, 41, 42], [12, 13, 13], seq=[0, 0, 0])
    assert int(z["n_allreduce"]) == 2 * 3 * 2                      # 2 per layer, 3 layers, 2 decodes
    assert int(z["weight_bytes"]) < m.weight_bytes                 # the shard streams less than the whole model
    for a, b in ((z["lg"], lg), (z["hid"], hid), (z["lg1"], lg1), (z["hid1"], hid1)):
        l2 = float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64)))
        assert l2 < 1e-3, l2                                       # same bound as tests/test_model_gpu.py
    # most rows are untouched by any flip: they agree to fp32 summation order
    row_err = np.abs(z["lg"] - lg).max(-1) / np.abs(lg).max()
    assert (row_err < 1e-5).sum() >= len(row_err) // 3
