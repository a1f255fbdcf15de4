"""Row-split weights (-sm row) end to end on ONE GPU: GGML_MI355X_SPLIT_FAKE_DEVICES makes csrc/split.cpp treat the device as several
logical devices (own slices, streams, staging buffers), so slicing, the activation broadcast, the per-slice mat-vec launches, the 2-D
gather of the dst slices and the event joins all run; real multi-GPU execution is the driver's to measure."""
import os
os.environ.setdefault("GGML_MI355X_SPLIT_FAKE_DEVICES", "3")          # read once, at the first use of the split buffer type
import numpy as np
import pytest

import oracle as orc
import qdata

pytestmark = pytest.mark.gpu
QTYPES = {"q4_0": 2, "q8_0": 8, "q4_K": 12, "q6_K": 14}


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(np.asarray(b, np.float64)).max() + 1e-30))


@pytest.mark.parametrize("tname", list(QTYPES))
@pytest.mark.parametrize("props", [None, [3, 1, 2]])
def test_mul_mat_over_row_split_weight(ea, gpu, tname, props):
    t = QTYPES[tname]
    rng = np.random.default_rng(41)
    rows, k = 1024 + 128 + 40, 1024                      # last slice ends off the tile grid: it stays in ggml's layout, the others are tiled
    w = qdata.random_blocks(t, rows, k, rng)
    gw = ea.Graph(gpu, ea.USAGE_WEIGHTS, split=(0, props))
    a = gw.tensor(t, k, rows)
    gw.alloc(); gw.set(a, w)
    assert np.array_equal(gw.get(a, np.uint8), w)        # gathered back in ggml's layout
    for T in (1, 6, 24):
        x = rng.standard_normal((T, k)).astype(np.float32)
        g = ea.Graph(gpu)
        b = g.tensor(ea.F32, k, T)
        c = g.mul_mat(a, b)
        d = g.unary(c, "relu")                           # something after the gather, on the caller's stream
        assert g.supports(c)
        g.alloc(); g.set(b, x); g.compute()
        want = orc.mul_mat_q(t, w, x, k, rows)
        assert rel(g.get(c).reshape(T, rows), want) < 2e-5, (tname, T)
        assert rel(g.get(d).reshape(T, rows), np.maximum(want, 0)) < 2e-5


def test_split_weight_is_declined_for_other_ops(ea, gpu):
    gw = ea.Graph(gpu, ea.USAGE_WEIGHTS, split=(0, None))
    a = gw.tensor(ea.F32, 256, 256)
    gw.alloc()
    g = ea.Graph(gpu)
    i = g.tensor(ea.I32, 4)
    assert not g.supports(g.get_rows(a, i))              # only MUL_MAT may read a split tensor (the scheduler then keeps the op off this device)
