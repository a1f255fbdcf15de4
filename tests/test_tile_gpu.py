"""Weight re-layout on the GPU: a quantised weight is permuted into 16-row tiles at its first MUL_MAT
(csrc/tile_layout.h, kernels_tile.hip) and the matrix-core kernel reads the tiles (kernels_mmt.hip) -- but every byte a host
can observe through the ABI (get_tensor, partial get / set) keeps ggml's row-major block layout."""
import numpy as np
import pytest

import oracle as orc
import qdata

pytestmark = pytest.mark.gpu
QTYPES = {"q4_0": 2, "q8_0": 8, "q4_K": 12, "q5_K": 13, "q6_K": 14}


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(np.asarray(b, np.float64)).max() + 1e-30))


@pytest.mark.parametrize("tname", list(QTYPES))
def test_tiled_weight_round_trip_and_partial_io(ea, gpu, tname):
    t = QTYPES[tname]
    rng = np.random.default_rng(31)
    rows, k, T = 64, 1024, 3
    w = qdata.random_blocks(t, rows, k, rng)
    x = rng.standard_normal((T, k)).astype(np.float32)
    g = ea.Graph(gpu, ea.USAGE_WEIGHTS)
    a, b = g.tensor(t, k, rows), g.tensor(ea.F32, k, T)
    c = g.mul_mat(a, b)
    g.alloc(); g.set(a, w); g.set(b, x)
    assert np.array_equal(g.get(a, np.uint8), w)                      # before the first use: plain copy
    g.compute()                                                       # first use: the weight is re-laid out in place
    assert rel(g.get(c).reshape(T, rows), orc.mul_mat_q(t, w, x, k, rows)) < 2e-5
    assert np.array_equal(g.get(a, np.uint8), w)                      # the host still sees ggml's layout
    rb = orc.row_bytes(t, k)
    assert np.array_equal(g.get_at(a, 3 * rb + 10, 2 * rb + 7), w[3 * rb + 10: 5 * rb + 17])
    # partial write into a tiled tensor, then compute again
    w2 = w.copy()
    patch = qdata.random_blocks(t, 3, k, rng)
    w2[5 * rb: 8 * rb] = patch
    g.set_at(a, patch, 5 * rb)
    assert np.array_equal(g.get(a, np.uint8), w2)
    g.compute()
    assert rel(g.get(c).reshape(T, rows), orc.mul_mat_q(t, w2, x, k, rows)) < 2e-5
    assert np.array_equal(g.get(a, np.uint8), w2)
    # whole-tensor overwrite of a tiled tensor
    g.set(a, w)
    g.compute()
    assert rel(g.get(c).reshape(T, rows), orc.mul_mat_q(t, w, x, k, rows)) < 2e-5


@pytest.mark.parametrize("tname", ["q4_K", "q6_K", "q8_0"])
@pytest.mark.parametrize("T", [1, 6, 8])
def test_folded_norm_is_materialised(ea, gpu, tname, T):
    """RMS_NORM -> MUL(w) folded into the mat-vec prologue: the normalised tensor is still written (block 0), so a reader outside
    the graph view finds it (graph.cpp reader counting)."""
    t = QTYPES[tname]
    rng = np.random.default_rng(5)
    rows, k = (256, 4096) if T != 6 else (64, 8192)          # k = 8192: two super-blocks per wave in the norm prologue (70B width)
    w = qdata.random_blocks(t, rows, k, rng)
    x = (rng.standard_normal((T, k)) * 3).astype(np.float32)
    nw = rng.standard_normal(k).astype(np.float32)
    g = ea.Graph(gpu, ea.USAGE_WEIGHTS)
    a, b, nwt = g.tensor(t, k, rows), g.tensor(ea.F32, k, T), g.tensor(ea.F32, k)
    xn = g.mul(g.rms_norm(b, 1e-6), nwt)
    c = g.mul_mat(a, xn)
    g.alloc(); g.set(a, w); g.set(b, x); g.set(nwt, nw); g.compute()
    want_n = orc.rms_norm(x, 1e-6) * nw
    assert rel(g.get(xn).reshape(T, k), want_n) < 2e-6
    assert rel(g.get(c).reshape(T, rows), orc.mul_mat_q(t, w, want_n.astype(np.float32), k, rows)) < 2e-5


@pytest.mark.parametrize("tname", list(QTYPES))
def test_tiled_kernel_token_counts_and_long_k(ea, gpu, tname):
    """token passes (1..8 in-kernel quantiser, 9..24 token groups through the HBM image, > 24 several passes), k-chunks when one LDS
    image cannot hold k, fused residual"""
    t = QTYPES[tname]
    rng = np.random.default_rng(17)
    for rows, k, T in [(16, 256, 1), (32, 768, 2), (48, 4096, 7), (64, 8192, 4), (32, 8192, 5), (16, 11008 // 256 * 256, 6), (48, 2048, 9), (32, 1024, 24), (16, 512, 61),
                       (16, 28672, 6), (16, 11264, 24)]:
        w = qdata.random_blocks(t, rows, k, rng)
        x = rng.standard_normal((T, k)).astype(np.float32)
        res = rng.standard_normal((T, rows)).astype(np.float32)
        want = orc.mul_mat_q(t, w, x, k, rows)
        for r in (None, res):
            g = ea.Graph(gpu, ea.USAGE_WEIGHTS)
            a, b = g.tensor(t, k, rows), g.tensor(ea.F32, k, T)
            c = g.mul_mat(a, b); out = c
            if r is not None:
                rt = g.tensor(ea.F32, rows, T); out = g.add(c, rt)
            g.alloc(); g.set(a, w); g.set(b, x)
            if r is not None:
                g.set(rt, r)
            g.compute()
            assert rel(g.get(out).reshape(T, rows), want + (0 if r is None else r)) < 2e-5, (tname, rows, k, T, r is not None)


@pytest.mark.parametrize("tname", ["q4_K", "q6_K", "q8_0", "q5_K", "q4_0"])
def test_big_batch_one_pass_kernel_full_7b_rows(ea, gpu, tname):
    """a3: prompts / wide tree verification (> 24 tokens) take k_mmt_ts -- waves split the tokens, k is cut into LDS-sized chunks that
    accumulate through the residual input -- at the full row counts of the 7B matrices; checked on row slices by the restatement"""
    t = QTYPES[tname]
    rng = np.random.default_rng(77)
    shapes = [(4096, 4096, 61), (4096, 4096, 128), (11008, 4096, 70), (4096, 11008, 61), (4096, 11008, 128), (512, 1024, 25), (256, 8192, 150)]
    if tname != "q4_K":                                  # the other types: one full-size matrix, the small / ragged-token cases
        shapes = [(4096, 4096, 61), (512, 1024, 25), (256, 8192, 150)] + ([(4096, 11008, 128)] if tname in ("q6_K", "q8_0") else [])
    for rows, k, T in shapes:
        w = qdata.random_blocks(t, rows, k, rng)
        x = rng.standard_normal((T, k)).astype(np.float32)
        res = rng.standard_normal((T, rows)).astype(np.float32)
        rb = orc.row_bytes(t, k)
        sel = np.r_[0:32, rows // 2:rows // 2 + 16, rows - 32:rows]
        wsel = np.concatenate([w[r * rb:(r + 1) * rb] for r in sel])
        want = orc.mul_mat_q(t, wsel, x, k, len(sel))
        for r in (None, res):
            g = ea.Graph(gpu, ea.USAGE_WEIGHTS)
            a, b = g.tensor(t, k, rows), g.tensor(ea.F32, k, T)
            c = g.mul_mat(a, b); out = c
            if r is not None:
                rt = g.tensor(ea.F32, rows, T); out = g.add(c, rt)
            g.alloc(); g.set(a, w); g.set(b, x)
            if r is not None:
                g.set(rt, r)
            g.compute()
            got = g.get(out).reshape(T, rows)[:, sel]
            assert rel(got, want + (0 if r is None else r[:, sel])) < 2e-5, (tname, rows, k, T, r is not None)


@pytest.mark.parametrize("tname", ["q4_K", "q6_K", "q8_0"])
@pytest.mark.parametrize("ids", [[0], [4, 1, 5], [0, 1, 2, 3, 4, 5], [5, 5, 2]])
def test_output_row_selection_in_the_epilogue(ea, gpu, tname, ids):
    """The last layer's cur = get_rows(wo x, inp_out_ids); inpSA = get_rows(inpSA, inp_out_ids); add(cur, inpSA)
    (R/src/llama.cpp build_llama / build_eagle) runs inside the output projection's epilogue (graph.cpp plan_member, M.ids)."""
    t = QTYPES[tname]
    rng = np.random.default_rng(77)
    rows, k, T = 256, 1024, 6
    ids = [i for i in ids if i < T]
    w = qdata.random_blocks(t, rows, k, rng)
    x = rng.standard_normal((T, k)).astype(np.float32)
    r = rng.standard_normal((T, rows)).astype(np.float32)
    g = ea.Graph(gpu, ea.USAGE_WEIGHTS)
    a, b, res, ix = g.tensor(t, k, rows), g.tensor(ea.F32, k, T), g.tensor(ea.F32, rows, T), g.tensor(ea.I32, len(ids))
    mm = g.mul_mat(a, b)
    out = g.add(g.get_rows(mm, ix), g.get_rows(res, ix))
    g.alloc(); g.set(a, w); g.set(b, x); g.set(res, r); g.set(ix, np.asarray(ids, np.int32))
    for _ in range(2):                                                # first run re-lays the weight out, second runs on tiles
        g.compute()
        want = (orc.mul_mat_q(t, w, x, k, rows) + r)[ids]
        assert rel(g.get(out).reshape(len(ids), rows), want) < 2e-5


def test_argmax_feeds_get_rows_in_one_launch(ea, gpu):
    """GET_ROWS(token_embd f16, ARGMAX(logits)) -- the draft chain's token -> embedding hand-off -- is one launch (graph.cpp GGML_OP_ARGMAX)."""
    rng = np.random.default_rng(78)
    V, E, T = 32000, 512, 3
    logits = rng.standard_normal((T, V)).astype(np.float32)
    logits[1, 777] = logits[1, 31999] = 9.0                           # a tie: the LAST index wins (ggml_vec_argmax_f32, R/ggml/src/ggml-cpu/ggml-cpu.c:2253)
    tab = rng.standard_normal((V, E)).astype(np.float16)
    g = ea.Graph(gpu)
    lg, tb = g.tensor(ea.F32, V, T), g.tensor(ea.F16, E, V)
    am = g.argmax(lg)
    rows = g.get_rows(tb, am)
    g.alloc(); g.set(lg, logits); g.set(tb, tab)
    g.compute()
    want = logits.argmax(axis=1)
    want[1] = 31999
    assert np.array_equal(g.get(am, np.int32).reshape(-1), want.astype(np.int32))
    assert np.array_equal(g.get(rows).reshape(T, E), tab[want].astype(np.float32))
