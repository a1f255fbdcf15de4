"""Parity tests proper: every hot-path op through the plugin's C ABI (vtables -> graph_compute) on a real
MI355X, checked against (1) the real reference CPU backend running the SAME graph through the same ABI
(oracle/_ref), (2) the CPU restatement (oracle/), (3) the committed golden vectors.

Tolerances (float paths; integer parts -- quantised activations, int32 dot products -- are identical by
construction, see kernels_mmvq.hip):  logits-style outputs 1e-3 relative (BASELINE.json north_star),
tightened here to what the arithmetic allows: mat-vec 2e-5, element-wise 2e-6 of the output scale.
"""
import os
import subprocess
import numpy as np
import pytest

import oracle as orc
import qdata
from conftest import have_ref, ROOT

pytestmark = pytest.mark.gpu
QTYPES = {"q4_0": 2, "q8_0": 8, "q4_K": 12, "q5_K": 13, "q6_K": 14}
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(np.asarray(b, np.float64)).max() + 1e-30))


def gpu_mul_mat(ea, gpu, t, w, x, k, rows, residual=None, usage=0):
    T = x.shape[0]
    g = ea.Graph(gpu, usage)
    a, b = g.tensor(t, k, rows), g.tensor(ea.F32, k, T)
    c = g.mul_mat(a, b)
    out = c
    if residual is not None:
        r = g.tensor(ea.F32, rows, T)
        out = g.add(c, r)
    g.alloc(); g.set(a, w); g.set(b, x)
    if residual is not None:
        g.set(r, residual)
    g.compute()
    return g.get(out).reshape(T, rows)


@pytest.mark.parametrize("tname", list(QTYPES))
def test_mul_mat_golden_vectors(ea, gpu, tname):
    z = np.load(os.path.join(GOLD, f"mul_mat_{tname}.npz"))
    k, rows = int(z["k"]), int(z["rows"])
    for T in (1, 2, 3, 5, 8):
        got = gpu_mul_mat(ea, gpu, QTYPES[tname], z["w"], z["x"][:T], k, rows)
        assert rel(got, z["dst_scalar"][:T]) < 2e-5, (tname, T)
        assert rel(got, z["dst_avx2"][:T]) < 2e-5, (tname, T)


@pytest.mark.parametrize("tname", list(QTYPES))
@pytest.mark.parametrize("shape", [(4096, 4096), (11008, 4096), (4096, 11008)])   # (rows, k): Vicuna-7B q/k/v/o, gate/up, down
@pytest.mark.parametrize("T", [1, 4, 6, 8])
def test_mul_mat_model_shapes_vs_oracle(ea, gpu, tname, shape, T):
    """full 7B layer shapes, random valid blocks, checked on a slice of rows by the restatement"""
    rows, k = shape
    if tname in ("q4_0", "q8_0") and T not in (1, 6):
        pytest.skip("covered by T=1,6")
    t = QTYPES[tname]
    rng = np.random.default_rng(hash((tname, rows, k, T)) % 2**32)
    w = qdata.random_blocks(t, rows, k, rng)
    x = rng.standard_normal((T, k)).astype(np.float32)
    got = gpu_mul_mat(ea, gpu, t, w, x, k, rows)
    rb = orc.row_bytes(t, k)
    sel = np.r_[0:48, rows//2:rows//2 + 16, rows - 48:rows]
    wsel = np.concatenate([w[r*rb:(r+1)*rb] for r in sel])
    want = orc.mul_mat_q(t, wsel, x, k, len(sel))
    assert rel(got[:, sel], want) < 2e-5


@pytest.mark.parametrize("tname", ["q4_K", "q6_K", "q8_0", "q4_0", "q5_K"])
def test_mul_mat_ragged_and_fused_residual(ea, gpu, tname):
    """rows not a multiple of the rows-per-block, k with a ragged last k-step, fused ADD epilogue, > 8 tokens"""
    t = QTYPES[tname]
    rng = np.random.default_rng(7)
    # T > 8: K-quants take up to 24 tokens per pass (3 groups of 8 on the matrix cores): whole groups, ragged groups, several passes
    for rows, k, T in [(1, 256, 1), (7, 768, 2), (33, 2816, 3), (130, 5120, 8), (50, 1024, 19), (40, 1024, 9), (48, 2048, 24), (21, 512, 61), (64, 4096, 128),
                       (16, 11008, 24), (16, 11008, 61), (8, 28672, 6)]:      # k too long for one LDS image: k-chunked passes chained through the residual
        w = qdata.random_blocks(t, rows, k, rng)
        x = rng.standard_normal((T, k)).astype(np.float32)
        res = rng.standard_normal((T, rows)).astype(np.float32)
        want = orc.mul_mat_q(t, w, x, k, rows)
        assert rel(gpu_mul_mat(ea, gpu, t, w, x, k, rows), want) < 2e-5, (rows, k, T)
        assert rel(gpu_mul_mat(ea, gpu, t, w, x, k, rows, residual=res), want + res) < 2e-5, (rows, k, T)


def ref_argmax(x):
    """ggml_vec_argmax_f32 (R/ggml/src/ggml-cpu/ggml-cpu.c:2253-2261) word for word: the LAST of equal maxima wins, a NaN resets the running maximum"""
    out = []
    for row in np.asarray(x, np.float32):
        mx = np.float32(-np.inf); idx = 0
        for i, v in enumerate(row):
            mx = mx if mx > v else v
            if mx == v:
                idx = i
        out.append(idx)
    return np.asarray(out, np.int32)


def test_argmax_follows_the_reference_tie_rule(ea, gpu, ref_cpu):
    """GGML_OP_ARGMAX (greedy draft / verify steps fetch one int per token) against the reference CPU backend and its loop restated above:
    ties (last index wins -- NOT numpy's first), a maximum in the last element, rows of -inf, rows with NaNs"""
    rng = np.random.default_rng(12)
    for ne0, rows in [(32000, 6), (100, 5), (1, 2), (4099, 3)]:
        x = rng.standard_normal((rows, ne0)).astype(np.float32)
        if ne0 > 50:
            x[0, 7] = x[0, 41] = 9.0          # tie: the later one wins
            x[1, ne0 - 1] = 11.0              # maximum in the last element
            x[2, :] = -np.inf                 # every element equals the running maximum: n - 1
            if rows > 3:
                x[3, 5] = np.nan; x[3, 3] = 50.0          # the NaN forgets the 50 in front of it
            if rows > 4:
                x[4, ne0 - 1] = np.nan; x[4, 17] = 40.0   # a trailing NaN leaves the index where it was
        want = ref_argmax(x)
        got = {}
        for name, be in (("gpu", gpu), ("ref", ref_cpu)):
            g = ea.Graph(be)
            a = g.tensor(ea.F32, ne0, rows); r = g.argmax(a)
            g.alloc(); g.set(a, x); g.compute()
            got[name] = g.get(r, np.int32).reshape(-1)
        assert np.array_equal(got["ref"], want), (ne0, rows, "restatement vs reference")
        assert np.array_equal(got["gpu"], want), (ne0, rows)
    assert ref_argmax(np.asarray([[1.0, 3.0, 3.0, 2.0]]))[0] == 2


def test_device_top_k_matches_host_candidates(ea, gpu, ref_cpu):
    """The plugin's top-k extension (tree drafting, SURVEY 8 f1: R/common/speculative.cpp:257-272 takes the k best candidates per live branch)
    against the host rule of host/tree_driver.cpp `candidates`: descending logit, ties -> LOWER id first.  Rows with ties across threads and
    waves, -inf entries, fewer finite entries than k, k = 1 / 40 / 64, a row subset, vocabularies of 32000 and 50257 (two register tiers)."""
    rng = np.random.default_rng(21)
    def host_topk(row, k):
        order = np.lexsort((np.arange(row.size), -row.astype(np.float64)))          # value descending, then index ascending
        return order[:k].astype(np.int32), row[order[:k]]
    for ne0, rows in [(32000, 10), (50257, 3), (777, 4), (40, 2)]:
        x = rng.standard_normal((rows, ne0)).astype(np.float32)
        x[0, [5 % ne0, 1029 % ne0, 2053 % ne0, 31 % ne0, 700 % ne0]] = 7.5          # one maximum in several threads / waves: ids ascending
        if rows > 1: x[1, :] = -np.inf; x[1, [3, 9]] = [1.0, 2.0]          # two finite entries, then -inf in index order
        if rows > 2: x[2, ne0 - 1] = 99.0; x[2, 0] = 99.0
        g = ea.Graph(gpu); a = g.tensor(ea.F32, ne0, rows); b = g.scale(a, 1.0)
        g.alloc(); g.set(a, x); g.compute()
        for k in (1, 8, 40, 64):
            if k > ne0: continue
            ids, vals = g.top_k(b, k)
            for r in range(rows):
                wi, wv = host_topk(x[r], k)
                assert np.array_equal(ids[r], wi), (ne0, r, k, ids[r][:8], wi[:8])
                assert np.array_equal(vals[r], wv), (ne0, r, k)
        sub = [rows - 1, 0]
        ids, vals = g.top_k(b, 5, rows=sub)
        for j, r in enumerate(sub):
            assert np.array_equal(ids[j], host_topk(x[r], 5)[0])
    # the reference CPU backend has no such extension: the host keeps its own path
    g = ea.Graph(ref_cpu); a = g.tensor(ea.F32, 64, 2); g.alloc()
    assert g.top_k(a, 4) is None


def test_activation_edge_cases(ea, gpu):
    """all-zero activations, a zero super-block, +-max ties: quantised image must follow the CPU rule"""
    rng = np.random.default_rng(8)
    k, rows = 1024, 16
    w = qdata.random_blocks(12, rows, k, rng)
    x = rng.standard_normal((4, k)).astype(np.float32)
    x[0] = 0
    x[1, 256:512] = 0
    x[2, 10] = 5.0; x[2, 20] = -5.0          # first of equal magnitudes decides the sign of the scale
    x[3, 700] = -7.0; x[3, 701] = 7.0
    assert rel(gpu_mul_mat(ea, gpu, 12, w, x, k, rows), orc.mul_mat_q(12, w, x, k, rows)) < 2e-5
    assert np.all(gpu_mul_mat(ea, gpu, 12, w, x[:1], k, rows) == 0)


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_reference_conformance_suite(ea):
    """the reference's OWN backend conformance test (tests/test-backend-ops.cpp, built unmodified under
    oracle/_ref) run against our plugin via GGML_BACKEND_PATH -- every op case it knows for our device"""
    env = dict(os.environ, GGML_BACKEND_PATH=ea.require_plugin())
    out = subprocess.run(["./test-backend-ops", "test", "-b", "MI355X0"], cwd=os.path.join(ROOT, "oracle", "_ref"),
                         env=env, capture_output=True, text=True, timeout=900)
    txt = out.stdout + out.stderr
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "test_backend_ops.log"), "w").write(txt)
    assert out.returncode == 0, txt[-3000:]
    assert "Backend MI355X0" in txt and "FAIL" not in txt
    n_ok = sum(1 for l in txt.splitlines() if l.rstrip().endswith("OK") or "\x1b[1;32mOK" in l)
    assert n_ok > 400


def _both(ea, gpu, ref_cpu, build, inputs):
    outs = []
    for be in (gpu, ref_cpu):
        g = ea.Graph(be)
        tens, out = build(g)
        g.alloc()
        for t, v in zip(tens, inputs):
            g.set(t, v)
        g.compute()
        outs.append(g.get(out).copy())
    return outs


def test_small_ops_vs_reference_backend(ea, gpu, ref_cpu):
    rng = np.random.default_rng(9)
    # RMS_NORM + MUL (fused on the GPU)
    x = rng.standard_normal((6, 4096)).astype(np.float32); wv = rng.standard_normal(4096).astype(np.float32)
    def b1(g):
        a = g.tensor(ea.F32, 4096, 6); w = g.tensor(ea.F32, 4096); return [a, w], g.mul(g.rms_norm(a, 1e-6), w)
    got, want = _both(ea, gpu, ref_cpu, b1, [x, wv]); assert rel(got, want) < 2e-6
    assert rel(got.reshape(6, 4096), orc.rms_norm(x, 1e-6) * wv) < 2e-6
    # ROPE with tree positions (siblings share a position)
    q = rng.standard_normal((6, 32, 128)).astype(np.float32); pos = np.array([128, 129, 129, 130, 130, 131], np.int32)
    def b2(g):
        a = g.tensor(ea.F32, 128, 32, 6); p = g.tensor(ea.I32, 6); return [a, p], g.rope(a, p, 128, 0)
    got, want = _both(ea, gpu, ref_cpu, b2, [q, pos]); assert rel(got, want) < 2e-6
    assert rel(got.reshape(6, 32, 128), orc.rope(q, pos, 128, 0)) < 2e-6
    # SwiGLU tail (SILU + MUL fused)
    ga = rng.standard_normal((6, 11008)).astype(np.float32) * 3; up = rng.standard_normal((6, 11008)).astype(np.float32)
    def b3(g):
        a = g.tensor(ea.F32, 11008, 6); u = g.tensor(ea.F32, 11008, 6); return [a, u], g.mul(g.unary(a, "silu"), u)
    got, want = _both(ea, gpu, ref_cpu, b3, [ga, up]); assert rel(got, want) < 2e-6
    # EAGLE head front: concat([embd ; hidden]) along dim 0, then ReLU (R/src/llama.cpp:1863-1869)
    e = rng.standard_normal((3, 4096)).astype(np.float32); hdn = rng.standard_normal((3, 4096)).astype(np.float32)
    def b4(g):
        a = g.tensor(ea.F32, 4096, 3); b = g.tensor(ea.F32, 4096, 3); return [a, b], g.unary(g.concat(a, b, 0), "relu")
    got, want = _both(ea, gpu, ref_cpu, b4, [e, hdn]); assert np.array_equal(got, want)
    # KV store: f32 -> f16 K rows and the transposed V write (R/src/llama.cpp:228-270)
    kcur = rng.standard_normal((5, 4096)).astype(np.float32)
    def b5(g):
        a = g.tensor(ea.F32, 4096, 5); cache = g.tensor(ea.F16, 4096 * 64)
        kview = g.view(cache, [5 * 4096], [2], 4096 * 2 * 7)
        return [a], g.cpy(a, kview)
    got, want = _both(ea, gpu, ref_cpu, lambda g: b5(g), [kcur])
    def b6(g):
        a = g.tensor(ea.F32, 4096, 5); cache = g.tensor(ea.F16, 4096 * 64)
        vt = g.transpose(a)
        vview = g.view(cache, [5, 4096], [2, 64 * 2], 7 * 2)
        g.cpy(vt, vview)
        return [a], g.cont(g.view(cache, [64 * 4096], [2], 0))
    # cache content is uninitialised outside the written cells: compare the written cells only
    outs = []
    for be in (gpu, ref_cpu):
        g = ea.Graph(be); tens, out = b6(g); g.alloc(); g.set(tens[0], kcur); g.compute()
        outs.append(g.get(out, np.float16).reshape(4096, 64)[:, 7:12].copy())
    assert np.array_equal(outs[0].view(np.uint16), outs[1].view(np.uint16))
    assert np.array_equal(outs[0], kcur.T.astype(np.float16))
    # GET_ROWS (inp_out_ids row select)
    tab = rng.standard_normal((9, 4096)).astype(np.float32); idx = np.array([8, 0, 3], np.int32)
    def b7(g):
        a = g.tensor(ea.F32, 4096, 9); i = g.tensor(ea.I32, 3); return [a, i], g.get_rows(a, i)
    got, want = _both(ea, gpu, ref_cpu, b7, [tab, idx]); assert np.array_equal(got, want)


def test_eagle_fc_front_fused_vs_reference_backend(ea, gpu, ref_cpu):
    """CONCAT([embd; hidd]) -> MUL_MAT(fc) -> ADD(bias row) -> RELU (R/src/llama.cpp:1863-1869) runs as ONE launch on the GPU
    (two-source quantiser, broadcast bias, relu epilogue); same graph on the reference CPU backend.  T = 1 takes the dp4a kernel
    with its in-block quantiser, T = 6 the quantise-once image + matrix-core kernel."""
    rng = np.random.default_rng(21)
    for t, T in [(12, 1), (12, 6), (14, 3), (8, 2), (12, 9), (12, 25), (14, 61), (8, 25), (2, 9), (2, 61)]:   # T > 8 / > 24: several token passes -- both CONCAT sources must advance
        E = 1024
        w = qdata.random_blocks(t, E, 2 * E, rng)
        e = rng.standard_normal((T, E)).astype(np.float32); hdn = rng.standard_normal((T, E)).astype(np.float32)
        bias = rng.standard_normal(E).astype(np.float32) * 20
        def build(g):
            a = g.tensor(ea.F32, E, T); b = g.tensor(ea.F32, E, T); wt = g.tensor(t, 2 * E, E); bb = g.tensor(ea.F32, E)
            return [a, b, wt, bb], g.unary(g.add(g.mul_mat(wt, g.concat(a, b, 0)), bb), "relu")
        got, want = _both(ea, gpu, ref_cpu, build, [e, hdn, w, bias])
        assert rel(got, want) < 2e-5, (t, T)
        assert (got >= 0).all() and (got == 0).any() and (got > 0).any()


def test_attention_subgraph_tree_mask(ea, gpu):
    """K.q -> soft_max(tree mask) -> V.p -> permute -> cont, on the fp16 KV layout of the reference"""
    z = np.load(os.path.join(GOLD, "small_ops.npz"))
    q, kc, vc, mask = z["at_q"], z["at_k"], z["at_v"], z["at_mask"]
    T, H, d = q.shape; Hkv, n_kv = kc.shape[0], kc.shape[1]
    g = ea.Graph(gpu)
    tq = g.tensor(ea.F32, d, H, T); tk = g.tensor(ea.F16, d, n_kv, Hkv); tv = g.tensor(ea.F16, n_kv, d, Hkv); tm = g.tensor(ea.F32, n_kv, 64)
    kqt = g.mul_mat(tk, g.permute(tq, 0, 2, 1, 3))
    sm = g.soft_max(kqt, tm, float(z["at_scale"]))
    res = g.cont(g.permute(g.mul_mat(tv, sm), 0, 2, 1, 3))
    g.alloc(); g.set(tq, q); g.set(tk, kc); g.set(tv, vc); g.set(tm, mask); g.compute()
    got = g.get(res).reshape(T, H, d)
    assert rel(got, z["at_y"]) < 1e-3            # the fused single-kernel path (intermediates are never materialised)
    assert rel(got, orc.attention(q, kc, vc, mask, float(z["at_scale"]), Hkv)) < 1e-3
    # same graph with the intermediates flagged as graph outputs: runs node by node, so they can be inspected
    g = ea.Graph(gpu)
    tq = g.tensor(ea.F32, d, H, T); tk = g.tensor(ea.F16, d, n_kv, Hkv); tv = g.tensor(ea.F16, n_kv, d, Hkv); tm = g.tensor(ea.F32, n_kv, 64)
    kqt = g.mul_mat(tk, g.permute(tq, 0, 2, 1, 3))
    sm = g.soft_max(kqt, tm, float(z["at_scale"]))
    res = g.cont(g.permute(g.mul_mat(tv, sm), 0, 2, 1, 3))
    ea.host().eh_tensor_set_flags(kqt, 2); ea.host().eh_tensor_set_flags(sm, 2)
    g.alloc(); g.set(tq, q); g.set(tk, kc); g.set(tv, vc); g.set(tm, mask); g.compute()
    assert rel(g.get(res).reshape(T, H, d), z["at_y"]) < 1e-3
    p = g.get(sm).reshape(H, T, n_kv)
    assert rel(p, orc.soft_max(g.get(kqt).reshape(H, T, n_kv), mask, float(z["at_scale"]))) < 2e-6
    assert np.all(p[:, np.isinf(mask[:T])] == 0)   # masked cells get exactly zero probability


@pytest.mark.parametrize("n_kv,T", [(4096, 6), (4096, 1), (2048, 13), (6144, 4)])
def test_attention_long_context(ea, gpu, n_kv, T):
    """The fused attention kernel keeps a token's scores and probabilities in ONE LDS row (probabilities in place): n_kv = 4096 at 8 tokens per block
    (Llama-2's context) stays on it; 6144 cells fall back to the node-by-node path.  Both against the restatement."""
    rng = np.random.default_rng(n_kv + T)
    H, Hkv, d = 8, 4, 128
    q = rng.standard_normal((T, H, d)).astype(np.float32)
    kc = rng.standard_normal((Hkv, n_kv, d)).astype(np.float16)
    vc = rng.standard_normal((Hkv, d, n_kv)).astype(np.float16)
    mask = np.zeros((64, n_kv), np.float32)
    for t in range(T):
        mask[t, n_kv - 3 * (T - t):] = -np.inf           # a causal tail
        mask[t, 100 + 7 * t: 140 + 7 * t] = -np.inf      # and a hole, different per token (tree-style)
    scale = 1.0 / np.sqrt(d)
    g = ea.Graph(gpu)
    tq = g.tensor(ea.F32, d, H, T); tk = g.tensor(ea.F16, d, n_kv, Hkv); tv = g.tensor(ea.F16, n_kv, d, Hkv); tm = g.tensor(ea.F32, n_kv, 64)
    kqt = g.mul_mat(tk, g.permute(tq, 0, 2, 1, 3))
    sm = g.soft_max(kqt, tm, float(scale))
    res = g.cont(g.permute(g.mul_mat(tv, sm), 0, 2, 1, 3))
    g.alloc(); g.set(tq, q); g.set(tk, kc); g.set(tv, vc); g.set(tm, mask); g.compute()
    got = g.get(res).reshape(T, H, d)
    assert rel(got, orc.attention(q, kc, vc, mask, float(scale), Hkv)) < 1e-3


@pytest.mark.gpu
def test_node_hooks_fire_in_graph_order_on_the_stream(ea, gpu):
    """ggml_backend_mi355x_set_node_hooks: the callback runs inside graph_compute right after the hooked node has been queued, and what it
    enqueues on the stream it is handed is ordered before every later node.  Here the hook zeroes the hooked tensor (hipMemsetAsync on
    that stream): the nodes behind it must see zeros, the nodes before it must not."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]; hip.hipMemsetAsync.restype = C.c_int
    rng = np.random.default_rng(5)
    g = ea.Graph(gpu)
    a = g.tensor(ea.F32, 256, 4); b = g.tensor(ea.F32, 256, 4)
    s1 = g.add(a, b)                      # hooked: zeroed by the callback after it was computed
    s2 = g.add(s1, b)                     # must be 0 + b
    s3 = g.add(s2, a)                     # hooked as well (left alone): order of the calls
    s4 = g.add(s3, b)
    g.alloc()
    xa = rng.standard_normal((4, 256)).astype(np.float32); xb = rng.standard_normal((4, 256)).astype(np.float32)
    g.set(a, xa); g.set(b, xb)
    calls = []
    HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)

    def cb(user, t, stream):
        calls.append(t)
        if t == s1:
            assert hip.hipMemsetAsync(ea.tensor_data(s1), 0, 4 * 256 * 4, stream) == 0
    hook = HOOK(cb)
    ea.set_node_hooks(gpu, [s1, s3], hook)
    g.compute()
    ea.set_node_hooks(gpu, [], None)
    assert len(calls) == 2 and calls[0] != calls[1]
    assert np.array_equal(g.get(s1).reshape(4, 256), np.zeros((4, 256), np.float32))
    np.testing.assert_allclose(g.get(s2).reshape(4, 256), xb, rtol=0, atol=0)
    np.testing.assert_allclose(g.get(s4).reshape(4, 256), (xb + xa) + xb, rtol=1e-6)
    g.compute()                            # hooks removed: the plain result again
    np.testing.assert_allclose(g.get(s2).reshape(4, 256), (xa + xb) + xb, rtol=1e-6)
