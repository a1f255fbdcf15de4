"""Pins the CPU restatement (oracle/*.c) to the reference itself.

* against oracle/_ref/libggml-ref-scalar.so (the reference compiled from /root/reference without SIMD
  flags, i.e. its ISA-independent branches): BIT-EXACT, on fresh seeded inputs;
* against tests/golden/*.npz (vectors the same reference build produced, committed): BIT-EXACT, so the
  pin also holds where oracle/_ref is absent.
CPU only.
"""
import os
import numpy as np
import pytest

import oracle as orc
from conftest import have_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")
QTYPES = {"q4_0": 2, "q8_0": 8, "q4_K": 12, "q5_K": 13, "q6_K": 14}
DOT = {"q4_0": "q4_0_q8_0", "q8_0": "q8_0_q8_0", "q4_K": "q4_K_q8_K", "q5_K": "q5_K_q8_K", "q6_K": "q6_K_q8_K"}
needs_ref = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@needs_ref
def test_fp16_conversions_match_reference():
    import refapi
    L, O = refapi.lib(True), orc.load()
    hs = np.arange(0, 65536, dtype=np.uint32)
    for h in hs[::7]:
        a, b = L.ggml_fp16_to_fp32(int(h)), O.orc_fp16_to_fp32(int(h))
        assert (np.isnan(a) and np.isnan(b)) or a == b
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.standard_normal(20000).astype(np.float32) * 10.0 ** rng.integers(-9, 6, 20000),
                         np.array([0, -0.0, 65504, 65519.99, 65520, 1e9, 5.96e-8, 2.98e-8, 2.9802322e-8, 2.99e-8, 6.1e-5, 6.0e-5], np.float32)])
    # halfway cases between adjacent halves
    hv = np.array([L.ggml_fp16_to_fp32(int(h)) for h in range(0x0001, 0x7bff, 97)], np.float32)
    xs = np.concatenate([xs, (hv[:-1].astype(np.float64) + hv[1:].astype(np.float64)).astype(np.float32) / 2])
    for x in xs:
        assert L.ggml_fp32_to_fp16(float(x)) == O.orc_fp32_to_fp16(float(x)), x


@needs_ref
@pytest.mark.parametrize("name", ["q8_0", "q8_1", "q8_K"])
def test_activation_quantisers_bit_exact(name):
    import refapi
    rng = np.random.default_rng(2)
    x = (rng.standard_normal(4096) * rng.choice([1e-3, 1, 30], 4096)).astype(np.float32)
    x[256:512] = 0                      # an all-zero super-block
    x[600] = -x[601]                    # equal magnitudes, opposite signs: first one wins in q8_K
    ours = getattr(orc, f"quantize_{name}")(x)
    ref = refapi.quantize_act(name, x)
    if name == "q8_K":
        # the reference leaves bsums of an all-zero super-block unwritten (R/ggml/src/ggml-quants.c:2493-2498);
        # they are multiplied by d == 0 later, so only the written bytes are compared there
        ours = ours.reshape(-1, 292).copy(); ref = ref.reshape(-1, 292).copy()
        zero = ours[:, :4].view(np.float32)[:, 0] == 0
        assert zero.sum() == 1
        ours[zero, 260:] = 0; ref[zero, 260:] = 0
    assert np.array_equal(ours, ref)


@needs_ref
@pytest.mark.parametrize("tname", list(QTYPES))
def test_dequant_and_vec_dot_bit_exact(tname):
    import refapi
    t = QTYPES[tname]
    rng = np.random.default_rng(3)
    k, rows = 1024, 8
    w = refapi.quantize(t, rng.standard_normal((rows, k)).astype(np.float32), k)
    rb = orc.row_bytes(t, k)
    x = rng.standard_normal(k).astype(np.float32)
    act = "q8_K" if tname.endswith("_K") else "q8_0"
    y = refapi.quantize_act(act, x)
    for r in range(rows):
        blk = w[r*rb:(r+1)*rb]
        assert np.array_equal(bits(orc.dequantize(t, blk, k)), bits(refapi.dequantize(t, blk, k)))
        ours = getattr(orc.load(), f"orc_vec_dot_{DOT[tname]}")(k, orc._p(blk), orc._p(y))
        assert bits(np.float32(ours)) == bits(refapi.vec_dot(DOT[tname], k, blk, y))
    # random bit patterns are valid blocks too (scale fields at their extremes)
    raw = rng.integers(0, 256, rb, dtype=np.uint8)
    if tname in ("q4_K", "q5_K"):
        raw.view(np.uint16)[0:2] = np.array([0x2c00, 0x2800], np.uint16)
    assert np.array_equal(bits(orc.dequantize(t, raw, k)), bits(refapi.dequantize(t, raw, k))) or tname in ("q4_0", "q8_0", "q6_K")


@needs_ref
@pytest.mark.parametrize("tname", list(QTYPES))
def test_mul_mat_q_matches_reference_graph(tname, ea, ref_scalar):
    """whole MUL_MAT through the reference CPU backend (scalar build) == restatement, bit for bit"""
    import refapi
    t = QTYPES[tname]
    rng = np.random.default_rng(4)
    k, rows, T = 512, 24, 3
    w = refapi.quantize(t, rng.standard_normal((rows, k)).astype(np.float32), k)
    x = rng.standard_normal((T, k)).astype(np.float32)
    g = ea.Graph(ref_scalar)
    a, b = g.tensor(t, k, rows), g.tensor(ea.F32, k, T)
    c = g.mul_mat(a, b)
    g.alloc(); g.set(a, w); g.set(b, x); g.compute()
    ref = g.get(c).reshape(T, rows)
    assert np.array_equal(bits(orc.mul_mat_q(t, w, x, k, rows)), bits(ref))


@needs_ref
def test_small_ops_match_reference_graph(ea, ref_scalar):
    rng = np.random.default_rng(5)
    # RMS_NORM
    x = rng.standard_normal((5, 320)).astype(np.float32)
    g = ea.Graph(ref_scalar); a = g.tensor(ea.F32, 320, 5); r = g.rms_norm(a, 1e-6); g.alloc(); g.set(a, x); g.compute()
    assert np.array_equal(bits(orc.rms_norm(x, 1e-6)), bits(g.get(r).reshape(5, 320)))
    # ROPE (NORM and NEOX), positions with duplicates as a token tree produces them
    xq = rng.standard_normal((6, 4, 128)).astype(np.float32); pos = np.array([7, 8, 8, 9, 9, 300], np.int32)
    for mode in (0, 2):
        g = ea.Graph(ref_scalar); a = g.tensor(ea.F32, 128, 4, 6); p = g.tensor(ea.I32, 6); r = g.rope(a, p, 128, mode)
        g.alloc(); g.set(a, xq); g.set(p, pos); g.compute()
        assert np.array_equal(bits(orc.rope(xq, pos, 128, mode)), bits(g.get(r).reshape(6, 4, 128)))
    # SOFT_MAX with a -inf tree mask
    kq = rng.standard_normal((4, 6, 96)).astype(np.float32); mask = np.zeros((64, 96), np.float32)
    mask[rng.random((64, 96)) < 0.4] = -np.inf; mask[:, 0] = 0
    g = ea.Graph(ref_scalar); a = g.tensor(ea.F32, 96, 6, 4); m = g.tensor(ea.F32, 96, 64); r = g.soft_max(a, m, 0.088)
    g.alloc(); g.set(a, kq); g.set(m, mask); g.compute()
    # exp(): every x86-64 build of the reference (even without SIMD flags: __SSE2__ is always set) takes the
    # vectorised polynomial ggml_v_expf (ggml-cpu.c:2144-2159); the restatement follows the ISA-independent
    # libm tail (:2168-2172).  Same algorithm otherwise; results agree to a few ulp, zeros exactly.
    o, rr = orc.soft_max(kq, mask, 0.088), g.get(r).reshape(4, 6, 96)
    assert np.array_equal(o == 0, rr == 0) and np.abs(o - rr).max() <= 4e-7 * np.abs(rr).max()
    # f16 mat-mul (attention flavour): src1 rounded to f16, double accumulation
    A = rng.standard_normal((40, 128)).astype(np.float16); B = rng.standard_normal((3, 128)).astype(np.float32)
    g = ea.Graph(ref_scalar); a = g.tensor(ea.F16, 128, 40); b = g.tensor(ea.F32, 128, 3); r = g.mul_mat(a, b)
    g.alloc(); g.set(a, A); g.set(b, B); g.compute()
    assert np.array_equal(bits(orc.mul_mat_f16(A, B)), bits(g.get(r).reshape(3, 40)))
    # SILU
    v = rng.standard_normal(1000).astype(np.float32) * 4
    g = ea.Graph(ref_scalar); a = g.tensor(ea.F32, 1000); r = g.unary(a, "silu"); g.alloc(); g.set(a, v); g.compute()
    assert np.abs(orc.silu(v) - g.get(r)).max() <= 4e-7 * np.abs(v).max()      # same expf note as above


@pytest.mark.parametrize("tname", list(QTYPES))
def test_golden_vectors(tname):
    """committed vectors produced by the reference build (tests/golden/make_golden.py)"""
    z = np.load(os.path.join(GOLD, f"mul_mat_{tname}.npz"))
    t = QTYPES[tname]
    k, rows = int(z["k"]), int(z["rows"])
    for T in (1, 2, 8):
        x = z["x"][:T]
        got = orc.mul_mat_q(t, z["w"], x, k, rows)
        assert np.array_equal(bits(got), bits(z["dst_scalar"][:T])), f"{tname} T={T}"
        # the SIMD build of the reference differs from its own scalar build only by fp32 summation order
        err = np.abs(got - z["dst_avx2"][:T]).max() / np.abs(z["dst_avx2"][:T]).max()
        assert err < 2e-6
    act = "q8_K" if tname.endswith("_K") else "q8_0"
    assert np.array_equal(getattr(orc, f"quantize_{act}")(z["x"][0]), z["x0_quant"])


def test_golden_small_ops():
    z = np.load(os.path.join(GOLD, "small_ops.npz"))
    assert np.array_equal(bits(orc.rms_norm(z["rms_x"], 1e-6)), bits(z["rms_y"]))
    assert np.array_equal(bits(orc.rope(z["rope_x"], z["rope_pos"], 128, 0)), bits(z["rope_y"]))
    # exp() based: few-ulp tolerance, see test_small_ops_match_reference_graph
    o = orc.soft_max(z["sm_x"], z["sm_mask"], float(z["sm_scale"]))
    assert np.array_equal(o == 0, z["sm_y"] == 0) and np.abs(o - z["sm_y"]).max() <= 4e-7 * z["sm_y"].max()
    o = orc.attention(z["at_q"], z["at_k"], z["at_v"], z["at_mask"], float(z["at_scale"]), int(z["at_hkv"]))
    assert np.abs(o - z["at_y"]).max() <= 2e-3 * np.abs(z["at_y"]).max()      # p is rounded to f16: a 1-ulp exp difference can flip a half
