"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref, built from /root/reference by
oracle/Makefile).  Run in the build container only:   python tests/golden/make_golden.py

Recorded build flags of the reference libraries (oracle/Makefile):
    libggml-ref-scalar.so : gcc 11.4 -O3 -ffp-contract=off, no SIMD flags  -> ISA-independent branches ("dst_scalar")
    libggml-ref.so        : gcc 11.4 -O3 -mavx2 -mfma -mf16c -mbmi2         -> default x86 build        ("dst_avx2")
Every array is data (inputs + the reference's outputs); no reference source text is stored.
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_package  # noqa: E402
import refapi  # noqa: E402

ea = load_package()
QTYPES = {"q4_0": 2, "q8_0": 8, "q4_K": 12, "q5_K": 13, "q6_K": 14}


def run_mul_mat(be, t, w, x, k, rows):
    T = x.shape[0]
    g = ea.Graph(be)
    a, b = g.tensor(t, k, rows), g.tensor(ea.F32, k, T)
    c = g.mul_mat(a, b)
    g.alloc(); g.set(a, w); g.set(b, x); g.compute()
    return g.get(c).reshape(T, rows).copy()


def main():
    scalar, avx2 = refapi.reference_cpu(ea, scalar=True, threads=4), refapi.reference_cpu(ea, threads=4)
    rng = np.random.default_rng(20251004)
    k, rows, T = 512, 64, 8
    for name, t in QTYPES.items():
        wf = (rng.standard_normal((rows, k)) * 0.05).astype(np.float32)
        w = refapi.quantize(t, wf, k)
        x = rng.standard_normal((T, k)).astype(np.float32)
        x[1, 256:512] = 0.0                                  # an all-zero activation super-block
        act = "q8_K" if name.endswith("_K") else "q8_0"
        np.savez_compressed(os.path.join(HERE, f"mul_mat_{name}.npz"), k=k, rows=rows, w=w, x=x,
                            dst_scalar=run_mul_mat(scalar, t, w, x, k, rows), dst_avx2=run_mul_mat(avx2, t, w, x, k, rows),
                            x0_quant=refapi.quantize_act(act, x[0]))
    # small ops + the attention sub-graph, on a tree-shaped mask
    out = {}
    x = rng.standard_normal((5, 320)).astype(np.float32)
    g = ea.Graph(scalar); a = g.tensor(ea.F32, 320, 5); r = g.rms_norm(a, 1e-6); g.alloc(); g.set(a, x); g.compute()
    out["rms_x"], out["rms_y"] = x, g.get(r).reshape(5, 320).copy()
    xq = rng.standard_normal((6, 4, 128)).astype(np.float32); pos = np.array([7, 8, 8, 9, 9, 300], np.int32)
    g = ea.Graph(scalar); a = g.tensor(ea.F32, 128, 4, 6); p = g.tensor(ea.I32, 6); r = g.rope(a, p, 128, 0)
    g.alloc(); g.set(a, xq); g.set(p, pos); g.compute()
    out["rope_x"], out["rope_pos"], out["rope_y"] = xq, pos, g.get(r).reshape(6, 4, 128).copy()
    # soft_max / attention: reference outputs come from the scalar build's libm-free tail?  No: x86 builds use the
    # SSE2 polynomial exp, so these two are stored from the restatement-independent reference and compared with a
    # few-ulp tolerance in test_oracle.py::test_golden_small_ops.
    n_kv, T, H, Hkv, d = 96, 6, 4, 2, 64
    mask = np.full((64, n_kv), -np.inf, np.float32)
    # tree: 40 prompt cells visible to all; 3 branches of 2 tokens each see their own ancestors only
    mask[:T, :40] = 0
    for j in range(T):
        br, depth = j % 3, j // 3
        for dd in range(depth + 1):
            mask[j, 40 + br + 3 * dd] = 0
    kq = rng.standard_normal((H, T, n_kv)).astype(np.float32)
    scale = 1.0 / np.sqrt(d)
    g = ea.Graph(scalar); a = g.tensor(ea.F32, n_kv, T, H); m = g.tensor(ea.F32, n_kv, 64); r = g.soft_max(a, m, scale)
    g.alloc(); g.set(a, kq); g.set(m, mask); g.compute()
    out["sm_x"], out["sm_mask"], out["sm_scale"], out["sm_y"] = kq, mask, np.float32(scale), g.get(r).reshape(H, T, n_kv).copy()
    # attention exactly as llm_build_kqv wires it (no flash attention)
    q = rng.standard_normal((T, H, d)).astype(np.float32)
    kc = rng.standard_normal((Hkv, n_kv, d)).astype(np.float16)
    vc = rng.standard_normal((Hkv, d, n_kv)).astype(np.float16)
    g = ea.Graph(scalar)
    tq = g.tensor(ea.F32, d, H, T); tk = g.tensor(ea.F16, d, n_kv, Hkv); tv = g.tensor(ea.F16, n_kv, d, Hkv); tm = g.tensor(ea.F32, n_kv, 64)
    qp = g.permute(tq, 0, 2, 1, 3)
    kqt = g.mul_mat(tk, qp)
    sm = g.soft_max(kqt, tm, scale)
    kqv = g.mul_mat(tv, sm)
    mer = g.permute(kqv, 0, 2, 1, 3)
    res = g.cont(mer)
    g.alloc(); g.set(tq, q); g.set(tk, kc); g.set(tv, vc); g.set(tm, mask); g.compute()
    out.update(at_q=q, at_k=kc.view(np.uint16), at_v=vc.view(np.uint16), at_mask=mask, at_scale=np.float32(scale), at_hkv=Hkv,
               at_y=g.get(res).reshape(T, H, d).copy())
    np.savez_compressed(os.path.join(HERE, "small_ops.npz"), **out)
    model_forward_fixture(ea, scalar)
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


def model_forward_fixture(ea, scalar):
    """one llama forward and one EAGLE-head forward on the tiny synthetic GGUF-shaped pair (n_embd 256, 4 heads, n_ff 768,
    n_vocab 512, 2 layers, Q4_K_M mix; weights regenerated from the seed by host/model.cpp), computed by the REAL reference CPU backend
    (scalar build, ISA independent) through the build_llama / build_eagle mirrors: prompt, single token, a 3-branch tree batch with shared
    positions, and the draft head fed with the target's features.  tests/test_golden_model.py replays them."""
    cfg, ftype, seed = "tiny", "q4_k_m", 3
    m = ea.Model(scalar, cfg, ftype, n_ctx=256, seed=seed, predictable=False)
    d = ea.Model(scalar, cfg, ftype, n_ctx=256, eagle_of=m, seed=seed, predictable=False)
    out = dict(config=cfg, ftype=ftype, seed=np.int32(seed))
    p_tok, p_pos = list(range(5, 21)), list(range(16))
    lg, hid = m.decode(p_tok, p_pos); out.update(p_tok=np.int32(p_tok), p_pos=np.int32(p_pos), p_logits=lg, p_hidden=hid)
    lg1, hid1 = m.decode([77], [16]); out.update(s_tok=np.int32([77]), s_pos=np.int32([16]), s_logits=lg1, s_hidden=hid1)
    h = ea._model_sigs()
    m.kv_seq_rm(0, 17, -1)
    for sq in (1, 2, 3):
        h.eh_model_kv_seq_cp(m.h, 0, sq, -1, -1)
    t_tok, t_pos, t_seq = [10, 11, 12, 13, 14, 15], [17, 18, 17, 18, 17, 18], [1, 1, 2, 2, 3, 3]
    lgt, hidt = m.decode(t_tok, t_pos, seq=t_seq); out.update(t_tok=np.int32(t_tok), t_pos=np.int32(t_pos), t_seq=np.int32(t_seq), t_logits=lgt, t_hidden=hidt)
    e_tok, e_pos = [30, 31, 32], [1, 2, 3]
    lgd, hidd = d.decode(e_tok, e_pos, hidd=hid[:3]); out.update(e_tok=np.int32(e_tok), e_pos=np.int32(e_pos), e_logits=lgd, e_hidden=hidd)
    lgd2, hidd2 = d.decode([33], [4], hidd=hidd[2:3]); out.update(e2_tok=np.int32([33]), e2_pos=np.int32([4]), e2_logits=lgd2, e2_hidden=hidd2)
    d.close(); m.close()
    out = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64 else v) for k, v in out.items()}
    np.savez_compressed(os.path.join(HERE, "model_forward_tiny.npz"), **out)


if __name__ == "__main__":
    main()
