"""north_star's 1e-3 where it can hold: teacher-forced parity at REAL widths (VERDICT r2 "next" #4b).

End to end, two IEEE-correct evaluations of a quantised transformer drift apart because the int8 activation quantiser turns a 1e-5
difference (f16 attention arithmetic, fp32 summation order) into a flipped int8 value now and then: ONE flipped value among the 5120
inputs of a token moves that mat-mul's output by ~3e-4, and every later mat-mul of the layer multiplies what it is handed
(profiles/r03_flip_replay_*.txt: wo 1 flip -> 1.4e-4, gate/up 16 flips -> 5e-4, down 133 flips -> 2.7e-3, LM head 915 flips -> 4.2e-3,
while the reference's own AVX2 and scalar builds, flip-free by luck in that decode, stay at 1e-6 -- and sit 1e-2 apart in others).
Teacher forcing takes the compounding away: the plugin is fed the REFERENCE's own intermediate values and only its own step is compared.

  * node level (asserted, 1e-3; lands near 1e-6) -- every quantised MUL_MAT of every decode of the width cases (tests/test_width_gpu.py
    CASES: Vicuna-7B Q4_K_M / Q8_0 layers, the Llama-2-13B TP=2 and Llama-2-70B TP=8 rank-0 shards) gets the reference's input
    activations for that node and must reproduce the reference's output: same int8 values, same integer sums;
  * the attention block (asserted, 1e-3; lands near 4e-5) -- in layer 0, and in every layer above fed the reference's layer input, the
    input of wo (norm -> q|k|v -> RoPE -> KV store -> K.q -> soft-max -> P.V) is compared before any flip can have happened upstream
    (the block's inputs are bit-identical or one ulp apart);
  * layer level (asserted: NEVER above 1e-3 without a counted flip) -- every layer reads the reference's output of the layer below
    (Model.force_layer_inputs); its output l_out-i is compared with the reference's, and the int8 values the reference's own quantiser
    (oracle/, the checker as a measuring tool) derives from the plugin's and from the reference's mat-mul inputs are counted.  An error
    above 1e-3 with zero flips in the layer would be a real defect; with flips it is the quantiser's step, not the backend's arithmetic.
The end-to-end figures stay in tests/test_width_gpu.py as recorded numbers."""
import numpy as np
import pytest

import refapi
from conftest import have_ref
from test_width_gpu import CASES

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")]

OP_MUL_MAT = 26
QUANT_TYPES = (2, 8, 12, 13, 14)
NORTH_STAR = 1e-3

DECODES = [
    (list(range(5, 21)), list(range(16)), None),                                   # prompt of 16
    ([77], [16], None),                                                            # T = 1
    ([90, 91, 92, 93, 94, 95], [17, 18, 19, 20, 21, 22], None),                    # T = 6 chain verification
]


def l2rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / (np.linalg.norm(b.astype(np.float64)) + 1e-30))


def _model(ea, be, dims, ftype, tp):
    kw = dict(tp_rank=tp[0], tp_size=tp[1]) if tp else {}
    m = ea.Model(be, tuple(dims), ftype, n_ctx=256, seed=7, predictable=False, **kw)
    if tp:
        m.set_allreduce(lambda ptr, n: None)                 # identity all-reduce: rank 0's partial sums, the same on both backends
    return m


def _matmuls(ea, m, values=True):
    """[(node index, weight handle, weight info, input rows [T, k], output rows [T, rows])] of the last decode's quantised MUL_MATs
    (values=False: handles only -- on the plugin a fused intermediate need not have been written)"""
    out = []
    for i, t in enumerate(m.nodes()):
        info = ea.tensor_info(t)
        if info["op"] != OP_MUL_MAT:
            continue
        w = ea.tensor_src(t, 0); x = ea.tensor_src(t, 1)
        wi = ea.tensor_info(w); xi = ea.tensor_info(x)
        if wi["type"] not in QUANT_TYPES or xi["type"] != 0 or xi["nb"][0] != 4 or xi["nb"][1] != 4 * xi["ne"][0] or xi["ne"][2] != 1:
            continue
        xin = m.read_tensor(x).reshape(-1, xi["ne"][0])[:xi["ne"][1]].copy() if values else None
        y = m.read_tensor(t).reshape(-1, info["ne"][0])[:info["ne"][1]].copy() if values else None
        out.append((i, w, wi, xin, y))
    return out


def _q_int8(orc, x, wtype):
    """the int8 activation values the reference derives from a row (Q8_K for K-quants, Q8_0 for q8_0 / q4_0)"""
    if wtype in (12, 13, 14):
        b = orc.quantize_q8_K(x).reshape(-1, 292); return b[:, 4:260].view(np.int8).reshape(-1)
    b = orc.quantize_q8_0(x).reshape(-1, 34); return b[:, 2:].view(np.int8).reshape(-1)


def _layer_of(wname):
    return int(wname.split(".")[1]) if wname.startswith("blk.") else -1             # -1: the LM head


@pytest.mark.parametrize("case", list(CASES))
def test_teacher_forced_nodes_and_layers_meet_1e_3(ea, gpu, case, capsys):
    import oracle as orc
    dims, ftype, tp = CASES[case]
    n_layer = dims[5]
    # Q8_0 activations: the reference's AVX2 quantiser is not its scalar one (id = 127 / max and round-half-even, against id = 1 / d and
    # roundf: R/ggml/src/ggml-cpu/ggml-cpu-quants.c quantize_row_q8_0) -- its two builds flip against each other at node level by design.
    # The plugin restates the scalar, ISA-independent rule, so the Q8_0 case is teacher-forced by the scalar build (vs AVX2 the same
    # nodes land at 3e-4: inside the bound, but that is the reference's own split, not a property of the backend)
    ref = _model(ea, refapi.reference_cpu(ea, scalar=(ftype in ("q8_0", "q4_0"))), dims, ftype, tp)
    dev = _model(ea, gpu, dims, ftype, tp)
    worst_node, worst_attn, n_nodes, report = 0.0, 0.0, 0, []
    for tokens, pos, seq in DECODES:
        ref.decode(tokens, pos, want_hidden=True)
        r_mm = _matmuls(ea, ref)
        r_lout = ref.named_nodes("l_out-")
        # ---- layer level: layer il >= 1 reads the reference's l_out of layer il - 1 (layer 0 reads the token embeddings, which are exact)
        dev.force_layer_inputs([None] + [r_lout[f"l_out-{il - 1}"] for il in range(1, n_layer)])
        dev.decode(tokens, pos, want_hidden=True)
        d_lout = dev.named_nodes("l_out-")
        d_mm = _matmuls(ea, dev)           # (every mat-mul input is a tensor the fused launches write too: folded norms, the SwiGLU product, the attention output)
        assert len(d_mm) == len(r_mm) and len(d_lout) == n_layer
        flips = {}
        for (i, _, wi, x_ref, _), (_, _, _, x_dev, _) in zip(r_mm, d_mm):
            il = _layer_of(wi["name"])
            n = sum(int((_q_int8(orc, a, wi["type"]) != _q_int8(orc, b, wi["type"])).sum()) for a, b in zip(x_dev, x_ref))
            flips[il] = flips.get(il, 0) + n
            if ".attn_output." in wi["name"]:                      # the attention block's result, before any flip can have happened in this layer
                e = l2rel(x_dev, x_ref); worst_attn = max(worst_attn, e)
                assert e <= NORTH_STAR, (case, len(tokens), wi["name"], e)
        for il in range(n_layer):
            e = l2rel(d_lout[f"l_out-{il}"], r_lout[f"l_out-{il}"])
            report.append((len(tokens), il, e, flips.get(il, 0)))
            assert e <= NORTH_STAR or flips.get(il, 0) > 0, (case, len(tokens), il, e, "above 1e-3 without a single flipped int8 activation")
        # ---- node level: the plugin's MUL_MAT over ITS copy of the weight (same synthetic bytes, possibly re-tiled), the reference's input
        for (i, w_ref, wi, xin, y_ref), (j, w_dev, wj, _, _) in zip(r_mm, d_mm):
            assert i == j and wi["ne"] == wj["ne"] and wi["type"] == wj["type"]
            T, k = xin.shape
            g = ea.Graph(gpu)
            x = g.tensor(ea.F32, k, T)
            y = g.mul_mat(w_dev, x)
            g.alloc(); g.set(x, xin); g.compute()
            e = l2rel(g.get(y).reshape(T, -1), y_ref)
            worst_node = max(worst_node, e); n_nodes += 1
            assert e <= NORTH_STAR, (case, len(tokens), i, wi["name"], e)
    with capsys.disabled():
        print(f"[teacher-forced] {case}: {n_nodes} quantised MUL_MATs fed the reference's inputs, worst rel-L2 {worst_node:.2e}; attention blocks worst {worst_attn:.2e}  (asserted <= {NORTH_STAR:.0e})")
        for T, il, e, nf in report:
            print(f"[teacher-forced] {case} T={T} layer {il}: l_out rel-L2 {e:.2e}, int8 activation values flipped inside the layer {nf}")
    ref.close(); dev.close()
