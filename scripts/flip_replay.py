"""Where do the 1e-3 .. 1e-2 whole-model differences come from?  Replay of the case that failed round 2's first bound
(gpurun_out/r2_t15.log: llama-2-13b TP=2 rank 0, the 6-token verification batch), node by node, on three backends running the SAME host graph:
the plugin (every node written: GGML_MI355X_NO_FUSION=1), the reference CPU backend built with AVX2, and the reference built scalar.

    python scripts/flip_replay.py [case] > profiles/r03_flip_replay_<case>.txt          (cases: tests/test_width_gpu.py CASES)

For every f32 node: rel-L2 plugin-vs-AVX2, plugin-vs-scalar, AVX2-vs-scalar.  For every quantised MUL_MAT: how far apart the three
INPUTS are, how many of the int8 activation values the reference's own quantiser (oracle/, quantize_row_q8_K / q8_0 -- the checker,
used here as a measuring tool) derives from those inputs differ, and how far apart the OUTPUTS are.  An input difference of 1e-7
becoming an output difference of 1e-3 with a handful of differing int8 values in between is the "flip": a value sitting on a rounding
boundary of round(x / d) lands on different sides for two summation orders.  Teacher forcing (tests/test_teacher_forced_gpu.py) is the
other half: the same MUL_MAT fed the reference's own input agrees with the reference to the op tolerance."""
import os, sys
os.environ["GGML_MI355X_NO_FUSION"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from conftest import load_package
import refapi
import oracle as orc
from test_width_gpu import CASES

OP_MUL_MAT = 26
QUANT = {2: "q4_0", 8: "q8_0", 12: "q4_K", 13: "q5_K", 14: "q6_K"}


def l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / (np.linalg.norm(b.astype(np.float64)) + 1e-30))


def run(ea, be, dims, ftype, tp):
    kw = dict(tp_rank=tp[0], tp_size=tp[1]) if tp else {}
    m = ea.Model(be, tuple(dims), ftype, n_ctx=256, seed=7, predictable=False, **kw)
    if tp:
        m.set_allreduce(lambda ptr, n: None)
    m.decode(list(range(5, 21)), list(range(16)), want_hidden=True)
    m.decode([77], [16])
    m.decode([90, 91, 92, 93, 94, 95], [17, 18, 19, 20, 21, 22])               # the 6-token chain verification
    out = []
    for t in m.nodes():
        info = ea.tensor_info(t)
        rec = {"info": info, "val": None, "w": None, "x": None}
        if info["op"] == OP_MUL_MAT and not info["name"]:
            info["name"] = ea.tensor_info(ea.tensor_src(t, 0))["name"].replace("blk.", "").replace(".weight", "")
        contiguous = info["nb"][0] == 4 and info["nb"][1] == 4 * info["ne"][0] and info["nb"][2] == info["nb"][1] * info["ne"][1]
        if info["type"] == 0 and contiguous:
            rec["val"] = m.read_tensor(t)
        if info["op"] == OP_MUL_MAT:
            w = ea.tensor_src(t, 0); x = ea.tensor_src(t, 1)
            wi = ea.tensor_info(w); xi = ea.tensor_info(x)
            if wi["type"] in QUANT and xi["type"] == 0 and xi["nb"][0] == 4 and xi["nb"][1] == 4 * xi["ne"][0]:
                rec["w"] = wi; rec["x"] = m.read_tensor(x).reshape(-1, xi["ne"][0])[:xi["ne"][1]]
        out.append(rec)
    m.close()
    return out


def q_int8(x, wtype):
    """the int8 activation values the reference derives from a row (Q8_K for K-quants, Q8_0 for q8_0 / q4_0)"""
    if wtype in (12, 13, 14):
        b = orc.quantize_q8_K(x).reshape(-1, 292); return b[:, 4:260].view(np.int8).reshape(-1)
    b = orc.quantize_q8_0(x).reshape(-1, 34); return b[:, 2:].view(np.int8).reshape(-1)


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "llama-2-13b-tp2-rank0"
    dims, ftype, tp = CASES[case]
    ea = load_package()
    g = run(ea, ea.Backend.mi355x(0), dims, ftype, tp)
    a = run(ea, refapi.reference_cpu(ea), dims, ftype, tp)
    s = run(ea, refapi.reference_cpu(ea, scalar=True), dims, ftype, tp)
    assert len(g) == len(a) == len(s)
    print(f"# flip replay: {case}, the 6-token verification decode, {len(g)} nodes; plugin unfused vs reference AVX2 vs reference scalar")
    print(f"# {'node':>4} {'name':<22} {'op':>3} {'shape':<18} {'gpu-avx2':>10} {'gpu-scal':>10} {'avx2-scal':>10}")
    first = None
    flips_total = [0, 0, 0]
    for i, (x, y, z) in enumerate(zip(g, a, s)):
        info = x["info"]
        if x["val"] is None or y["val"] is None:
            continue
        e = (l2(x["val"], y["val"]), l2(x["val"], z["val"]), l2(y["val"], z["val"]))
        shape = "x".join(str(v) for v in info["ne"] if v != 1) or "1"
        line = f"  {i:4d} {info['name'][:22]:<22} {info['op']:3d} {shape:<18} {e[0]:10.2e} {e[1]:10.2e} {e[2]:10.2e}"
        if x["w"] is not None:
            wt = x["w"]["type"]
            ein = (l2(x["x"], y["x"]), l2(x["x"], z["x"]), l2(y["x"], z["x"]))
            qg, qa, qs = (np.concatenate([q_int8(r, wt) for r in v["x"]]) for v in (x, y, z))
            nd = (int((qg != qa).sum()), int((qg != qs).sum()), int((qa != qs).sum()))
            for j in range(3):
                flips_total[j] += nd[j]
            line += f"   | {QUANT[wt]} MUL_MAT: inputs {ein[0]:.1e} {ein[1]:.1e} {ein[2]:.1e} -> int8 values differing (of {qg.size}) {nd[0]} {nd[1]} {nd[2]}"
            if first is None and max(e) > 1e-5:
                first = (i, info["name"], ein, nd, e, qg.size)
        print(line)
    print()
    if first:
        i, nm, ein, nd, e, n = first
        print(f"first quantised MUL_MAT whose outputs differ by more than 1e-5: node {i} ({nm})")
        print(f"  inputs differ by     gpu-avx2 {ein[0]:.2e}   gpu-scalar {ein[1]:.2e}   avx2-scalar {ein[2]:.2e}")
        print(f"  int8 values flipped  gpu-avx2 {nd[0]}   gpu-scalar {nd[1]}   avx2-scalar {nd[2]}   (of {n})")
        print(f"  outputs differ by    gpu-avx2 {e[0]:.2e}   gpu-scalar {e[1]:.2e}   avx2-scalar {e[2]:.2e}")
    print(f"int8 activation values that differ, summed over all quantised MUL_MATs of the decode: gpu-avx2 {flips_total[0]}, gpu-scalar {flips_total[1]}, avx2-scalar {flips_total[2]}")
    last = [r for r in zip(g, a, s) if r[0]["val"] is not None][-1]
    print(f"last node ({last[0]['info']['name']}): gpu-avx2 {l2(last[0]['val'], last[1]['val']):.2e}  gpu-scalar {l2(last[0]['val'], last[2]['val']):.2e}  avx2-scalar {l2(last[1]['val'], last[2]['val']):.2e}")


if __name__ == "__main__":
    main()
