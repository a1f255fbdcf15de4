"""Per-shape roofline of the quantised mat-vec kernel, measured with the plugin's HIP-event hooks.
Each graph multiplies NREP distinct weight tensors (> 256 MiB in total, so nothing is served from the Infinity Cache)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_package
import numpy as np, qdata
ea = load_package()
gpu = ea.Backend.mi355x(0)
lib = C.CDLL(ea.require_plugin())
lib.ggml_backend_mi355x_profile_end.restype = C.c_int; lib.ggml_backend_mi355x_profile_end.argtypes = [C.POINTER(C.c_double)]
rng = np.random.default_rng(0)
NAMES = {12: "q4_K", 14: "q6_K", 8: "q8_0", 2: "q4_0", 13: "q5_K"}
def run(t, rows, k, T, mode, nrep=None):
    rb = k // ea.TYPE_TRAITS[t][0] * ea.TYPE_TRAITS[t][1]
    per = rows * rb * (2 if mode == "swiglu" else (3 if mode == "qkv" else 1))
    nrep = nrep or max(2, int(320e6 // per) + 1)
    g = ea.Graph(gpu)
    x = g.tensor(ea.F32, k, T); nw = g.tensor(ea.F32, k)
    ws, outs, xs = [], [], []
    for i in range(nrep):
        if mode == "norm" or mode == "swiglu" or mode == "qkv":
            xin = g.mul(g.rms_norm(x, 1e-6), nw)
        else:
            xin = x
        if mode == "swiglu":
            a = g.tensor(t, k, rows); b = g.tensor(t, k, rows); ws += [a, b]
            gate = g.unary(g.mul_mat(a, xin), "silu"); up = g.mul_mat(b, xin); outs.append(g.mul(gate, up))
        elif mode == "qkv":
            a = g.tensor(t, k, rows); b = g.tensor(t, k, rows); c = g.tensor(t, k, rows); ws += [a, b, c]
            outs += [g.mul_mat(a, xin), g.mul_mat(b, xin), g.mul_mat(c, xin)]
        elif mode == "single":      # one matrix per launch: own activations, so nothing is merged
            xi = g.tensor(ea.F32, k, T); xs.append(xi)
            a = g.tensor(t, k, rows); ws.append(a); outs.append(g.mul_mat(a, xi))
        else:
            a = g.tensor(t, k, rows); ws.append(a); outs.append(g.mul_mat(a, xin))
    g.alloc()
    blk = qdata.random_blocks(t, rows, k, rng)
    for w in ws: g.set(w, blk)
    g.set(x, rng.standard_normal((T, k)).astype(np.float32)); g.set(nw, np.ones(k, np.float32))
    for xi in xs: g.set(xi, rng.standard_normal((T, k)).astype(np.float32))
    g.compute()
    best = None
    for it in range(3):
        lib.ggml_backend_mi355x_profile_begin(); g.compute()
        out = (C.c_double * 4)(); n = lib.ggml_backend_mi355x_profile_end(out)
        gbs = out[1] / (out[0] * 1e-3) / 1e9
        if best is None or gbs > best[0]: best = (gbs, out[0] * 1e3 / n, n)
    print("%-5s rows %6d k %6d T %d %-7s launches %3d avg %7.2f us  %7.1f GB/s (%4.1f%% of 8 TB/s)" % (NAMES[t], rows, k, T, mode, best[2], best[1], best[0], best[0] / 80), flush=True)
shapes = [(4096, 4096), (11008, 4096), (4096, 11008), (32000, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "q8":          # Q8_0 shapes (config 3), T = 1 / 6 / 24
    for T in (1, 6, 24):
        for rows, k in ((4096, 4096), (11008, 4096), (4096, 11008)): run(8, rows, k, T, "plain")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "tg":          # 8 tokens (one group, in-kernel quantiser) against 9 / 16 / 24 (token groups, image launch): the a2 path
    for t in (12, 8):
        for T in (8, 9, 16, 24):
            for rows, k in ((4096, 4096), (4096, 11008), (32000, 4096)): run(t, rows, k, T, "single")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "single":      # the model's single-matrix launches (wo, ffn_down, lm_head) at T = 6
    for t in (12, 14):
        for rows, k in ((4096, 4096), (4096, 11008), (32000, 4096)): run(t, rows, k, 6, "single")
    run(12, 11008, 4096, 6, "swiglu")
    sys.exit(0)
for T in (1, 6):
    for t in (12, 14):
        for rows, k in shapes:
            run(t, rows, k, T, "plain")
    run(12, 4096, 4096, T, "norm")
    run(12, 4096, 4096, T, "qkv")
    run(12, 11008, 4096, T, "swiglu")
    run(8, 4096, 4096, T, "plain"); run(2, 4096, 4096, T, "plain")
