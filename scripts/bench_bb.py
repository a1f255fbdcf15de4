"""Per-shape time of the big-batch product (25+ tokens) with the plugin's HIP-event hooks: python scripts/bench_bb.py [T ...]
(GGML_MI355X_BB_OLD=1: round 2's kernel; GGML_MI355X_BB_NQB=n: token tiles per block)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_package
import numpy as np, qdata
ea = load_package(); gpu = ea.Backend.mi355x(0)
lib = C.CDLL(ea.require_plugin())
lib.ggml_backend_mi355x_profile_end.restype = C.c_int; lib.ggml_backend_mi355x_profile_end.argtypes = [C.POINTER(C.c_double)]
rng = np.random.default_rng(0)
NAMES = {12: "q4_K", 14: "q6_K", 8: "q8_0"}
def run(t, rows, k, T):
    rb = k // ea.TYPE_TRAITS[t][0] * ea.TYPE_TRAITS[t][1]
    nrep = max(2, int(320e6 // (rows * rb)) + 1)
    g = ea.Graph(gpu); ws, outs = [], []
    x = g.tensor(ea.F32, k, T)
    for i in range(nrep):
        a = g.tensor(t, k, rows); ws.append(a); outs.append(g.mul_mat(a, x))
    g.alloc()
    blk = qdata.random_blocks(t, rows, k, rng)
    for w in ws: g.set(w, blk)
    g.set(x, rng.standard_normal((T, k)).astype(np.float32))
    g.compute()
    best = None
    for it in range(3):
        lib.ggml_backend_mi355x_profile_begin(); g.compute()
        out = (C.c_double * 4)(); n = lib.ggml_backend_mi355x_profile_end(out)
        us = out[0] * 1e3 / n
        if best is None or us < best: best = us
    tops = 2.0 * rows * k * T / (best * 1e-6) / 1e12
    print("%-5s rows %6d k %6d T %3d: %8.2f us per launch, %6.1f GB/s of weights, %6.1f Top/s" % (NAMES[t], rows, k, T, best, rows * rb / best / 1e3, tops), flush=True)
Ts = [int(a) for a in sys.argv[1:]] or [69, 128]
for T in Ts:
    for t in (12, 8, 14):
        for rows, k in ((4096, 4096), (11008, 4096), (4096, 11008)):
            run(t, rows, k, T)
