"""In-kernel phase stamps of k_mmt (diagnostic build, GGML_MI355X_MMT_STAMPS=1): where a launch spends its time.
    python scripts/mmt_stamps.py [T] [rows] [k] [mode]
Phases (100 MHz s_memrealtime, per wave, lane 0): 0 entry | 1 own share of the activation image quantised | 2 after the prologue barrier |
3 first row group's units done | 4 after the reduction barrier | 5 first group's epilogue done | 6 kernel exit."""
import os; os.environ.setdefault("EH_LAB_PLUGIN", "1")      # lab knobs / stamp kernels live in the --lab build of the plugin only
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ["GGML_MI355X_MMT_STAMPS"] = "1"
from conftest import load_package
import numpy as np, qdata
ea = load_package(); gpu = ea.Backend.mi355x(0)
lib = C.CDLL(ea.require_plugin())
lib.ggml_backend_mi355x_mmt_stamps.restype = C.c_int; lib.ggml_backend_mi355x_mmt_stamps.argtypes = [C.POINTER(C.c_uint64)]
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
k = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
mode = sys.argv[4] if len(sys.argv) > 4 else "norm"
t = 12
rng = np.random.default_rng(0)
g = ea.Graph(gpu)
nw = g.tensor(ea.F32, k); ws = []; xs = []; outs = []
nrep = 40
for i in range(nrep):                                   # independent launches, distinct weights (> 256 MiB in total)
    xi = g.tensor(ea.F32, k, T); xs.append(xi)
    xin = g.mul(g.rms_norm(xi, 1e-6), nw) if mode == "norm" else xi
    a = g.tensor(t, k, rows); ws.append(a); outs.append(g.mul_mat(a, xin))
g.alloc()
blk = qdata.random_blocks(t, rows, k, rng)
for w in ws: g.set(w, blk)
g.set(nw, np.ones(k, np.float32))
for xi in xs: g.set(xi, rng.standard_normal((T, k)).astype(np.float32))
g.compute(); g.compute()
NS = 12
n = 4 * 256 * 16 * NS
buf = (C.c_uint64 * n)()
assert lib.ggml_backend_mi355x_mmt_stamps(buf) == n
st = np.frombuffer(buf, dtype=np.uint64).reshape(4, 256, 16, NS).astype(np.int64)
# the last four launches of the graph sit in the ring; order them by entry time
order = np.argsort([st[i, :, :, 0][st[i, :, :, 0] > 0].min() for i in range(4)])
print(f"q4_K rows {rows} k {k} T {T} mode {mode}: stamps in us relative to the first wave of the launch (min / median / max over waves)")
names = ["entry", "image share done", "after prologue barrier", "first group units done", "after reduce barrier", "first epilogue done", "exit"]
prev_exit = None
for i in order:
    s = st[i]; valid = s[:, :, 0] > 0
    t0 = s[:, :, 0][valid].min()
    line = []
    for j in range(7):
        v = (s[:, :, j][valid] - t0) / 100.0
        line.append("%s %.2f/%.2f/%.2f" % (names[j], v.min(), np.median(v), v.max()))
        if j == 0:      # inside the quantiser (stamp build only: the activation loads are waited for before the weight prefetch goes out)
            for jj, nm in ((7, "activations landed"), (8, "partial sums parked"), (9, "after norm barrier"), (10, "norm scale known")):
                vv = s[:, :, jj][valid]
                if (vv > 0).any(): vv = (vv[vv > 0] - t0) / 100.0; line.append("%s %.2f/%.2f/%.2f" % (nm, vv.min(), np.median(vv), vv.max()))
    gap = "" if prev_exit is None else "  [gap from previous launch's last exit to this first entry: %.2f us]" % ((t0 - prev_exit) / 100.0)
    print(" | ".join(line) + gap)
    prev_exit = s[:, :, 6][valid].max()
