"""In-kernel phase stamps of k_attn_small (diagnostic build): python scripts/attn_stamps.py [T] [n_kv]
0 entry | 1 V prefetch + q conversion issued | 2 scores written | 3 after barrier | 4 soft-max done | 5 after barrier | 6 V.p done | 7 exit"""
import os; os.environ.setdefault("EH_LAB_PLUGIN", "1")      # lab knobs / stamp kernels live in the --lab build of the plugin only
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ["GGML_MI355X_ATTN_STAMPS"] = "1"
from conftest import load_package
import numpy as np
ea = load_package(); gpu = ea.Backend.mi355x(0)
lib = C.CDLL(ea.require_plugin())
lib.ggml_backend_mi355x_attn_stamps.restype = C.c_int; lib.ggml_backend_mi355x_attn_stamps.argtypes = [C.POINTER(C.c_uint64)]
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_kv = int(sys.argv[2]) if len(sys.argv) > 2 else 224
H, d = 32, 128
rng = np.random.default_rng(0)
g = ea.Graph(gpu)
outs = []; ins = []
for i in range(12):            # independent attention sub-graphs, distinct caches
    tq = g.tensor(ea.F32, d, H, T); tk = g.tensor(ea.F16, d, n_kv, H); tv = g.tensor(ea.F16, n_kv, d, H); tm = g.tensor(ea.F32, n_kv, (T + 63) // 64 * 64)
    kqt = g.mul_mat(tk, g.permute(tq, 0, 2, 1, 3)); sm = g.soft_max(kqt, tm, 1.0 / np.sqrt(d))
    outs.append(g.cont(g.permute(g.mul_mat(tv, sm), 0, 2, 1, 3))); ins.append((tq, tk, tv, tm))
g.alloc()
for tq, tk, tv, tm in ins:
    g.set(tq, rng.standard_normal((T, H, d)).astype(np.float32)); g.set(tk, rng.standard_normal((H, n_kv, d)).astype(np.float16))
    g.set(tv, rng.standard_normal((H, d, n_kv)).astype(np.float16)); g.set(tm, np.zeros(((T + 63) // 64 * 64, n_kv), np.float32))
g.compute(); g.compute()
n = 512 * 4 * 8
buf = (C.c_uint64 * n)(); assert lib.ggml_backend_mi355x_attn_stamps(buf) == n
st = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4, 8).astype(np.int64)
valid = st[:, :, 0] > 0
t0 = st[:, :, 0][valid].min()
names = ["entry", "q/V issued", "scores written", "after barrier", "soft-max done", "after barrier", "V.p done", "exit"]
print(f"attention T {T} n_kv {n_kv} heads {H}: {valid.sum()} waves; us relative to the first wave (min / median / max)")
for j in range(8):
    v = (st[:, :, j][valid] - t0) / 100.0
    print("  %-16s %.2f / %.2f / %.2f" % (names[j], v.min(), np.median(v), v.max()))
