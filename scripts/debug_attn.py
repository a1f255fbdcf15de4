import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_package
import numpy as np
import refapi
ea = load_package()
gpu = ea.Backend.mi355x(0); cpu = refapi.reference_cpu(ea, threads=8)
rng = np.random.default_rng(5)
def rel(a, b): return float(np.abs(a.astype(np.float64)-b.astype(np.float64)).max()/(np.abs(b).max()+1e-30))
for (T, n_kv, H, Hkv, D) in [(16, 32, 4, 4, 64), (16, 32, 8, 2, 64), (6, 96, 4, 2, 64), (16, 160, 32, 32, 128)]:
    q = rng.standard_normal((T, H, D)).astype(np.float32)
    kc = rng.standard_normal((Hkv, n_kv, D)).astype(np.float16); vc = rng.standard_normal((Hkv, D, n_kv)).astype(np.float16)
    mask = np.full((64, n_kv), -np.inf, np.float32)
    for j in range(T): mask[j, :j+1+(n_kv-T)] = 0
    res = {}
    for name, be in (("gpu", gpu), ("cpu", cpu)):
        g = ea.Graph(be)
        tq = g.tensor(ea.F32, D, H, T); tk = g.tensor(ea.F16, D, n_kv, Hkv); tv = g.tensor(ea.F16, n_kv, D, Hkv); tm = g.tensor(ea.F32, n_kv, 64)
        kq = g.mul_mat(tk, g.permute(tq, 0, 2, 1, 3)); sm = g.soft_max(kq, tm, 1.0/np.sqrt(D)); kqv = g.mul_mat(tv, sm)
        out = g.cont(g.permute(kqv, 0, 2, 1, 3))
        g.alloc(); g.set(tq, q); g.set(tk, kc); g.set(tv, vc); g.set(tm, mask); g.compute()
        res[name] = [g.get(kq).copy(), g.get(sm).copy(), g.get(kqv).copy(), g.get(out).copy()]
    print((T, n_kv, H, Hkv, D), "kq %.2e sm %.2e kqv %.2e out %.2e" % tuple(rel(a, b) for a, b in zip(res["gpu"], res["cpu"])))
# per-layer drift inside the tiny model
for ft in ("q8_0",):
    outs = {}
    for name, be in (("gpu", gpu), ("cpu", cpu)):
        m = ea.Model(be, "tiny", ft, n_ctx=256, seed=3, predictable=False)
        lg, hid = m.decode(list(range(5, 21)), list(range(16)))
        outs[name] = (lg, hid); m.close()
    for r in range(16):
        print("row", r, "logits rel %.2e hidden rel %.2e" % (rel(outs["gpu"][0][r], outs["cpu"][0][r]), rel(outs["gpu"][1][r], outs["cpu"][1][r])))
