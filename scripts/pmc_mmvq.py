"""Tiny workload for rocprofv3 --pmc: a handful of mat-vec launches at T=1 and T=6 (Q4_K, 11008x4096 and 4096x4096)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_package
import numpy as np, qdata
ea = load_package()
gpu = ea.Backend.mi355x(0)
rng = np.random.default_rng(0)
for T in (1, 6):
    for (rows, k) in ((11008, 4096), (4096, 11008)):
        g = ea.Graph(gpu)
        x = g.tensor(ea.F32, k, T); ws = []; outs = []
        for i in range(8):
            a = g.tensor(12, k, rows); ws.append(a); outs.append(g.mul_mat(a, x))
        g.alloc()
        blk = qdata.random_blocks(12, rows, k, rng)
        for w in ws: g.set(w, blk)
        g.set(x, rng.standard_normal((T, k)).astype(np.float32))
        g.compute(); g.compute()
print("ok")
