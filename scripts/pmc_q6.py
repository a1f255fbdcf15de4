"""Workload for rocprofv3 --pmc: the model's single-matrix Q6_K / Q4_K ffn_down launch at T = 6 (4096 x 11008), 4 launches each."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_package
import numpy as np, qdata
ea = load_package()
gpu = ea.Backend.mi355x(0)
rng = np.random.default_rng(0)
for t in (12, 14):
    g = ea.Graph(gpu); ws = []; xs = []
    for i in range(4):
        x = g.tensor(ea.F32, 11008, 6); a = g.tensor(t, 11008, 4096); xs.append(x); ws.append(a); g.mul_mat(a, x)
    g.alloc()
    blk = qdata.random_blocks(t, 4096, 11008, rng)
    for w in ws: g.set(w, blk)
    for x in xs: g.set(x, rng.standard_normal((6, 11008)).astype(np.float32))
    g.compute(); g.compute()
print("ok")
