"""Wall time per dependent tiny kernel (chain of small ADDs) and per small quantise+mat-vec pair: the per-boundary floor of this stack."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_package
import numpy as np, qdata
ea = load_package()
gpu = ea.Backend.mi355x(0)
rng = np.random.default_rng(0)
for n in (100, 400):
    g = ea.Graph(gpu)
    x = g.tensor(ea.F32, 4096, 6); y = g.tensor(ea.F32, 4096, 6)
    cur = x
    for i in range(n): cur = g.add(cur, y)
    g.alloc(); g.set(x, np.zeros((6, 4096), np.float32)); g.set(y, np.ones((6, 4096), np.float32))
    g.compute()
    best = 1e9
    for it in range(5):
        t0 = time.perf_counter(); g.compute(); dt = time.perf_counter() - t0; best = min(best, dt)
    print("chain of %d tiny ADD kernels: %.2f us per kernel (wall %.0f us)" % (n, best*1e6/n, best*1e6), flush=True)
