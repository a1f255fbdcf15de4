"""Which mat-vec kernel serves a (type, T, k) product (GGML_MI355X_DEBUG_MMQ=1)."""
import os; os.environ.setdefault("EH_LAB_PLUGIN", "1")      # lab knobs / stamp kernels live in the --lab build of the plugin only
import sys, os
os.environ["GGML_MI355X_DEBUG_MMQ"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_package
import numpy as np, qdata
ea = load_package(); gpu = ea.Backend.mi355x(0); rng = np.random.default_rng(0)
for t in (8, 12):
    for T in (6, 24):
        g = ea.Graph(gpu); x = g.tensor(ea.F32, 4096, T); a = g.tensor(t, 4096, 512); g.mul_mat(a, x)
        g.alloc(); g.set(a, qdata.random_blocks(t, 512, 4096, rng)); g.set(x, rng.standard_normal((T, 4096)).astype(np.float32)); g.compute()
