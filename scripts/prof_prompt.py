"""rocprofv3 workload: 128-token prompt passes of the synthetic Vicuna-7B Q4_K_M target (big-batch path)."""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import load_package
import numpy as np
ea = load_package()
be = ea.Backend.mi355x(0)
ftype = sys.argv[1] if len(sys.argv) > 1 else "q4_k_m"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
tgt = ea.Model(be, "vicuna-7b", ftype, n_ctx=1024, seed=42)
rng = np.random.default_rng(1234)
for rep in range(4):
    tgt.kv_clear()
    toks = [1] + [int(x) for x in rng.integers(5, 32000, T - 1)]
    t0 = time.perf_counter(); tgt.decode(toks, list(range(T)), logits=[0] * (T - 1) + [1], want_hidden=False); dt = time.perf_counter() - t0
    print("prompt of %d tokens: %.2f ms" % (T, dt * 1e3), flush=True)
