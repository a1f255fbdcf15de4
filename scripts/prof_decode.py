"""Small workload for rocprofv3: Vicuna-7B-shaped Q4_K_M target + EAGLE head, a few plain steps and a few speculative rounds."""
import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import load_package
import numpy as np
ea = load_package()
be = ea.Backend.mi355x(0)
tgt = ea.Model(be, "vicuna-7b", "q4_k_m", n_ctx=1024, seed=42)
dft = ea.Model(be, "vicuna-7b", "q4_k_m", n_ctx=1024, eagle_of=tgt, seed=42, accept_p=0.8)
rng = np.random.default_rng(1234)
prompt = [1]+[int(x) for x in rng.integers(5, 32000, 127)]
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
if mode in ("plain", "both"):
    plain, ps = ea.plain_generate(tgt, prompt, 24)
    print("plain tok/s", ps["n_predict"]/ps["t_decode_us"]*1e6)
if mode in ("spec", "both"):
    spec, ss = ea.spec_generate(tgt, dft, prompt, 48, n_draft=5)
    print("spec tok/s", ss["n_predict"]/ss["t_decode_us"]*1e6, ss)
