"""Host-side time split of the speculative rounds bench.py times: python scripts/host_timers.py [rounds]
(per round, us: graph build | input image + upload | launch issue | wait, for the draft chain and the verification pass)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_package
import numpy as np
ea = load_package(); be = ea.Backend.mi355x(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
tgt = ea.Model(be, "vicuna-7b", "q4_k_m", n_ctx=2048, seed=42)
dft = ea.Model(be, "vicuna-7b", "q4_k_m", n_ctx=2048, eagle_of=tgt, seed=42, accept_p=0.8)
prompt = [int(x) for x in np.random.default_rng(1234).integers(5, 31000, 128)]
s = ea.SpecSession(tgt, dft, prompt)
s.rounds(4, n_draft=5)
tgt.timers(reset=True); dft.timers(reset=True)
t0 = time.perf_counter(); toks, st = s.rounds(R, n_draft=5); dt = time.perf_counter() - t0
for name, m in (("draft chain", dft), ("verification", tgt)):
    t = m.timers()
    print(name, {k: round(v / R, 1) if k != "n_decode" else v for k, v in t.items()})
print("round %.1f us; driver: draft phase %.1f us, verify phase %.1f us" % (dt / R * 1e6, st["t_draft_us"] / R, st["t_verify_us"] / R))
