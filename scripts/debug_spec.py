import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import load_package
import numpy as np
import refapi
ea = load_package()
which = sys.argv[1] if len(sys.argv) > 1 else "gpu"
cfgname = sys.argv[2] if len(sys.argv) > 2 else "tiny"
be = ea.Backend.mi355x(0) if which == "gpu" else refapi.reference_cpu(ea, threads=8)
tgt = ea.Model(be, cfgname, "q4_k_m", n_ctx=512, seed=42)
dft = ea.Model(be, cfgname, "q4_k_m", n_ctx=512, eagle_of=tgt, seed=42, accept_p=0.8)
V = tgt.n_vocab
rng = np.random.default_rng(1234)
prompt = [1]+[int(x) for x in rng.integers(5, V, 31)]
plain, ps = ea.plain_generate(tgt, prompt, 40)
print("plain", plain[:24], "uniq", len(set(plain)))
s = ea.SpecSession(tgt, dft, prompt)
for it in range(8):
    toks, st = s.rounds(1, n_draft=5)
    print("round", it, "out", toks, "accepted", st["n_accept"], "drafted", st["n_drafted"])
# direct check of the draft on single tokens: does it predict what the target predicts?
tgt.kv_clear(); dft.kv_clear()
good = 0
for i, t in enumerate(plain[:20]):
    lg, hid = tgt.decode([t], [i])
    lgd, _ = dft.decode([t], [i+1], hidd=hid)
    a, b = int(lg[0].argmax()), int(lgd[0].argmax())
    good += a == b
    print(t, "->", a, b, "top2 gap tgt %.2f dft %.2f" % (np.sort(lg[0])[-1]-np.sort(lg[0])[-2], np.sort(lgd[0])[-1]-np.sort(lgd[0])[-2]))
print("draft agrees on", good, "of 20")
