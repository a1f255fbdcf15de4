"""Prompt pass (128 tokens, Vicuna-7B Q4_K_M shapes) wall time: python scripts/prompt_time.py   (GGML_MI355X_MMT_BB_TGW=1|2|4 selects the big-batch tiling)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_package
import numpy as np
ea = load_package(); be = ea.Backend.mi355x(0)
ft = sys.argv[1] if len(sys.argv) > 1 else "q4_k_m"
tgt = ea.Model(be, "vicuna-7b", ft, n_ctx=2048, seed=42)
prompt = [int(x) for x in np.random.default_rng(1234).integers(5, 31000, 128)]
ea.plain_generate(tgt, prompt, 4)
ts = []
for _ in range(5):
    toks, st = ea.plain_generate(tgt, prompt, 4); ts.append(st["t_prompt_us"] / 1e3)
print("prompt(128) ms: min %.2f median %.2f  [TGW=%s, %s]  first tokens %s" % (min(ts), sorted(ts)[2], os.environ.get("GGML_MI355X_MMT_BB_TGW", "default"), ft, toks[:4]))
