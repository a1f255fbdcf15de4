"""Tree driver on the synthetic 7B pair: tokens/s, forks, verify batch, per-round phase times; EH_TREE_DEBUG=1 prints the candidate distributions.
    python scripts/tree_probe.py [ftype] [n_seq_dft] [n_draft] [p_split] [temp_dft] [top_k] [n_tokens]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_package
import numpy as np
ea = load_package(); be = ea.Backend.mi355x(0)
a = sys.argv[1:] + [None] * 8
ftype = a[0] or "q4_k_m"; np_ = int(a[1] or 4); nd = int(a[2] or 8); ps = float(a[3] or 0.02); td = float(a[4] or 4.0); tk = int(a[5] or 8); nt = int(a[6] or 96)
rng = np.random.default_rng(1234); prompt = [1] + [int(v) for v in rng.integers(5, 31999, 127)]
tgt = ea.Model(be, "vicuna-7b", ftype, n_ctx=2048, seed=42)
dft = ea.Model(be, "vicuna-7b", ftype, n_ctx=2048, eagle_of=tgt, seed=42, accept_p=0.8)
ts = ea.TreeSession(tgt, dft, prompt, n_seq_dft=np_, n_draft=nd, p_split=ps, temp=0.0, temp_dft=td, top_k=tk)
ts.run(8)
t0 = time.perf_counter(); tt, st = ts.run(nt); d = time.perf_counter() - t0
it = max(1.0, st["n_iters"])
print(f"{ftype} np {np_} draft-max {nd} p_split {ps} temp_dft {td} top_k {tk}: {len(tt)/d:.1f} tokens/s, {st['n_predict']/it:.2f} tokens/round, forks {int(st['n_forks'])}, "
      f"max verify batch {int(st['max_batch'])}, draft {st['t_draft_us']/it/1e3:.2f} ms/round ({st['n_draft_calls']/it:.1f} decodes), verify {st['t_verify_us']/it/1e3:.2f} ms/round")
for nm, m in (("draft", dft), ("target", tgt)):
    t = m.timers()
    n = max(1.0, t[4]) if not isinstance(t, dict) else max(1.0, t.get("n_decode", 1))
    print(f"  {nm} model host timers per decode: {t}")
