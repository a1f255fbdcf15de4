import sys, time; sys.path.insert(0,'/root/repo/tests')
from conftest import load_package
import numpy as np
ea = load_package()
be = ea.Backend.mi355x(0)
print(be.name, be.description, flush=True)
t0=time.time()
tgt = ea.Model(be, "vicuna-7b", "q4_k_m", n_ctx=2048, seed=42)
dft = ea.Model(be, "vicuna-7b", "q4_k_m", n_ctx=2048, eagle_of=tgt, seed=42, accept_p=0.8)
print("models built in", time.time()-t0, "s; weight bytes", tgt.weight_bytes/1e9, dft.weight_bytes/1e9, flush=True)
rng = np.random.default_rng(1234)
prompt = [1]+[int(x) for x in rng.integers(5, 32000, 127)]
for rep in range(2):
    tgt.timers(reset=True)
    plain, ps = ea.plain_generate(tgt, prompt, 128)
    tm = tgt.timers()
    print("plain: %.1f tok/s  prompt %.1f ms; per-decode: build %.1f us upload %.1f compute %.1f download %.1f nodes %d issue %.1f wait %.1f" % (ps["n_predict"]/ps["t_decode_us"]*1e6, ps["t_prompt_us"]/1e3, tm["build_us"]/tm["n_decode"], tm["upload_us"]/tm["n_decode"], tm["compute_us"]/tm["n_decode"], tm["download_us"]/tm["n_decode"], tgt.n_nodes, tm["issue_us"]/tm["n_decode"], tm["wait_us"]/tm["n_decode"]), flush=True)
for nd in (3,5,7):
    tgt.timers(reset=True); dft.timers(reset=True)
    spec, ss = ea.spec_generate(tgt, dft, prompt, 128, n_draft=nd)
    print("spec n_draft=%d: %.1f tok/s accept %d/%d iters %d draft %.1f ms verify %.1f ms same=%s" % (nd, ss["n_predict"]/ss["t_decode_us"]*1e6, ss["n_accept"], ss["n_drafted"], ss["n_iters"], ss["t_draft_us"]/1e3, ss["t_verify_us"]/1e3, spec[:len(plain)]==plain[:len(spec)]), flush=True)
    tm = dft.timers(); print("   draft per-decode: build %.1f upload %.1f compute %.1f download %.1f (n=%d)" % (tm["build_us"]/tm["n_decode"], tm["upload_us"]/tm["n_decode"], tm["compute_us"]/tm["n_decode"], tm["download_us"]/tm["n_decode"], tm["n_decode"]))
    tm = tgt.timers(); print("   target per-decode: build %.1f upload %.1f compute %.1f download %.1f (n=%d)" % (tm["build_us"]/tm["n_decode"], tm["upload_us"]/tm["n_decode"], tm["compute_us"]/tm["n_decode"], tm["download_us"]/tm["n_decode"], tm["n_decode"]))
