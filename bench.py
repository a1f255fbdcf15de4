#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X EAGLE speculative-decoding path.

Metric (BASELINE.json): accepted tokens/sec + accept-rate, Vicuna-7B Q4_K_M + EAGLE draft head, tree depth 5
(configs[1]).  One "step" = one speculative round = `n_draft` autoregressive EAGLE-head forwards (mat-vec,
T <= accepted+1) + ONE batched target verification forward (T = n_draft+1) + greedy acceptance + KV fix-up.
`value` = tokens emitted (accepted drafts + the bonus token of every round) per second over exactly K rounds.

Weights are synthetic (no checkpoints / network here): random valid quant blocks with the exact tensor shapes and
the exact Q4_K_M type mix of Vicuna-7B (4.0 GB of mat-mul weights streamed per target forward), constructed so
that the target's greedy continuation is a fixed permutation of the vocabulary and the draft head predicts it
with probability `accept_p` per token -- acceptance is decided by the real logits our kernels compute
(eagle-in-llama.cpp_amd/host/model.cpp); `accept_rate` in the JSON line is measured, not assumed.

    python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1 (one process per GPU, launched by torch.distributed.run): see DESIGN.md "multi-GPU".
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_DRAFT = 5
PROMPT_LEN = 128
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy)


def load_pkg():
    import importlib.util
    p = os.path.join(ROOT, "eagle-in-llama.cpp_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("eagle_amd", p, submodule_search_locations=[os.path.dirname(p)])
    m = importlib.util.module_from_spec(spec)
    sys.modules["eagle_amd"] = m
    spec.loader.exec_module(m)
    return m


def prompt_tokens(seed, n=PROMPT_LEN, vocab=32000):
    import numpy as np
    rng = np.random.default_rng(seed)
    return [1] + [int(x) for x in rng.integers(5, vocab, n - 1)]          # BOS + uniform{5..vocab-1}


def plugin_profile(ea):
    """HIP-event timing of every mat-vec launch inside the plugin (on the stream the kernels run on)."""
    lib = C.CDLL(ea.require_plugin())
    lib.ggml_backend_mi355x_profile_begin.restype = None
    lib.ggml_backend_mi355x_profile_end.restype = C.c_int
    lib.ggml_backend_mi355x_profile_end.argtypes = [C.POINTER(C.c_double)]
    return lib


def pmc_traffic():
    """HBM bytes per mat-vec launch from the committed rocprofv3 --pmc FETCH_SIZE pass of this same command
    (profiles/r01_pmc_traffic.json, written by scripts/pmc_summary.py; counters cannot be read from inside the run)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return json.load(f)["hbm_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(ea, cfg, ftype):
    """The reference's own ggml CPU backend (oracle/_ref, built from /root/reference) running the SAME driver on a
    bounded sample of the same workload.  Falls back to nothing (null) when oracle/_ref is absent."""
    import refapi                                                  # tests/refapi.py: the only place that knows oracle/_ref
    if not os.path.exists(refapi.REF_GGML_PATH):
        return None
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 4
    threads = max(1, min(threads, 32))
    be = refapi.reference_cpu(ea, threads=threads)
    t0 = time.time()
    tgt = ea.Model(be, cfg, ftype, n_ctx=256, seed=42)
    dft = ea.Model(be, cfg, ftype, n_ctx=256, eagle_of=tgt, seed=42, accept_p=0.8)
    prompt = prompt_tokens(1234, 16)
    s = ea.SpecSession(tgt, dft, prompt)
    rounds = 3
    toks, st = s.rounds(rounds, n_draft=N_DRAFT)
    s.close(); dft.close(); tgt.close()
    return {"value": round(st["n_predict"] / st["t_decode_us"] * 1e6, 3), "unit": "tokens/s", "cores": threads, "kind": "reference",
            "sample": f"{rounds} speculative rounds (n_draft={N_DRAFT}) after a 16-token prompt, same synthetic Vicuna-7B {ftype} + EAGLE head, "
                      f"reference ggml CPU backend (AVX2 build), {threads} threads; setup {time.time()-t0:.0f}s not timed",
            "accept_rate": round(st["n_accept"] / max(1.0, st["n_drafted"]), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", default="vicuna-7b")
    ap.add_argument("--ftype", default="q4_k_m")
    ap.add_argument("--accept-p", type=float, default=0.8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1 or os.environ.get("EH_FORCE_TP"):      # EH_FORCE_TP=1: rehearse the TP path with a 1-rank communicator
        from bench_tp import main_tp           # one process per GPU, row-split tensor parallel over RCCL
        return main_tp(args, rank, world, local)

    import torch                                # plumbing only: barrier-free at N=1, used for cuda.synchronize()
    ea = load_pkg()
    be = ea.Backend.mi355x(0)                   # raises if the HIP plugin is missing: no CPU fallback on the product path
    tgt = ea.Model(be, args.config, args.ftype, n_ctx=2048, seed=42)
    dft = ea.Model(be, args.config, args.ftype, n_ctx=2048, eagle_of=tgt, seed=42, accept_p=args.accept_p)
    prompt = prompt_tokens(1234)

    # non-speculative decode on the same model: the 1x the ">= 2x" target refers to
    plain_toks, pst = ea.plain_generate(tgt, prompt, 64)
    plain_tps = (pst["n_predict"] - 1) / pst["t_decode_us"] * 1e6

    sess = ea.SpecSession(tgt, dft, prompt)
    if args.warmup > 0:
        sess.rounds(args.warmup, n_draft=N_DRAFT)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks, st = sess.rounds(args.steps, n_draft=N_DRAFT)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    value = st["n_predict"] / dt

    # roofline of the dominant kernel (quantised mat-vec), same K rounds again with HIP events around every launch
    prof = plugin_profile(ea)
    prof.ggml_backend_mi355x_profile_begin()
    sess.rounds(args.steps, n_draft=N_DRAFT)
    out = (C.c_double * 4)()
    n_launch = prof.ggml_backend_mi355x_profile_end(out)
    raw_ms, alg_bytes, pair_ms = out[0], out[1], out[2]
    sess.close()
    # an event pair with nothing in between still measures `pair_ms` (queue markers): the kernel time is net of it
    kern_ms = max(raw_ms - pair_ms * n_launch, 1e-9)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if n_launch > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": "quantised mat-vec family: k_mmq (int8 MFMA, 2..8 tokens) + k_mmvq (dp4a, 1 token)", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(),
                "launches": n_launch, "avg_launch_us": round(kern_ms * 1e3 / max(1, n_launch), 2),
                "avg_launch_us_raw_event_pair": round(raw_ms * 1e3 / max(1, n_launch), 2), "empty_event_pair_us": round(pair_ms * 1e3, 2),
                "algorithmic_bytes_per_launch": round(alg_bytes / max(1, n_launch))}

    res = {"metric": "accepted tokens/sec + accept-rate, Vicuna-7B Q4_K_M + EAGLE, 1/8 GPU", "value": round(value, 2), "unit": "tokens/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "q4_K/q6_K x int8 -> int32 (MFMA / dp4a) -> f32", "data": "synthetic",
           "config": {"workload": f"{args.config} {args.ftype} target + EAGLE head, chain/tree depth {N_DRAFT} (verify batch {N_DRAFT+1}), "
                                  f"{PROMPT_LEN}-token synthetic prompt, greedy", "n_draft": N_DRAFT, "accept_p_synthetic": args.accept_p},
           "accept_rate": round(st["n_accept"] / max(1.0, st["n_drafted"]), 4), "tokens_per_round": round(st["n_predict"] / args.steps, 3),
           "plain_decode_tokens_per_s": round(plain_tps, 2), "speedup_vs_plain": round(value / plain_tps, 3),
           "target_weight_bytes": tgt.weight_bytes, "draft_weight_bytes": dft.weight_bytes,
           "roofline": roofline}
    if not args.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline(ea, args.config, args.ftype)
        except Exception as e:                                       # the baseline must never take the GPU number down with it
            res["cpu_baseline"] = {"error": str(e)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
