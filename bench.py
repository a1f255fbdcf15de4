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
(eagle-in-llama.cpp_amd/host/model.cpp); `accept_rate` in the JSON line is measured, not assumed, and it is a
property of that synthetic dial, not of trained EAGLE weights (see `extra.accept_p_sweep` for tokens/s at other settings).

Roofline of the dominant kernel family (quantised mat-vec): algorithmic bytes per launch over
  * `achieved`: the rocprofv3 kernel-trace durations of the same K rounds, measured by running this very script as a child under
    `rocprofv3 --kernel-trace` (`--rocprof-child`; the plugin brackets the rounds with a marker kernel and counts launches and bytes
    without inserting anything between the kernels) -- the figure the committed profiles/ summary has to agree with;
  * `hip_events`: HIP events around every launch on the plugin's stream in this process (raw pairs and net of an empty pair).

    python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1: one process per GPU over RCCL (bench_tp.py).  Under `python -m torch.distributed.run ... bench.py --gpus N` every rank reads RANK /
WORLD_SIZE from the environment; a bare `python bench.py --gpus N` starts its own N ranks first (spawn_ranks) -- see DESIGN.md "multi-GPU".
"""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_DRAFT = 5
PROMPT_LEN = 128
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy)
INT8_PEAK_TOPS = 5000.0        # MI355X_MICROARCH.md, matrix cores: I8 runs at 2x the BF16 rate (~2.5 PF dense)
MATVEC_KERNELS = ("k_mmt<", "k_mmt2<", "k_mmt_bb<", "k_bb<", "k_mmq<", "k_mmvq<")


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


def plugin_lib(ea):
    lib = C.CDLL(ea.require_plugin())
    lib.ggml_backend_mi355x_profile_begin.restype = None
    lib.ggml_backend_mi355x_profile_end.restype = C.c_int
    lib.ggml_backend_mi355x_profile_end.argtypes = [C.POINTER(C.c_double)]
    lib.ggml_backend_mi355x_count_begin.restype = None
    lib.ggml_backend_mi355x_count_end.restype = C.c_long
    lib.ggml_backend_mi355x_count_end.argtypes = [C.POINTER(C.c_double)]
    return lib


def pmc_traffic(args):
    """HBM bytes per mat-vec launch, measured in THIS run: a second rocprofv3 child (`--pmc FETCH_SIZE`, counters in a pass of their own as
    MI355X_MICROARCH.md prescribes) runs the same speculative rounds between the plugin's marker kernels; FETCH_SIZE is in KiB and on
    gfx950 tallies the 128-byte requests of wide streaming reads at 64 bytes, hence x 1024 x 2.  Writes (T x rows x 4 bytes per launch) are
    not counted: WRITE_SIZE does not fit the same pass and is < 0.1 % of the traffic here."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    steps = max(2, min(args.steps, 10))
    cmd = [exe, "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
           "--steps", str(steps), "--warmup", "1", "--config", args.config, "--ftype", args.ftype, "--accept-p", str(args.accept_p)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
        files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
        if not files:
            return None, "no counter file: " + (r.stderr or r.stdout)[-200:]
        rows = [x for x in csv.DictReader(open(files[0])) if x.get("Counter_Name") == "FETCH_SIZE" or "k_profile_mark" in x.get("Kernel_Name", "")]
        rows.sort(key=lambda x: int(x["Dispatch_Id"]))
        marks = [i for i, x in enumerate(rows) if "k_profile_mark" in x["Kernel_Name"]]
        if len(marks) < 2:
            return None, "marker kernels not found in the counter file"
        sel = [x for x in rows[marks[-2] + 1:marks[-1]] if x.get("Counter_Name") == "FETCH_SIZE" and any(k in x["Kernel_Name"] for k in MATVEC_KERNELS)]
        if not sel:
            return None, "no mat-vec dispatches between the markers"
        kib = sum(float(x["Counter_Value"]) for x in sel)
        keep = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(keep):
            try:
                shutil.copy(files[0], os.path.join(keep, "bench_pmc_counter_collection.csv"))
            except Exception:
                pass
        return round(kib * 1024 * 2 / len(sel)), f"rocprofv3 --pmc FETCH_SIZE child of this run: {len(sel)} mat-vec launches of {steps} rounds, KiB x 1024 x 2 (gfx950 correction)"
    except Exception as e:
        return None, str(e)
    finally:
        shutil.rmtree(out, ignore_errors=True)


def cpu_baseline(ea, cfg, ftype, rounds=10):
    """The reference's own ggml CPU backend (oracle/_ref, built from /root/reference) running the SAME driver on the same
    128-token prompt for `rounds` speculative rounds.  Falls back to nothing (null) when oracle/_ref is absent."""
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
    tgt = ea.Model(be, cfg, ftype, n_ctx=512, seed=42)
    dft = ea.Model(be, cfg, ftype, n_ctx=512, eagle_of=tgt, seed=42, accept_p=0.8)
    s = ea.SpecSession(tgt, dft, prompt_tokens(1234))
    t_setup = time.time() - t0
    toks, st = s.rounds(rounds, n_draft=N_DRAFT)
    s.close(); dft.close(); tgt.close()
    return {"value": round(st["n_predict"] / st["t_decode_us"] * 1e6, 3), "unit": "tokens/s", "cores": threads, "kind": "reference",
            "sample": f"{rounds} speculative rounds (n_draft={N_DRAFT}) after the same {PROMPT_LEN}-token prompt, same synthetic Vicuna-7B {ftype} + EAGLE head, "
                      f"reference ggml CPU backend (AVX2 build), {threads} threads; model build + prompt pass {t_setup:.0f}s not timed",
            "accept_rate": round(st["n_accept"] / max(1.0, st["n_drafted"]), 4)}


def rocprof_child(args):
    """Inner run for rocprofv3: build the models, prompt, warm-up, then K rounds bracketed by the plugin's marker kernel."""
    ea = load_pkg()
    lib = plugin_lib(ea)
    be = ea.Backend.mi355x(0)
    tgt = ea.Model(be, args.config, args.ftype, n_ctx=2048, seed=42)
    dft = ea.Model(be, args.config, args.ftype, n_ctx=2048, eagle_of=tgt, seed=42, accept_p=args.accept_p)
    sess = ea.SpecSession(tgt, dft, prompt_tokens(1234))
    sess.rounds(max(args.warmup, 1), n_draft=N_DRAFT)
    lib.ggml_backend_mi355x_count_begin()
    sess.rounds(args.steps, n_draft=N_DRAFT)
    nbytes = C.c_double(0)
    n = lib.ggml_backend_mi355x_count_end(C.byref(nbytes))
    sess.close()
    print(json.dumps({"launches": int(n), "alg_bytes": nbytes.value}), flush=True)


def rocprof_roofline(args):
    """Runs this script under rocprofv3 --kernel-trace as a CHILD process and cuts the trace at the marker kernels."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix="bench_rocprof_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--rocprof-child",
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--config", args.config, "--ftype", args.ftype, "--accept-p", str(args.accept_p)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")
        info = None
        for line in r.stdout.splitlines():
            if line.startswith("{") and "alg_bytes" in line:
                info = json.loads(line)
        files = glob.glob(out + "/**/*_kernel_trace.csv", recursive=True)
        if info is None or not files:
            return {"error": (r.stderr or r.stdout)[-300:]}
        rows = list(csv.DictReader(open(files[0])))
        rows.sort(key=lambda x: int(x["Start_Timestamp"]))
        marks = [i for i, x in enumerate(rows) if "k_profile_mark" in x["Kernel_Name"]]
        if len(marks) < 2:
            return {"error": "marker kernels not found in the trace"}
        sel = rows[marks[-2] + 1:marks[-1]]
        mv = [x for x in sel if any(k in x["Kernel_Name"] for k in MATVEC_KERNELS)]
        dur = sum(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in mv) * 1e-9
        tot = sum(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in sel) * 1e-9
        span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) * 1e-9 if sel else 0.0
        keep = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(keep):                                   # leave the raw trace where profiles/ summaries are made from
            try:
                shutil.copy(files[0], os.path.join(keep, "bench_rocprof_kernel_trace.csv"))
            except Exception:
                pass
        return {"launches": len(mv), "launches_counted_by_plugin": info["launches"], "alg_bytes": info["alg_bytes"], "matvec_s": dur,
                "all_kernels_s": tot, "span_s": span, "n_dispatches": len(sel)}
    except Exception as e:
        return {"error": str(e)}
    finally:
        shutil.rmtree(out, ignore_errors=True)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks ourselves (one process per GPU through
    torch.distributed.run, rendezvous on 127.0.0.1 and a free port), BEFORE this process has made any GPU call, relay rank 0's JSON line
    and exit with the launcher's code.  Under the driver's own `python -m torch.distributed.run ... bench.py --gpus N` WORLD_SIZE is set and
    this function is not reached.  `--rank-script` (tests/test_bench_launcher.py) substitutes the per-rank program."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    script = os.path.abspath(args.rank_script) if args.rank_script else os.path.abspath(__file__)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("EH_FORCE_TP", None)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    line = None
    for ln in p.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line is None:
        sys.stderr.write(f"bench.py: the {args.gpus} ranks ended (exit {rc}) without a result line\n")
        return rc or 1
    res = json.loads(line)
    res["launcher"] = f"bench.py started {args.gpus} ranks itself (torch.distributed.run, 127.0.0.1:{port})"
    print(json.dumps(res), flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", default="vicuna-7b")
    ap.add_argument("--ftype", default="q4_k_m")
    ap.add_argument("--accept-p", type=float, default=0.8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rocprof", action="store_true", help="skip the rocprofv3 child run (roofline.achieved then comes from HIP events)")
    ap.add_argument("--no-extra", action="store_true", help="skip the accept_p sweep and the tree / Q8_0 workloads")
    ap.add_argument("--rocprof-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rank-script", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.rocprof_child or args.pmc_child:
        return rocprof_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        argv = [a for a in sys.argv[1:]]
        if args.rank_script:                                       # (the substitute gets the same flags minus its own name)
            i = argv.index("--rank-script"); del argv[i:i + 2]
        return spawn_ranks(args, argv)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1 or os.environ.get("EH_FORCE_TP"):      # EH_FORCE_TP=1: rehearse the TP path with a 1-rank communicator
        from bench_tp import main_tp           # one process per GPU, row-split tensor parallel over RCCL
        return main_tp(args, rank, world, local)

    # the rocprofv3 pass first, while this process has not touched the GPU yet: it is a child process of its own
    rp = None if args.no_rocprof else rocprof_roofline(args)
    pmc = (None, "skipped (--no-rocprof)") if args.no_rocprof else pmc_traffic(args)

    import torch                                # plumbing only: barrier-free at N=1, used for cuda.synchronize()
    ea = load_pkg()
    lib = plugin_lib(ea)
    be = ea.Backend.mi355x(0)                   # raises if the HIP plugin is missing: no CPU fallback on the product path
    tgt = ea.Model(be, args.config, args.ftype, n_ctx=2048, seed=42)
    dft = ea.Model(be, args.config, args.ftype, n_ctx=2048, eagle_of=tgt, seed=42, accept_p=args.accept_p)
    prompt = prompt_tokens(1234)

    # non-speculative decode on the same model: the 1x the ">= 2x" target refers to.  Same treatment as the speculative path:
    # device-side arg-max, and a warm pass before the timed one (first launches, weight re-layout, attribute set-up are not timed)
    ea.plain_generate(tgt, prompt, 16)
    plain_toks, pst = ea.plain_generate(tgt, prompt, 96)
    plain_tps = (pst["n_predict"] - 1) / pst["t_decode_us"] * 1e6
    prompt_ms = pst["t_prompt_us"] / 1e3

    sess = ea.SpecSession(tgt, dft, prompt)
    if args.warmup > 0:
        sess.rounds(args.warmup, n_draft=N_DRAFT)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks, st = sess.rounds(args.steps, n_draft=N_DRAFT)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    value = st["n_predict"] / dt

    # HIP events around every mat-vec launch, same K rounds again (on the stream the kernels are launched on)
    lib.ggml_backend_mi355x_profile_begin()
    sess.rounds(args.steps, n_draft=N_DRAFT)
    out = (C.c_double * 4)()
    n_launch = lib.ggml_backend_mi355x_profile_end(out)
    raw_ms, alg_bytes, pair_ms = out[0], out[1], out[2]
    sess.close()
    ev = {"launches": n_launch, "avg_launch_us_raw_event_pair": round(raw_ms * 1e3 / max(1, n_launch), 2), "empty_event_pair_us": round(pair_ms * 1e3, 2),
          "avg_launch_us_net_of_empty_pair": round(max(raw_ms - pair_ms * n_launch, 1e-9) * 1e3 / max(1, n_launch), 2)}
    bytes_per_launch = alg_bytes / max(1, n_launch)

    if rp and "matvec_s" in rp and rp["launches"] > 0:
        avg_us = rp["matvec_s"] / rp["launches"] * 1e6
        bpl = rp["alg_bytes"] / max(1, rp["launches_counted_by_plugin"])
        achieved = bpl / (avg_us * 1e-6) / 1e9
        src = "rocprofv3 --kernel-trace of the same %d rounds in a child process of this run (durations between the plugin's marker kernels)" % args.steps
        extra_rp = {"launches": rp["launches"], "launches_counted_by_plugin": rp["launches_counted_by_plugin"], "matvec_kernel_ms_per_round": round(rp["matvec_s"] * 1e3 / args.steps, 4),
                    "all_kernel_ms_per_round": round(rp["all_kernels_s"] * 1e3 / args.steps, 4), "wall_ms_per_round_under_rocprof": round(rp["span_s"] * 1e3 / args.steps, 4)}
    else:
        avg_us = ev["avg_launch_us_net_of_empty_pair"]; bpl = bytes_per_launch
        achieved = bpl / (avg_us * 1e-6) / 1e9 if n_launch else 0.0
        src = "HIP events net of an empty event pair (rocprofv3 child run unavailable: %s)" % (rp or {}).get("error", "skipped")
        extra_rp = None
    roofline = {"bound": "hbm", "kernel": "quantised mat-vec family: k_mmt (tiled weights, int8 MFMA, in-kernel Q8_K quantiser; 1..8 tokens)", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc[0], "traffic_source": pmc[1],
                "avg_launch_us": round(avg_us, 2), "algorithmic_bytes_per_launch": round(bpl), "source": src, "rocprof": extra_rp, "hip_events": ev}

    res = {"metric": "accepted tokens/sec + accept-rate, Vicuna-7B Q4_K_M + EAGLE, 1/8 GPU", "value": round(value, 2), "unit": "tokens/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "q4_K/q6_K x int8 -> int32 (MFMA) -> f32", "data": "synthetic",
           "config": {"workload": f"{args.config} {args.ftype} target + EAGLE head, chain/tree depth {N_DRAFT} (verify batch {N_DRAFT+1}), "
                                  f"{PROMPT_LEN}-token synthetic prompt, greedy", "n_draft": N_DRAFT, "accept_p_synthetic": args.accept_p},
           "accept_rate": round(st["n_accept"] / max(1.0, st["n_drafted"]), 4), "tokens_per_round": round(st["n_predict"] / args.steps, 3),
           "plain_decode_tokens_per_s": round(plain_tps, 2), "speedup_vs_plain": round(value / plain_tps, 3), "prompt_128_ms": round(prompt_ms, 2),
           "target_weight_bytes": tgt.weight_bytes, "draft_weight_bytes": dft.weight_bytes,
           "roofline": roofline}

    if not args.no_extra:
        extra = {}
        # tokens/s as a function of the synthetic acceptance dial (measured, not modelled)
        sweep = []
        for ap_ in (0.0, 0.5, 0.8, 1.0):
            d2 = ea.Model(be, args.config, args.ftype, n_ctx=2048, eagle_of=tgt, seed=42, accept_p=ap_)
            s2 = ea.SpecSession(tgt, d2, prompt)
            s2.rounds(2, n_draft=N_DRAFT)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            _, st2 = s2.rounds(20, n_draft=N_DRAFT)
            torch.cuda.synchronize(); d1 = time.perf_counter() - t1
            s2.close(); d2.close()
            sweep.append({"accept_p": ap_, "tokens_per_s": round(st2["n_predict"] / d1, 1), "tokens_per_round": round(st2["n_predict"] / 20, 3),
                          "accept_rate": round(st2["n_accept"] / max(1.0, st2["n_drafted"]), 4), "ms_per_round": round(d1 / 20 * 1e3, 3)})
        extra["accept_p_sweep"] = sweep
        # the reference's TREE driver (host/tree_driver.cpp): 4 branches forking on p_split, greedy verification
        try:
            # (the synthetic draft's best logit stands ~55 above the rest: a second candidate passes p_split only once the draft distribution
            #  is flattened -- temperature 13 gives a few forks per round without throwing the chain's acceptance away; draft-max 5 keeps the batch <= 9)
            ts = ea.TreeSession(tgt, dft, prompt, n_seq_dft=4, n_draft=5, p_split=0.02, temp=0.0, temp_dft=13.0, top_k=8)
            ts.run(8)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            tt, tst = ts.run(96)
            torch.cuda.synchronize(); d1 = time.perf_counter() - t1
            ts.close()
            extra["tree_driver"] = {"workload": "np 4, draft-max 5, p_split 0.02, draft temperature 13 (the synthetic draft is near one-hot: it forks only when flattened; on this synthetic pair a fork is pure cost -- the second candidate is never the target's token -- the same driver without forks, temp_dft 10, draft-max 7, runs 1400 tokens/s: profiles/r03_tree_probe.txt), top-k on the device, greedy verification", "tokens_per_s": round(len(tt) / d1, 1),
                                    "tokens_per_round": round(tst["n_predict"] / max(1.0, tst["n_iters"]), 3), "forks": int(tst["n_forks"]), "max_verify_batch": int(tst["max_batch"]),
                                    "verify_ms_per_round": round(tst["t_verify_us"] / max(1.0, tst["n_iters"]) / 1e3, 3), "draft_ms_per_round": round(tst["t_draft_us"] / max(1.0, tst["n_iters"]) / 1e3, 3)}
        except Exception as e:
            extra["tree_driver"] = {"error": str(e)}
        res["extra"] = extra
    dft.close(); tgt.close()

    if not args.no_extra:
        # BASELINE config 3: Q8_0 target, 10 branches, 60 drafts -> verification batches of > 60 tokens through the big-batch kernel (a3)
        try:
            t8 = ea.Model(be, args.config, "q8_0", n_ctx=2048, seed=42)
            d8 = ea.Model(be, args.config, "q8_0", n_ctx=2048, eagle_of=t8, seed=42, accept_p=args.accept_p)
            ea.plain_generate(t8, prompt, 8)
            _, p8 = ea.plain_generate(t8, prompt, 64)                       # the 1x of this configuration: plain greedy decode of the same Q8_0 target
            plain8 = (p8["n_predict"] - 1) / p8["t_decode_us"] * 1e6
            ts = ea.TreeSession(t8, d8, prompt, n_seq_dft=10, n_draft=60, p_split=0.01, temp=0.0, temp_dft=20.0, top_k=12)
            ts.run(8)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            tt, tst = ts.run(64)
            torch.cuda.synchronize(); d1 = time.perf_counter() - t1
            # roofline of the verification GEMM (k_bb, >= 25 tokens): HIP events around every launch of 64 more tokens, split by batch size
            lib.ggml_backend_mi355x_profile_end_by_batch.restype = C.c_int
            lib.ggml_backend_mi355x_profile_end_by_batch.argtypes = [C.c_int, C.POINTER(C.c_double)]
            lib.ggml_backend_mi355x_profile_begin()
            _, tst2 = ts.run(64)
            o9 = (C.c_double * 9)()
            lib.ggml_backend_mi355x_profile_end_by_batch(25, o9)
            ts.close()
            big_ms, big_bytes, big_ops, big_n, pair = o9[4], o9[5], o9[6], o9[7], o9[8]
            net_ms = max(big_ms - pair * big_n, 1e-9)
            rf3 = None
            if big_n > 0:
                gbs = big_bytes / (net_ms * 1e-3) / 1e9; tops = big_ops / (net_ms * 1e-3) / 1e12
                rf3 = {"bound": "hbm", "kernel": "k_bb: int8 GEMM on 32x32x32 MFMA tiles, verification batches of >= 25 tokens (HIP events net of an empty pair)", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "launches": int(big_n), "avg_launch_us": round(net_ms * 1e3 / big_n, 2),
                       "algorithmic_bytes_per_launch": round(big_bytes / big_n), "int8_tops": round(tops, 1), "int8_peak_tops": INT8_PEAK_TOPS, "mfma_frac": round(tops / INT8_PEAK_TOPS, 4),
                       "gemm_ms_per_round": round(net_ms / max(1.0, tst2["n_iters"]), 3)}
            res["extra"]["config3_q8_0_tree"] = {"workload": "vicuna-7b q8_0 + EAGLE head, np 10, draft-max 60 (BASELINE configs[2]: width 10, depth 6), draft temperature 20 so that the tree forks", "forks": int(tst["n_forks"]), "draft_decodes_per_round": round(tst["n_draft_calls"] / max(1.0, tst["n_iters"]), 2), "tokens_per_s": round(len(tt) / d1, 1),
                                                 "plain_q8_0_tokens_per_s": round(plain8, 1), "speedup_vs_plain_q8_0": round(len(tt) / d1 / plain8, 3),
                                                 "tokens_per_round": round(tst["n_predict"] / max(1.0, tst["n_iters"]), 3), "max_verify_batch": int(tst["max_batch"]),
                                                 "verify_ms_per_round": round(tst["t_verify_us"] / max(1.0, tst["n_iters"]) / 1e3, 3), "draft_ms_per_round": round(tst["t_draft_us"] / max(1.0, tst["n_iters"]) / 1e3, 3),
                                                 "target_weight_bytes": t8.weight_bytes, "roofline": rf3}
            d8.close(); t8.close()
        except Exception as e:
            res["extra"]["config3_q8_0_tree"] = {"error": str(e)}

    if not args.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline(ea, args.config, args.ftype)
        except Exception as e:                                       # the baseline must never take the GPU number down with it
            res["cpu_baseline"] = {"error": str(e)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    sys.exit(main() or 0)
