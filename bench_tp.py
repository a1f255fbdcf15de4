"""bench_tp.py -- the N > 1 leg of bench.py: row-split tensor parallelism, one process per GPU, RCCL over xGMI.

    python bench.py --gpus N ...                       (bench.py starts the N ranks itself when WORLD_SIZE is not set)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Sharding (SURVEY.md 8e / DESIGN.md 6): every rank holds a slice of the SAME global synthetic Vicuna-7B tensors --
wq/wk/wv/gate/up by output rows (heads), wo/ffn_down by k in whole 256-element super-blocks -- keeps its own heads' KV,
and the layer needs two all-reduces of [n_embd, T] fp32, enqueued on the plugin's HIP stream from INSIDE graph_compute (the plugin's
node hooks call host/tp.cpp back after wo / ffn_down: a forward is one submission; EH_TP_SEGMENTS=1 restores the segment loop).  The EAGLE head and the LM head stay on rank 0: rank 0 drafts, the
draft tokens are broadcast (8 ints, gloo), all ranks verify together, rank 0 accepts and broadcasts the count.
Total work is fixed as N grows => "scaling": "strong".  The same code path runs on CPU with gloo in
tests/test_tp_gloo.py (world_size 2) and, with EH_FORCE_TP=1, on one GPU with a 1-rank RCCL communicator.
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
N_DRAFT = 5


class GpuRanks:
    """What a rank needs from its platform: one MI355X per process, the plugin as backend, RCCL for the data path.
    (tests/bench_rank_cpu.py has the CPU twin -- reference CPU backend + gloo -- with which the launcher and this file's round
    protocol are rehearsed without a GPU; nothing in this file or in bench.py knows about it.)"""
    name = "MI355X, RCCL over xGMI"
    has_profile = True

    def init(self, rank, world, local):
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        return dist.new_group(backend="gloo")               # small host-side control messages (token ids)

    def backend(self, ea, local):
        return ea.Backend.mi355x(local)

    def bind_allreduce(self, ea, be, model, rank, world, ctl):
        """RCCL communicator on the plugin's own HIP stream (host/tp.cpp); returns the communicator's size as RCCL reports it."""
        import torch
        import torch.distributed as dist
        h = ea._model_sigs()
        idbuf = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            raw = C.create_string_buffer(128)
            assert h.eh_tp_unique_id(raw) == 0, "librccl not loadable"
            idbuf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
        dist.broadcast(idbuf, 0, group=ctl)
        comm = h.eh_tp_init(be.h, bytes(idbuf.numpy().tobytes()), rank, world)
        assert comm, "ncclCommInitRank failed"
        h.eh_tp_bind(comm, model.h)
        return int(h.eh_tp_comm_size(comm))

    def device_sync(self):
        import torch
        torch.cuda.synchronize()


def main_tp(args, rank, world, local, platform=None):
    import numpy as np
    import torch
    import torch.distributed as dist
    from bench import load_pkg, prompt_tokens, PROMPT_LEN

    plat = platform or GpuRanks()
    ctl = plat.init(rank, world, local)
    ea = load_pkg()
    h = ea._model_sigs()
    be = plat.backend(ea, local)
    n_ctx = 2048 if args.config != "tiny-gqa" else 512

    tgt = ea.Model(be, args.config, args.ftype, n_ctx=n_ctx, seed=42, tp_rank=rank, tp_size=world)
    comm_size = plat.bind_allreduce(ea, be, tgt, rank, world, ctl)
    dft = ea.Model(be, args.config, args.ftype, n_ctx=n_ctx, eagle_of=tgt, seed=42, accept_p=args.accept_p) if rank == 0 else None
    prompt = prompt_tokens(1234, vocab=tgt.n_vocab if hasattr(tgt, "n_vocab") else 32000)

    def sync():
        plat.device_sync()
        dist.barrier()

    # ---- prompt (all ranks decode it; rank 0 also primes the draft) -------------------------------------------------
    sess = None
    if rank == 0:
        pr = (C.c_int32 * len(prompt))(*prompt)
    # followers mirror rank 0's target calls one for one: prompt decode with all outputs
    if rank == 0:
        sess = h.eh_spec_begin(tgt.h, dft.h, pr, len(prompt))
        assert sess
    else:
        tgt.kv_clear()
        tgt.decode(prompt, list(range(len(prompt))), want_hidden=True)
    n_past = len(prompt)
    st = (C.c_double * 16)()
    out = (C.c_int32 * (N_DRAFT + 2))()
    drafts = (C.c_int32 * (N_DRAFT + 2))()
    msg = torch.zeros(N_DRAFT + 4, dtype=torch.int32)

    def one_round():
        nonlocal n_past
        if rank == 0:
            nd = h.eh_spec_draft(sess, N_DRAFT, 0.0, drafts, st)
            npast = C.c_int32(); idl = C.c_int32(); h.eh_spec_state(sess, C.byref(npast), C.byref(idl))
            msg[0] = nd; msg[1] = idl.value; msg[2] = npast.value
            for i in range(nd):
                msg[3 + i] = drafts[i]
        dist.broadcast(msg, 0, group=ctl)
        nd, idl, npast = int(msg[0]), int(msg[1]), int(msg[2])
        if rank == 0:
            n_out = h.eh_spec_verify(sess, out, st)
            msg[0] = n_out
        else:
            toks = [idl] + [int(msg[3 + i]) for i in range(nd)]
            tgt.decode(toks, [npast + i for i in range(nd + 1)], want_hidden=True)
        dist.broadcast(msg[:1], 0, group=ctl)
        n_out = int(msg[0])
        n_past = npast + n_out                      # id_last + accepted drafts stay in the cache
        if rank != 0:
            tgt.kv_seq_rm(0, n_past, -1)
        return n_out

    for _ in range(args.warmup):
        one_round()
    for i in range(16):
        st[i] = 0.0
    sync()
    t0 = time.perf_counter()
    n_tok = 0
    for _ in range(args.steps):
        n_tok += one_round()
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=ctl)
    dt = float(tmax[0])
    # roofline of rank 0's mat-vec launches (its shard of the weights): the same K rounds again, every rank takes part in the
    # collectives, rank 0 brackets its launches with HIP events (plugin profile hooks)
    n_acc, n_dr = st[2], st[1]
    from bench import plugin_lib, HBM_PEAK_GBS
    prof = plugin_lib(ea) if plat.has_profile else None
    if rank == 0 and prof: prof.ggml_backend_mi355x_profile_begin()
    if prof:
        for _ in range(args.steps):
            one_round()
    roofline = None
    if rank == 0 and prof:
        out = (C.c_double * 4)()
        n_launch = prof.ggml_backend_mi355x_profile_end(out)
        raw_ms, alg_bytes, pair_ms = out[0], out[1], out[2]
        kern_ms = max(raw_ms - pair_ms * n_launch, 1e-9)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if n_launch > 0 else 0.0
        roofline = {"bound": "hbm", "kernel": "quantised mat-vec family on rank 0 (its weight shard): k_mmt (HIP events net of an empty pair)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "launches": n_launch,
                    "avg_launch_us": round(kern_ms * 1e3 / max(1, n_launch), 2), "empty_event_pair_us": round(pair_ms * 1e3, 2),
                    "algorithmic_bytes_per_launch": round(alg_bytes / max(1, n_launch))}
    sync()
    if rank == 0:
        res = {"metric": "accepted tokens/sec + accept-rate, Vicuna-7B Q4_K_M + EAGLE, 1/8 GPU", "value": round(n_tok / dt, 2), "unit": "tokens/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "q4_K/q6_K x int8 -> int32 (MFMA) -> f32", "data": "synthetic",
               "config": {"workload": f"{args.config} {args.ftype} target row-split TP={world} (2 RCCL all-reduces/layer) + EAGLE head on rank 0, depth {N_DRAFT}, "
                                      f"{PROMPT_LEN}-token synthetic prompt, greedy", "n_draft": N_DRAFT, "accept_p_synthetic": args.accept_p},
               "accept_rate": round(n_acc / max(1.0, n_dr), 4), "tokens_per_round": round(n_tok / args.steps, 3),
               "ranks": world, "communicator_size": comm_size, "platform": plat.name,
               "allreduce_submission": "enqueued from inside graph_compute (plugin node hooks): one submission per forward" if os.environ.get("EH_TP_SEGMENTS") is None and plat.has_profile else "host-issued between graph segments",
               "shard_weight_bytes": tgt.weight_bytes, "allreduces": tgt.n_allreduce, "roofline": roofline,
               "cpu_baseline": None}      # the CPU baseline is timed by the N = 1 run only (bench.py)
        print(json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()
