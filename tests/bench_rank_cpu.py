"""CPU twin of a bench.py rank (test infrastructure): the per-rank program tests/test_bench_launcher.py hands to bench.py's launcher
(`bench.py --gpus 2 --rank-script tests/bench_rank_cpu.py ...`).  It runs bench_tp.main_tp -- the very round protocol of the N > 1
bench: rank 0 drafts, draft tokens broadcast, all ranks verify their shard, two all-reduces per layer -- on the reference CPU backend
(oracle/_ref) with gloo as the collective, on the tiny config.  Nothing outside tests/ imports this file."""
import argparse
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)


class CpuRanks:
    name = "CPU rehearsal: reference ggml CPU backend + gloo"
    has_profile = False

    def init(self, rank, world, local):
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=rank, world_size=world)
        return dist.new_group(backend="gloo")

    def backend(self, ea, local):
        import refapi
        return refapi.reference_cpu(ea, threads=2)

    def bind_allreduce(self, ea, be, model, rank, world, ctl):
        import numpy as np
        import torch
        import torch.distributed as dist

        def allreduce(ptr, n):
            buf = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (n,))
            dist.all_reduce(torch.from_numpy(buf))                 # in place on the backend's (host) memory
        model.set_allreduce(allreduce)
        return dist.get_world_size()

    def device_sync(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="tiny-gqa")
    ap.add_argument("--ftype", default="q4_k_m")
    ap.add_argument("--accept-p", type=float, default=0.8)
    args, _ = ap.parse_known_args()
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"]); local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, (world, args.gpus)
    from bench_tp import main_tp
    main_tp(args, rank, world, local, platform=CpuRanks())


if __name__ == "__main__":
    main()
