"""Worker of tests/test_tp_gloo.py: one rank of a world_size-N tensor-parallel decode on the reference CPU backend,
all-reduce over gloo.  Usage: tp_worker.py RANK WORLD PORT OUTFILE"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_package
import refapi

def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ea = load_package()
    be = refapi.reference_cpu(ea, threads=2)
    m = ea.Model(be, "tiny-gqa", "q4_k_m", n_ctx=128, seed=9, predictable=False, tp_rank=rank, tp_size=world)
    def allreduce(ptr, n):
        buf = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (n,))
        t = torch.from_numpy(buf)
        dist.all_reduce(t)                      # in place on the backend's (host) memory
    m.set_allreduce(allreduce)
    res = {}
    lg, hid = m.decode(list(range(7, 19)), list(range(12)))            # prompt
    lg1, hid1 = m.decode([40, 41, 42], [12, 13, 13], seq=[0, 0, 0])    # small batch
    if rank == 0:
        np.savez(out, lg=lg, hid=hid, lg1=lg1, hid1=hid1, n_allreduce=m.n_allreduce, weight_bytes=m.weight_bytes)
    dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    main()
