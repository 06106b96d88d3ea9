"""One rank of a sharded run (launched by tests/test_gpu_sharded.py and tests/test_dist_cpu.py with RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT in the environment).  mode `gpu`: run a limb-sharded op on the (shared) GPU with the gloo
transport and check it against the oracle on rank 0.  mode `transport`: CPU-only check of GlooTransport."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def gather_buffer(op, name, n_limbs, N, copy=0):
    """every rank contributes the limbs it owns; rank 0 returns the assembled [n_limbs][N] array"""
    mine = op.read(name, copy=copy)
    full = torch.zeros((n_limbs, N), dtype=torch.int64)
    own = op.owned(n_limbs)
    full[own] = torch.from_numpy(mine[own].view(np.int64))
    dist.reduce(full, dst=0, op=dist.ReduceOp.SUM)   # disjoint supports: the sum is the assembly
    return full.numpy().view(np.uint64)


def run_gpu(cfg, opname, L, ell, alpha, logN, batch=1):
    from homulator_amd import host
    from homulator_amd.dist import GlooTransport
    rank, world = dist.get_rank(), dist.get_world_size()
    op = host.Op(cfg, opname, L, ell, alpha, rank=rank, world=world, overrides={"batch": batch} if batch > 1 else None)
    tr = GlooTransport()
    op.comm_init_external(tr.cfunc)
    op.execute(1)
    op.execute(1)  # the plan must be re-runnable
    N = 1 << logN
    n_out = ell - 1 if opname == "hmult" else ell
    outs = [(gather_buffer(op, "out.c0", n_out, N, c), gather_buffer(op, "out.c1", n_out, N, c)) for c in range(batch)]
    ok = True
    if rank == 0:
        from oracle.homoracle import Oracle
        o = Oracle(logN, L, alpha)
        o.set_threads(4)
        evk = o.synth_evk(ell, host.SEED + 10000)   # one evaluation key for the whole batch
        for c in range(batch):
            S = host.SEED + c * 100000              # host/src/Arch.cpp kBatchSeedStride
            ct1, ct2 = o.synth_ct(ell, S), o.synth_ct(ell, S + 2000)
            exp = o.hmult(ell, ct1, ct2, evk) if opname == "hmult" else o.hrotate(ell, ct1, 5, evk)
            ok = ok and bool(np.array_equal(outs[c][0], exp[0]) and np.array_equal(outs[c][1], exp[1]))
        print(f"sharded {opname} world={world} {cfg} L={L} l={ell} alpha={alpha}: {'OK' if ok else 'MISMATCH'}; "
              f"exchanges={tr.calls} bytes_sent_rank0={tr.bytes_sent}", flush=True)
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, src=0)
    op.close()
    return int(flag.item()) == 1


def run_transport():
    """GlooTransport on host memory: every rank sends a distinct pattern to every peer at distinct offsets"""
    from homulator_amd.dist import GlooTransport, host_memcpy
    import ctypes as C
    rank, world = dist.get_rank(), dist.get_world_size()
    tr = GlooTransport(memcpy=host_memcpy())
    words = 1024
    send = np.zeros(world * words, dtype=np.uint64)
    recv = np.zeros(world * words, dtype=np.uint64)
    for p in range(world):
        send[p * words:(p + 1) * words] = np.arange(words, dtype=np.uint64) + (rank << 32) + (p << 48)
    off = (C.c_size_t * world)(*[p * words * 8 for p in range(world)])
    nb = (C.c_size_t * world)(*[0 if p == rank else (words - p) * 8 for p in range(world)])       # ragged sizes
    rb = (C.c_size_t * world)(*[0 if p == rank else (words - rank) * 8 for p in range(world)])
    rc = tr._exchange(None, send.ctypes.data, off, nb, recv.ctypes.data, off, rb)
    ok = rc == 0
    for p in range(world):
        got = recv[p * words:(p + 1) * words]
        if p == rank:
            ok &= not got.any()
        else:
            exp = np.arange(words - rank, dtype=np.uint64) + (p << 32) + (rank << 48)
            ok &= bool(np.array_equal(got[:words - rank], exp)) and not got[words - rank:].any()
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1


def main():
    mode = sys.argv[1]
    dist.init_process_group("gloo")
    if mode == "gpu":
        cfg, opname, L, ell, alpha, logN = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
        ok = run_gpu(cfg, opname, L, ell, alpha, logN, int(sys.argv[8]) if len(sys.argv) > 8 else 1)
    else:
        ok = run_transport()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
