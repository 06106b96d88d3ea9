"""Multi-GPU plumbing: one process per GPU, torch.distributed for rendezvous.

 * RCCL over xGMI (the product path): rank 0 draws the 128-byte RCCL unique id, torch.distributed broadcasts it, and
   every rank hands it to the HIP library, which then issues the ncclSend/ncclRecv groups itself on its own stream.
 * GlooTransport (tests): the same exchange carried by torch.distributed P2P ops on host copies, so that several
   ranks can share ONE GPU (RCCL refuses two ranks per device) or run without any GPU (memcpy injected).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import host

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p,
                          C.POINTER(C.c_size_t), C.POINTER(C.c_size_t))


def init_rccl(op, group=None, device="cuda"):
    """collective over `group`: distribute rank 0's RCCL unique id and create the communicator inside the HIP library.

    Every step that can fail on one rank only is followed by an agreement, so that no rank is ever left inside a
    collective the others skipped: (1) rank 0 draws the id inside try/except and broadcasts the id OR a failure marker
    — all ranks return False together (the caller falls back) before any RCCL call; (2) ncclCommInitRank is entered by
    every rank; its outcome is MIN-reduced, and a failure on any rank after that GPU call raises on ALL ranks (fatal:
    no in-process retry)."""
    obj = [None]
    if dist.get_rank(group) == 0:
        try:
            obj[0] = host.rccl_unique_id()
        except Exception as e:  # noqa: BLE001 - reported through the marker
            obj[0] = "ERR:" + repr(e)
    dist.broadcast_object_list(obj, src=0, group=group)
    if not isinstance(obj[0], (bytes, bytearray)):
        return False
    err = None
    try:
        op.comm_init_rccl(obj[0])
    except Exception as e:  # noqa: BLE001
        err = e
    ok = torch.tensor([0 if err else 1], device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0:
        raise RuntimeError(f"RCCL communicator creation failed on at least one rank (this rank: {err!r})")
    return True


def _hip_runtime():
    """the HIP runtime this process already uses (torch bundles its own copy; loading a second one by name would give
    the staging copies a different runtime from the one that owns the buffers)"""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.split()[-1]
                if "libamdhip64.so" in path:
                    return C.CDLL(path)
    except OSError:
        pass
    return C.CDLL("libamdhip64.so")


def _hip_memcpy():
    hip = _hip_runtime()
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int

    def d2h(dst_host, src_dev, n):
        if hip.hipMemcpy(dst_host, src_dev, n, 2):
            raise RuntimeError("hipMemcpy D2H failed")

    def h2d(dst_dev, src_host, n):
        if hip.hipMemcpy(dst_dev, src_host, n, 1):
            raise RuntimeError("hipMemcpy H2D failed")
    return d2h, h2d


def host_memcpy():
    """for CPU-only tests: 'device' pointers are host pointers"""
    def cp(dst, src, n):
        C.memmove(dst, src, n)
    return cp, cp


def _hip_memcpy_d2d():
    hip = _hip_runtime()
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int

    def cp(dst, src, n):
        if hip.hipMemcpy(dst, src, n, 3):
            raise RuntimeError("hipMemcpy D2D failed")
    return cp, cp


class GlooTransport:
    """hm_exchange_fn implemented with torch.distributed P2P ops on host staging tensors."""

    staging_device = "cpu"

    def __init__(self, group=None, memcpy=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.d2h, self.h2d = memcpy if memcpy is not None else self._default_memcpy()
        self.calls = 0
        self.bytes_sent = 0
        self.cfunc = EXCHANGE_FN(self._exchange)

    @staticmethod
    def _default_memcpy():
        return _hip_memcpy()

    def _exchange(self, user, send_dev, send_off, send_bytes, recv_dev, recv_off, recv_bytes):
        try:
            ops, recvs = [], []
            for p in range(self.world):
                if p == self.rank:
                    continue
                nb = send_bytes[p]
                if nb:
                    t = torch.empty(nb, dtype=torch.uint8, device=self.staging_device)
                    self.d2h(t.data_ptr(), send_dev + send_off[p], nb)
                    ops.append(dist.P2POp(dist.isend, t, p, self.group))
                    self.bytes_sent += nb
                nb = recv_bytes[p]
                if nb:
                    r = torch.empty(nb, dtype=torch.uint8, device=self.staging_device)
                    ops.append(dist.P2POp(dist.irecv, r, p, self.group))
                    recvs.append((r, recv_off[p], nb))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                if self.staging_device != "cpu":
                    torch.cuda.synchronize()
            for r, off, nb in recvs:
                self.h2d(recv_dev + off, r.data_ptr(), nb)
            if self.staging_device != "cpu":
                # device-to-device staging copies run on the null stream, the backend's stream is non-blocking: make sure
                # they have landed before the unpack kernels read the buffer
                torch.cuda.synchronize()
            self.calls += 1
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("GlooTransport failed:", repr(e), flush=True)
            return 1


class TorchNcclTransport(GlooTransport):
    """the same exchange through torch.distributed's own NCCL (= RCCL) process group on device staging tensors: a
    fallback for multi-GPU runs when the HIP library's private communicator cannot be created.  Slower than
    hm_comm_init_rccl (host synchronisation per exchange), never used silently: bench.py labels it."""

    def __init__(self, group=None):
        self.staging_device = torch.device("cuda", torch.cuda.current_device())
        super().__init__(group, memcpy=_hip_memcpy_d2d())


class InProcessGroup:
    """G ranks as G threads of ONE process on one GPU (every rank its own HIP context / stream / HBM pool): the exchange is
    a rendezvous of the threads plus device-to-device copies out of each peer's send buffer.  This is how the 8-rank split
    of BASELINE configs[4] is executed on a one-GPU box (RCCL refuses two ranks per device, and a GPU box admits at most 6
    processes on its card, so 8 gloo processes are not an option).  `transport(rank)` gives rank `rank`'s hm_exchange_fn."""

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.posted = [None] * world
        hip = _hip_runtime()
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipMemcpy.restype = C.c_int
        hip.hipDeviceSynchronize.restype = C.c_int
        self.hip = hip
        self.calls = [0] * world
        self.bytes_recv = [0] * world
        self.failed = False
        self._fns = [self._make(r) for r in range(world)]

    def transport(self, rank):
        return self._fns[rank]

    def _make(self, me):
        def exchange(user, send_dev, send_off, send_bytes, recv_dev, recv_off, recv_bytes):
            try:
                W = self.world
                self.posted[me] = (send_dev or 0, [send_off[p] for p in range(W)], [send_bytes[p] for p in range(W)])
                self.barrier.wait(timeout=120)          # every rank's send buffer is complete (the caller synchronised its stream)
                for p in range(W):
                    if p == me:
                        continue
                    nb = recv_bytes[p]
                    pdev, poff, pbytes = self.posted[p]
                    # both sides must agree on every pair's size, zero included: over RCCL a send the peer does not expect (or
                    # a receive nobody feeds) is a hang, not an error
                    if pbytes[me] != nb:
                        raise RuntimeError(f"rank {me} expects {nb} bytes from rank {p}, which sends {pbytes[me]}")
                    if not nb:
                        continue
                    if self.hip.hipMemcpy(recv_dev + recv_off[p], pdev + poff[me], nb, 3):
                        raise RuntimeError("hipMemcpy D2D failed")
                    self.bytes_recv[me] += nb
                self.hip.hipDeviceSynchronize()
                self.barrier.wait(timeout=120)          # all copies have landed: send buffers may be reused
                self.calls[me] += 1
                return 0
            except Exception as e:  # noqa: BLE001 - never let an exception cross the C boundary
                self.failed = True
                try:
                    self.barrier.abort()
                except Exception:  # noqa: BLE001
                    pass
                print(f"InProcessGroup rank {me} failed:", repr(e), flush=True)
                return 1
        return EXCHANGE_FN(exchange)


def run_in_process(world, make_op, body):
    """run `body(rank, op)` on `world` threads; make_op(rank) builds rank's host.Op (world / rank overrides included).
    Returns the list of results; raises the first exception of any rank."""
    import threading
    grp = InProcessGroup(world)
    ops, res, err = [None] * world, [None] * world, [None] * world
    for r in range(world):
        ops[r] = make_op(r)

    def work(r):
        try:
            # hm_comm_init_* is a collective since round 6 (every rank tells every peer its replicate threshold over the new
            # communicator): each rank's thread makes its own
            ops[r].comm_init_external(grp.transport(r))
            grp.barrier.wait(timeout=120)
            if r == 0:   # the counters describe the op's own exchanges, not the communicator's first word
                grp.calls = [0] * world
                grp.bytes_recv = [0] * world
            grp.barrier.wait(timeout=120)
            res[r] = body(r, ops[r])
        except Exception as e:  # noqa: BLE001
            err[r] = e
            try:
                grp.barrier.abort()
            except Exception:  # noqa: BLE001
                pass
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return ops, res, grp
