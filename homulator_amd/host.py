"""ctypes binding of host/include/homulator_host.h (libhomulator_host.so): build one FHE operation with the
C++ Operation / InsGen / Driver layer and run it on the HIP backend (or the no-GPU `count` backend that only
accounts instructions)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("HOMULATOR_HOST_LIB") or os.path.join(ROOT, "host", "lib", "libhomulator_host.so")  # override: A/B builds
CONFIG_DIR = os.path.join(ROOT, "config")
BACKEND_HIP, BACKEND_COUNT, BACKEND_SIM = 0, 1, 2
SEED = 0x484F4D55

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build() (make -C host)")
        if os.environ.get("HOMULATOR_HIP_LIB"):
            # an A/B build of the backend: load it first — the host library's NEEDED entry (soname libhomulator_hip.so) then binds
            # to it instead of the in-tree build on its runpath
            from . import hip
            hip.load()
        L = C.CDLL(LIB_PATH)
        vp, u32, u64p = C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)
        L.hh_op_create.argtypes = [C.POINTER(vp), C.c_char_p, C.c_char_p, u32, u32, u32, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.hh_op_destroy.argtypes = [vp]
        L.hh_last_error.restype = C.c_char_p
        L.hh_op_simulate.argtypes = [vp]
        L.hh_op_execute.argtypes = [vp, u32, C.POINTER(C.c_double)]
        L.hh_op_write_buffer.argtypes = [vp, C.c_char_p, u32, vp]
        L.hh_op_sim_run.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.hh_op_sim_stats.argtypes = [vp, C.c_char_p, u32]
        L.hh_op_enqueue.argtypes = [vp, u32]
        L.hh_op_sync.argtypes = [vp]
        L.hh_op_total_instructions.argtypes = [vp, u64p]
        L.hh_op_launch_count.argtypes = [vp, u64p]
        L.hh_op_stage_bytes.argtypes = [vp, u64p]
        L.hh_op_buffer_limbs.argtypes = [vp, C.c_char_p, C.POINTER(u32)]
        L.hh_op_read_buffer.argtypes = [vp, C.c_char_p, vp]
        L.hh_op_read_buffer_copy.argtypes = [vp, C.c_char_p, u32, vp]
        L.hh_op_batch.restype = u32
        L.hh_op_batch.argtypes = [vp]
        L.hh_op_buffer_names.argtypes = [vp, C.c_char_p, u32]
        L.hh_op_N.restype = u32
        L.hh_op_N.argtypes = [vp]
        L.hh_op_plan.argtypes = [vp, C.c_char_p, u32]
        L.hh_op_stage_times.argtypes = [vp, u32, C.c_char_p, u32]
        L.hh_op_backend_counter.argtypes = [vp, C.c_char_p, u64p]
        L.hh_op_bind_input.argtypes = [vp, C.c_char_p, vp]
        L.hh_chain_create.argtypes = [C.POINTER(vp), C.c_char_p, C.c_char_p, u32, u32, u32, C.c_char_p, C.c_int]
        L.hh_chain_destroy.argtypes = [vp]
        L.hh_chain_size.restype = u32
        L.hh_chain_size.argtypes = [vp]
        L.hh_chain_op.restype = vp
        L.hh_chain_op.argtypes = [vp, u32]
        L.hh_chain_execute.argtypes = [vp, u32, C.POINTER(C.c_double)]
        L.hh_chain_enqueue.argtypes = [vp, u32]
        L.hh_chain_sync.argtypes = [vp]
        L.hh_op_refill.argtypes = [vp, C.c_char_p, C.c_uint64]
        L.hh_op_snapshot.argtypes = [vp, C.c_char_p, u32]
        L.hh_op_snapshot_read.argtypes = [vp, u32, vp]
        L.hh_chain_simulate.argtypes = [vp]
        L.hh_comm_unique_id.argtypes = [vp]
        L.hh_op_comm_init_rccl.argtypes = [vp, vp]
        L.hh_op_comm_init_external.argtypes = [vp, vp, vp]
        _lib = L
    return _lib


class HostError(RuntimeError):
    pass


class Op:
    """One operation instance: `Op("config_4.cfg", "hmult", 45, 35, 15)` = `./Homulator.run config_4.cfg hmult 45 35 15`."""

    def __init__(self, cfg, op, max_level, cur_level, alpha, backend=BACKEND_HIP, fuse=True, device=0, overrides=None, quiet=True,
                 rank=0, world=1):
        self.L = load()
        self.rank, self.world, self.cur_level = rank, world, cur_level
        if world > 1:
            overrides = dict(overrides or {}, world=world, rank=rank)
        if not os.path.isabs(cfg) and not os.path.exists(cfg):
            cfg = os.path.join(CONFIG_DIR, cfg)
        self.h = C.c_void_p()
        ov = None if not overrides else ";".join(f"{k}={v}" for k, v in overrides.items()).encode()
        st = self.L.hh_op_create(C.byref(self.h), cfg.encode(), op.encode(), max_level, cur_level, alpha, backend, 1 if fuse else 0,
                                 device, ov, 1 if quiet else 0)
        if st:
            raise HostError(self.L.hh_last_error().decode())
        self.N = self.L.hh_op_N(self.h)

    def _ck(self, st):
        if st:
            raise HostError(self.L.hh_last_error().decode())

    def close(self):
        if self.h:
            if not getattr(self, "borrowed", False):
                self.L.hh_op_destroy(self.h)
            self.h = None

    def refill(self, input_name, seed):
        """asynchronous: new synthetic data for input ciphertext `input_name` ("ct1" / "ct2"), stream-ordered"""
        self._ck(self.L.hh_op_refill(self.h, input_name.encode(), C.c_uint64(int(seed))))

    def snapshot(self, name, slot=0):
        """asynchronous device-side copy of a named buffer as it is at this point of the op's stream"""
        self._ck(self.L.hh_op_snapshot(self.h, name.encode(), slot))

    def snapshot_read(self, name, slot=0):
        n = C.c_uint32()
        self._ck(self.L.hh_op_buffer_limbs(self.h, name.encode(), C.byref(n)))
        out = np.empty((n.value, self.N), dtype=np.uint64)
        self._ck(self.L.hh_op_snapshot_read(self.h, slot, out.ctypes.data_as(C.c_void_p)))
        return out

    def bind_input(self, input_name, producer):
        """continuous execution: this op's input ciphertext (\"ct1\" / \"ct2\") is `producer`'s output, kept in HBM"""
        self._ck(self.L.hh_op_bind_input(self.h, input_name.encode(), producer.h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def simulate(self):
        self._ck(self.L.hh_op_simulate(self.h))

    def write(self, name, data, copy=0):
        """upload real data ([n_limbs][N] uint64, fully reduced, evaluation form) into a named input / key buffer"""
        n = C.c_uint32()
        self._ck(self.L.hh_op_buffer_limbs(self.h, name.encode(), C.byref(n)))
        a = np.ascontiguousarray(data, dtype=np.uint64)
        if a.shape != (n.value, self.N):
            raise HostError(f"{name}: expected shape {(n.value, self.N)}, got {a.shape}")
        self._ck(self.L.hh_op_write_buffer(self.h, name.encode(), copy, a.ctypes.data_as(C.c_void_p)))

    def sim_run(self):
        """backend = BACKEND_SIM: run the cycle model of the reference accelerator to completion.
        Returns {"cycles", "retired", "drained", "stats"}: the reference's cycle count and stat block for this op."""
        cyc, ret, ok = C.c_uint64(), C.c_uint64(), C.c_int()
        self._ck(self.L.hh_op_sim_run(self.h, C.byref(cyc), C.byref(ret), C.byref(ok)))
        buf = C.create_string_buffer(1 << 16)
        self._ck(self.L.hh_op_sim_stats(self.h, buf, len(buf)))
        stats = {ln.rsplit(" ", 1)[0]: int(ln.rsplit(" ", 1)[1]) for ln in buf.value.decode().splitlines()}
        return {"cycles": cyc.value, "retired": ret.value, "drained": bool(ok.value), "stats": stats}

    def execute(self, iters=1):
        """runs the whole op `iters` times; returns device ns per iteration"""
        ns = C.c_double()
        self._ck(self.L.hh_op_execute(self.h, iters, C.byref(ns)))
        return ns.value

    def enqueue(self, iters=1):
        self._ck(self.L.hh_op_enqueue(self.h, iters))

    def sync(self):
        self._ck(self.L.hh_op_sync(self.h))

    def _u64(self, fn):
        v = C.c_uint64()
        self._ck(fn(self.h, C.byref(v)))
        return v.value

    def total_instructions(self):
        return self._u64(self.L.hh_op_total_instructions)

    def launch_count(self):
        return self._u64(self.L.hh_op_launch_count)

    def stage_bytes(self):
        return self._u64(self.L.hh_op_stage_bytes)

    def buffer_names(self):
        buf = C.create_string_buffer(1 << 16)
        self._ck(self.L.hh_op_buffer_names(self.h, buf, len(buf)))
        return [s for s in buf.value.decode().split("\n") if s]

    def plan(self):
        buf = C.create_string_buffer(1 << 20)
        self._ck(self.L.hh_op_plan(self.h, buf, len(buf)))
        return [s for s in buf.value.decode().split("\n") if s]

    def backend_counter(self, name):
        """hm_get_counter of the op's HIP context (synchronises): e.g. "ntt_cross_xcd" = limb-polys of one-launch transforms whose
        workgroups were spread over XCDs (slow path), "ntt_fused_small" = the one-launch threshold in force (0 = off)"""
        return self._u64(lambda h, p: self.L.hh_op_backend_counter(h, name.encode(), p))

    def stage_times(self, iters=5):
        """[(kind, stage names, ns)] per launch of the plan, each launch bracketed by its own HIP event pair (collective
        when sharded: every rank calls it with the same iters)"""
        buf = C.create_string_buffer(1 << 20)
        self._ck(self.L.hh_op_stage_times(self.h, iters, buf, len(buf)))
        rows = []
        for s in buf.value.decode().split("\n"):
            if s:
                kind, rest = s.split(" ", 1)
                name, ns = rest.rsplit(" ", 1)
                rows.append((kind, name, int(ns)))
        return rows

    # ---- multi-GPU transports (one of them before the first execute when world > 1)
    def comm_init_rccl(self, unique_id):
        """unique_id: the 128 bytes of homulator_amd.host.rccl_unique_id() from rank 0, identical on every rank"""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._ck(self.L.hh_op_comm_init_rccl(self.h, buf))

    def comm_init_external(self, exchange_cfunc):
        self._keep_cb = exchange_cfunc
        self._ck(self.L.hh_op_comm_init_external(self.h, C.cast(exchange_cfunc, C.c_void_p), None))

    def owned(self, n_limbs):
        """limb indices of an n-limb buffer that live on this rank (limb e -> e % world)"""
        return [e for e in range(n_limbs) if e % self.world == self.rank]

    def read(self, name, copy=0):
        """limbs of buffer `name` of op `copy` of the batch (overrides={"batch": B}: every launch carries B independent ops)"""
        n = C.c_uint32()
        self._ck(self.L.hh_op_buffer_limbs(self.h, name.encode(), C.byref(n)))
        out = np.empty((n.value, self.N), dtype=np.uint64)
        self._ck(self.L.hh_op_read_buffer_copy(self.h, name.encode(), copy, out.ctypes.data_as(C.c_void_p)))
        return out

    @property
    def batch(self):
        return self.L.hh_op_batch(self.h)


class Chain:
    """Continuous execution: `Chain("config_4.cfg", "hmult,hrotate,hadd", 45, 35, 15)` runs the ops back to back with
    the ciphertext resident in HBM (op k+1's ct1 = op k's output; levels follow the data).  `chain[i]` is a borrowed Op
    view for `read(...)`."""

    def __init__(self, cfg, ops, max_level, cur_level, alpha, overrides=None, quiet=True):
        self.L = load()
        if not os.path.isabs(cfg) and not os.path.exists(cfg):
            cfg = os.path.join(CONFIG_DIR, cfg)
        self.h = C.c_void_p()
        ov = None if not overrides else ";".join(f"{k}={v}" for k, v in overrides.items()).encode()
        if self.L.hh_chain_create(C.byref(self.h), cfg.encode(), ops.encode(), max_level, cur_level, alpha, ov, 1 if quiet else 0):
            raise HostError(self.L.hh_last_error().decode())
        self.views = []
        for i in range(self.L.hh_chain_size(self.h)):
            v = Op.__new__(Op)
            v.L, v.h, v.borrowed = self.L, C.c_void_p(self.L.hh_chain_op(self.h, i)), True
            v.N = self.L.hh_op_N(v.h)
            self.views.append(v)

    def __len__(self):
        return len(self.views)

    def __getitem__(self, i):
        return self.views[i]

    def execute(self, iters=1):
        """runs the whole chain `iters` times; returns wall ns per pass"""
        ns = C.c_double()
        if self.L.hh_chain_execute(self.h, iters, C.byref(ns)):
            raise HostError(self.L.hh_last_error().decode())
        return ns.value

    def enqueue(self, iters=1):
        if self.L.hh_chain_enqueue(self.h, iters):
            raise HostError(self.L.hh_last_error().decode())

    def sync(self):
        if self.L.hh_chain_sync(self.h):
            raise HostError(self.L.hh_last_error().decode())

    def close(self):
        if self.h:
            for v in self.views:
                v.h = None
            self.L.hh_chain_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_unique_id():
    buf = C.create_string_buffer(128)
    if load().hh_comm_unique_id(buf):
        raise HostError(load().hh_last_error().decode())
    return buf.raw
