"""The backend's RCCL code path with MORE THAN ONE RANK, on the one GPU of the test box: hm_comm_unique_id, the collective
hm_comm_init_rccl and the grouped ncclSend / ncclRecv of every exchange (homulator_amd/csrc/hm_backend.hip: all_to_all) run against
a test double of the eight RCCL entry points (tests/mock_rccl/mock_rccl.cpp, selected by HOMULATOR_RCCL_LIB).  The ranks are
threads, every rank has its own context / stream / pool; the double matches every send with the peer's receive, checks counts,
datatype sizes, peers and streams, pair by pair (no global barrier: ranks only meet the peers they exchange with, an empty group is
a no-op), and turns what would hang on a node (one-sided send or receive) or corrupt data (size disagreement) into an error.  BASELINE configs[4] — hmult 45/35/15 over 8 ranks — must come out bit-exact.
Runs in a child process: the library choice is made when the backend first loads RCCL."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import ctypes, sys, threading
sys.path.insert(0, %(root)r)
import numpy as np
from homulator_amd import host
from oracle.homoracle import Oracle
world, cfg, opname, L, ell, alpha, logN, batch, plan = %(case)r
uid = host.rccl_unique_id()                      # ncclGetUniqueId of the double (also loads it once, before the threads)
ops = [host.Op(cfg, opname, L, ell, alpha, rank=r, world=world, overrides={"batch": batch, "shard_plan": plan}) for r in range(world)]
err = [None] * world
def work(r):
    try:
        ops[r].comm_init_rccl(uid)               # collective: returns when all ranks have arrived
        ops[r].execute(1)
        ops[r].execute(1)
    except Exception as e:
        err[r] = e
th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join() for t in th]
for e in err:
    if e is not None: raise e
N = 1 << logN
n_out = ell - 1 if opname == "hmult" else ell
o = Oracle(logN, L, alpha); o.set_threads(8)
evk = o.synth_evk(ell, host.SEED + 10000)
def assemble(name, c):
    full = np.zeros((n_out, N), dtype=np.uint64); seen = np.zeros(n_out, dtype=int)
    for op in ops:
        mine = op.read(name, copy=c)
        for l in op.owned(n_out): full[l] = mine[l]; seen[l] += 1
    assert (seen == 1).all()
    return full
for c in range(batch):
    S = host.SEED + c * 100000
    ct1, ct2 = o.synth_ct(ell, S), o.synth_ct(ell, S + 2000)
    exp = o.hmult(ell, ct1, ct2, evk) if opname == "hmult" else o.hrotate(ell, ct1, 5, evk)
    assert np.array_equal(assemble("out.c0", c), exp[0]) and np.array_equal(assemble("out.c1", c), exp[1])
lib = ctypes.CDLL(%(lib)r)
g, b, e = ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
lib.mock_rccl_stats(ctypes.byref(g), ctypes.byref(b), ctypes.byref(e))
print("RCCL double: ranks", world, "groups", g.value, "bytes", b.value, "errors", e.value)
assert e.value == 0 and g.value > 0 and b.value > 0
"""


@pytest.mark.parametrize("case", [
    (8, "config_4.cfg", "hmult", 45, 35, 15, 16, 1, 0),      # BASELINE configs[4]: all-to-all plan (the choice above 4 ranks)
    (8, "config_4.cfg", "hrotate", 45, 35, 15, 16, 1, 0),
    (4, "config_4_N15.cfg", "hmult", 16, 10, 4, 15, 3, 1),   # batched
    (2, "config_4_N15.cfg", "hmult", 6, 5, 2, 15, 1, 1),
    (8, "config_4.cfg", "hmult", 45, 35, 15, 16, 1, 2),      # gather plan: 3 send/receive groups per hmult (every owner to every peer)
    (4, "config_4_N15.cfg", "hmult", 16, 10, 4, 15, 3, 0),   # ... where it is the automatic choice, batched
    (2, "config_4.cfg", "hrotate", 45, 35, 15, 16, 1, 0),
    (8, "config_4.cfg", "hmult", 45, 35, 15, 16, 2, 0),      # two ops' residues from one owner: scatter + exchange of chunks (one group more)
], ids=lambda c: f"{c[0]}ranks-{c[2]}-{c[3]}-{c[4]}-{c[5]}-b{c[7]}-plan{c[8]}")
def test_rccl_code_path_with_many_ranks(case):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "mock_rccl")], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "tests", "mock_rccl", "libmockrccl.so")
    env = dict(os.environ, HOMULATOR_RCCL_LIB=lib)
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "case": case, "lib": lib}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "errors 0" in r.stdout, r.stdout
    world = case[0]
    if world == 8 and case[2] == "hmult" and case[8] == 2:
        assert "groups 56 " in r.stdout, r.stdout     # (2 runs x 3 gathers + the threshold exchange of hm_comm_init_rccl, round 6) x 8 ranks
    if world == 8 and case[2] == "hmult" and case[8] == 0 and case[7] == 2:
        assert "groups 168 " in r.stdout, r.stdout    # (2 runs x (8 all-to-alls + the replicate in two phases) + the threshold exchange at init) x 8 ranks
    if world == 8 and case[2] == "hmult" and case[8] == 0 and case[7] == 1:
        # per-digit pipelined exchanges (default when sharded): 2 runs x (2 beta + 2 = 8 all-to-alls + 1 replicate) x 8 ranks = 144
        # groups; every rank enters every one, also the ranks that own nothing of a list; + 8: the one word every rank tells every peer when
        # the communicator is made (hm_comm_init_rccl verifies that the replicate threshold agrees: round 6)
        assert "groups 152 " in r.stdout, r.stdout


def test_the_double_reports_what_would_hang():
    """a receive without a matching send is an error of the double (it would be a hang over RCCL): checked directly on its API"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "mock_rccl")], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "tests", "mock_rccl", "libmockrccl.so")
    script = r'''
import ctypes, threading, sys
L = ctypes.CDLL(%r)
hip = ctypes.CDLL("libamdhip64.so")
class Id(ctypes.Structure): _fields_ = [("internal", ctypes.c_char * 128)]
uid = Id(); assert L.ncclGetUniqueId(ctypes.byref(uid)) == 0
L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, Id, ctypes.c_int]
L.ncclRecv.argtypes = L.ncclSend.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
L.ncclGetErrorString.restype = ctypes.c_char_p
res = [None, None]
def rank(r):
    comm = ctypes.c_void_p(); assert L.ncclCommInitRank(ctypes.byref(comm), 2, uid, r) == 0
    buf = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(buf), 4096)
    L.ncclGroupStart()
    if r == 0: L.ncclRecv(buf, 16, 5, 1, comm, None)      # rank 1 sends nothing
    rc = L.ncclGroupEnd()
    res[r] = (rc, L.ncclGetErrorString(rc).decode())
th = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
[t.start() for t in th]; [t.join() for t in th]
print(res)
assert res[0][0] != 0 and "sends nothing" in res[0][1], res
assert res[1][0] == 0, res          # an empty group is a no-op, as over RCCL
''' % lib
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, env=dict(os.environ, MOCK_RCCL_TIMEOUT_S="3"))
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]


TWO_COMMS = r"""
import sys, threading
sys.path.insert(0, %(root)r)
import numpy as np
from homulator_amd import host
from oracle.homoracle import Oracle
world, cfg, L, ell, alpha, logN = 4, "config_4_N15.cfg", 16, 10, 4, 15
uids = [host.rccl_unique_id(), host.rccl_unique_id()]
ops = [[host.Op(cfg, "hmult", L, ell, alpha, rank=r, world=world, overrides={"seed": host.SEED + 7 * i, "shard_plan": %(plan)d}) for i in range(2)] for r in range(world)]
err = [None] * world
def work(r):
    try:
        for i in range(2): ops[r][i].comm_init_rccl(uids[i])       # communicator i of every rank, created in the same order
        for _ in range(3):
            for i in range(2): ops[r][i].enqueue(1)                 # A, B, A, B, ...: the order every rank uses
        for i in range(2): ops[r][i].sync()
    except Exception as e:
        err[r] = e
th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join() for t in th]
for e in err:
    if e is not None: raise e
o = Oracle(logN, L, alpha); o.set_threads(8)
for i in range(2):
    S = host.SEED + 7 * i
    exp = o.hmult(ell, o.synth_ct(ell, S), o.synth_ct(ell, S + 2000), o.synth_evk(ell, S + 10000))
    for name, want in (("out.c0", exp[0]), ("out.c1", exp[1])):
        full = np.zeros((ell - 1, 1 << logN), dtype=np.uint64)
        for r in range(world):
            mine = ops[r][i].read(name)
            for l in ops[r][i].owned(ell - 1): full[l] = mine[l]
        assert np.array_equal(full, want), (i, name)
print("two communicators per rank: ok")
"""


@pytest.mark.parametrize("plan", [1, 2], ids=["all-to-all", "gather"])
def test_two_sharded_instances_per_rank_with_a_communicator_each(plan):
    """the opt-in overlap mode of the bench (--sharded-streams 2): every rank drives two sharded instances, each with its own RCCL
    communicator and stream, enqueued alternately; through the double, 4 ranks, three passes each, both instances bit-exact; both sharded plans"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "mock_rccl")], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "tests", "mock_rccl", "libmockrccl.so")
    r = subprocess.run([sys.executable, "-c", TWO_COMMS % {"root": ROOT, "plan": plan}], env=dict(os.environ, HOMULATOR_RCCL_LIB=lib), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "two communicators per rank: ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
