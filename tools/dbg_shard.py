import sys, os, threading, subprocess
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
import numpy as np
subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "mock_rccl")], stdout=subprocess.DEVNULL)
os.environ["HOMULATOR_RCCL_LIB"] = os.path.join(ROOT, "tests", "mock_rccl", "libmockrccl.so")
from homulator_amd import host
from oracle.homoracle import Oracle
SEED = 0x484F4D55
world, cfg, opname, L, ell, alpha, logN = 8, "config_4.cfg", sys.argv[1], 45, 35, 15, 16
o = Oracle(logN, L, alpha); o.set_threads(8)
evk = o.synth_evk(ell, SEED + 10000)
ct1, ct2 = o.synth_ct(ell, SEED), o.synth_ct(ell, SEED + 2000)
exp = o.hmult(ell, ct1, ct2, evk) if opname == "hmult" else o.hrotate(ell, ct1, 5, evk)
n_out = ell - 1 if opname == "hmult" else ell
for pipe in [int(x) for x in sys.argv[2].split(',')]:
    for runs in [2] * int(sys.argv[3]):
        uid = host.rccl_unique_id()
        ops = [host.Op(cfg, opname, L, ell, alpha, rank=r, world=world, overrides={"pipeline_digits": pipe}) for r in range(world)]
        err = [None] * world
        def work(r):
            try:
                ops[r].comm_init_rccl(uid)
                for _ in range(runs): ops[r].execute(1)
            except Exception as e: err[r] = e
        th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        [t.start() for t in th]; [t.join() for t in th]
        bad = {}
        for name, e in (("out.c0", exp[0]), ("out.c1", exp[1])):
            for r, op in enumerate(ops):
                mine = op.read(name)
                for l in op.owned(n_out):
                    if not np.array_equal(mine[l], e[l]): bad.setdefault(name, []).append((r, l, int((mine[l] != e[l]).sum())))
        print(f"pipe={pipe} runs={runs} err={[str(x) for x in err if x]} bad={bad}", flush=True)
        for op in ops: op.close()
