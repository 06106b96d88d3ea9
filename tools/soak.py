"""soak: two batched hmult instances at full size interleaved for many iterations, outputs re-checked against the oracle
usage: python tools/soak.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from homulator_amd import host
from oracle.homoracle import Oracle
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
o = Oracle(16, 45, 15); o.set_threads(16)
evk = {i: o.synth_evk(35, host.SEED + 7 * i + 10000) for i in range(2)}
ops = [host.Op("config_4.cfg", "hmult", 45, 35, 15, overrides={"seed": host.SEED + 7 * i, "batch": 4}) for i in range(2)]
t0 = time.time()
for it in range(iters):
    ops[it % 2].enqueue(1)
    if it % 2000 == 1999:
        for op in ops: op.sync()
        print(f"{it + 1} enqueues ({(it + 1) * 4} hmults), {time.time() - t0:.1f} s", flush=True)
for op in ops: op.sync()
bad = 0
for i, op in enumerate(ops):
    for c in range(4):
        s = host.SEED + 7 * i + c * 100000
        exp = o.hmult(35, o.synth_ct(35, s), o.synth_ct(35, s + 2000), evk[i])
        ok = np.array_equal(op.read("out.c0", copy=c), exp[0]) and np.array_equal(op.read("out.c1", copy=c), exp[1])
        bad += not ok
print("soak", "OK" if not bad else f"MISMATCH in {bad} outputs", f"{iters * 4 / (time.time() - t0):.0f} hmult/s incl. syncs")
sys.exit(1 if bad else 0)
