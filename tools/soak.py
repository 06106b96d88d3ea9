"""soak: two batched hmult instances at full size interleaved for many iterations.  The plan recomputes the same
deterministic outputs on every pass; all outputs (2 instances x batch B) are compared with the oracle at EVERY periodic
synchronisation (every 2000 enqueues) and at the end, so a corruption at any point of the run is seen at the next check.
B = 4: the batched launches (two-kernel transforms); B = 1: every transform of both instances is a one-launch transform
(k_ntt_fused8: two kernels with rendezvous in flight on one GPU at a time).
usage: python tools/soak.py [iterations] [batch] [hmult|hrotate]   (hrotate, round 6: the plan without an automorphism launch — the ModUp INTT, the key
product and the final add read the ciphertext through the automorphism)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from homulator_amd import host
from oracle.homoracle import Oracle
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
OPN = sys.argv[3] if len(sys.argv) > 3 else "hmult"
o = Oracle(16, 45, 15); o.set_threads(16)
evk = {i: o.synth_evk(35, host.SEED + 7 * i + 10000) for i in range(2)}
ops = [host.Op("config_4.cfg", OPN, 45, 35, 15, overrides={"seed": host.SEED + 7 * i, "batch": B}) for i in range(2)]
exp = {}
for i in range(2):
    for c in range(B):
        s = host.SEED + 7 * i + c * 100000
        exp[i, c] = o.hmult(35, o.synth_ct(35, s), o.synth_ct(35, s + 2000), evk[i]) if OPN == "hmult" else o.hrotate(35, o.synth_ct(35, s), 5, evk[i])


def check():
    n = 0
    for i, op in enumerate(ops):
        for c in range(B):
            n += not (np.array_equal(op.read("out.c0", copy=c), exp[i, c][0]) and np.array_equal(op.read("out.c1", copy=c), exp[i, c][1]))
    return n


t0 = time.time()
bad = 0
for it in range(iters):
    ops[it % 2].enqueue(1)
    if it % 2000 == 1999:
        for op in ops: op.sync()
        bad += check()
        print(f"{it + 1} enqueues ({(it + 1) * B} {OPN}s), {time.time() - t0:.1f} s, mismatching outputs so far: {bad}", flush=True)
for op in ops: op.sync()
bad += check()
print("soak", "OK" if not bad else f"MISMATCH in {bad} outputs", f"{iters * B / (time.time() - t0):.0f} {OPN}/s incl. syncs")
sys.exit(1 if bad else 0)
