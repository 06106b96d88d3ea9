"""the 50-limb forward sweep (6 rotating buffer pairs, as bench.py) for the library named by HOMULATOR_HIP_LIB (ablation builds: tools/ablate.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
def t(fn, iters=60):
    for _ in range(60): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3
out = [os.path.basename(os.environ.get("HOMULATOR_HIP_LIB", "default"))]
for n in (16, 35, 50, 100):
    bufs = [(ctx.alloc(n), ctx.alloc(n)) for _ in range(6)]
    ids = [i % 50 for i in range(n)]
    for a, _ in bufs: ctx.fill_uniform(a, ids, 1)
    k = [0]
    def f():
        a, b = bufs[k[0] % 6]; k[0] += 1
        ctx.ntt(a, b, ids)
    us = sorted(t(f) for _ in range(3))[1]
    out.append(f"n={n}: {us:6.1f} us")
    for a, b in bufs: a.free(); b.free()
print(" | ".join(out))
