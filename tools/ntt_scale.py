"""forward NTT time vs launch size (does the per-limb time approach max(compute, memory) for big launches?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
def t(fn, iters=20):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3
for n in (16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024):
    a, b = ctx.alloc(n), ctx.alloc(n)
    ids = [(i // 2) % 60 for i in range(n)]
    ctx.fill_uniform(a, ids, 1)
    us = t(lambda: ctx.ntt(a, b, ids))
    usi = t(lambda: ctx.ntt(a, b, ids, inverse=True))
    print(f"n={n:5d} fwd {us:8.1f} us  {us/n:6.3f} us/limb   inv {usi:8.1f} us {usi/n:6.3f} us/limb")
    a.free(); b.free()
