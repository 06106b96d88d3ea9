"""forward NTT vs forward NTT with the fused (minuend - x) * k [+ addend] epilogue, at the launch sizes of the batched plan"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
def t(fn, iters=20):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3
for n in (70, 128, 280):
    ids = [i % 35 for i in range(n)]
    a, b, m, d = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
    ctx.fill_uniform(a, ids, 1); ctx.fill_uniform(m, ids, 2); ctx.fill_uniform(d, ids, 3)
    k = [3] * n
    print(f"n={n:4d}  plain {t(lambda: ctx.ntt(a, b, ids)):7.1f} us   sub_scale {t(lambda: ctx.ntt_sub_scale(a, m, b, ids, k)):7.1f} us   "
          f"sub_scale_add {t(lambda: ctx.ntt_sub_scale(a, m, b, ids, k, addend=d)):7.1f} us", flush=True)
    for x in (a, b, m, d): x.free()
