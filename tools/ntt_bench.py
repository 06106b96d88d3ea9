"""NTT timing at the launch sizes of the hmult plan (C-level loop through the host layer is not needed: hipEvents
around back-to-back launches; sizes big enough that the Python call overhead hides behind the GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
def t(fn, iters=30):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3
a, b = ctx.alloc(115), ctx.alloc(115)
ctx.fill_uniform(a, [i % 60 for i in range(115)], 1)
for name, ids in (("fwd 70 (35 pairs)", list(range(35)) * 2), ("fwd 115 (ModUp)", list(range(35)) * 2 + [45 + i % 15 for i in range(45)]),
                  ("fwd 35 singles", list(range(35)))):
    print(f"{name:22s} {t(lambda: ctx.ntt(a, b, ids)):7.1f} us")
for name, ids in (("inv 35 singles", list(range(35))), ("inv 30 (15 pairs)", [45 + i % 15 for i in range(30)]), ("inv 2", [34, 34])):
    print(f"{name:22s} {t(lambda: ctx.ntt(a, b, ids, inverse=True)):7.1f} us")
