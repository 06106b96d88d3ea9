"""base conversion at the shapes of the hmult plan: ModUp (3 digits: 15->35, 15->35, 5->45) and ModDown (2 keys: 15->35)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip
L, K = 45, 15
ctx = hip.Context(16, L, K)
def t(fn, iters=30):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3
src, dst = ctx.alloc(35), ctx.alloc(115)
ctx.fill_uniform(src, list(range(35)), 9)
P = [L + i for i in range(K)]
d0 = (list(range(0, 15)), list(range(15, 35)) + P)
d1 = (list(range(15, 30)), list(range(0, 15)) + list(range(30, 35)) + P)
d2 = (list(range(30, 35)), list(range(0, 30)) + P)
def modup():
    ctx.bconv_batch([(src, d0[0], d0[0], dst, list(range(0, 35)), d0[1]), (src, d1[0], d1[0], dst, list(range(35, 70)), d1[1]),
                     (src, d2[0], d2[0], dst, list(range(70, 115)), d2[1])])
print(f"ModUp conversion (15->35, 15->35, 5->45): {t(modup):7.1f} us" if hasattr(ctx, "bconv_batch") else "no bconv_batch binding")
us = t(lambda: ctx.bconv(src, d0[0], dst, d0[1]))
print(f"single 15->35: {us:7.1f} us  ({15*35*65536/us*1e-3:6.1f} GMAC/s)")
# the batched plan's shape: B ops -> 2B conversions 15->35 and B conversions 5->45 in one call
for B in (4, 10):
    srcs = [ctx.alloc(35) for _ in range(B)]; dsts = [ctx.alloc(115) for _ in range(B)]
    for s_ in srcs: ctx.fill_uniform(s_, list(range(35)), 9)
    probs = []
    for s_, d_ in zip(srcs, dsts):
        probs += [(s_, d0[0], d0[0], d_, list(range(0, 35)), d0[1]), (s_, d1[0], d1[0], d_, list(range(35, 70)), d1[1]), (s_, d2[0], d2[0], d_, list(range(70, 115)), d2[1])]
    us = t(lambda: ctx.bconv_batch(probs))
    print(f"ModUp conversion, batch {B}: {us:7.1f} us = {us / B:6.1f} us per op")
