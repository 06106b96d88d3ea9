"""small-launch geometry (8 coefficients per thread) against the wide one: us per transform launch, interleaved rounds"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
def t(fn, iters):
    for _ in range(200): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3
for n in (2, 16, 32, 35, 50, 68, 115, 128, 192, 256):
    sets = 6
    bufs = [(ctx.alloc(n), ctx.alloc(n)) for _ in range(sets)]
    ids = (list(range(50)) * 6)[:n]
    for a, _ in bufs: ctx.fill_uniform(a, ids, 1)
    res = {}
    for rnd in range(3):
        for small in (4096, 0):
            ctx.set_option("ntt_small_limbs", small)
            k = [0]
            def f(inv=False):
                a, b = bufs[k[0] % sets]; k[0] += 1
                ctx.ntt(a, b, ids, inverse=inv)
            res.setdefault((small, 0), []).append(t(lambda: f(False), 48))
            res.setdefault((small, 1), []).append(t(lambda: f(True), 48))
    line = f"n={n:4d}:"
    for small in (4096, 0):
        for inv in (0, 1):
            v = sorted(res[(small, inv)])
            line += f"  {'ept8 ' if small else 'ept16'} {'inv' if inv else 'fwd'} {v[1]:7.1f} us"
    print(line, flush=True)
    for a, b in bufs: a.free(); b.free()
