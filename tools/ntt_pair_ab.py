"""small transform launches: one tile per workgroup (ntt_small_mode 3) against two tiles per workgroup, pipelined (15 = both passes, 7 = COL only,
11 = ROW only); us per hm_ntt call of n limb-polys over 6 rotating buffer pairs, interleaved rounds.  Checks the results of the modes against each
other first."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
def t(fn, iters):
    for _ in range(200): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3
ctx.set_option("ntt_small_limbs", 64)
n = 13
ids = [(7 * i + 1) % 60 for i in range(n)]
a, b3, b15 = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
ctx.fill_uniform(a, ids, 3)
for inv in (False, True):
    ctx.set_option("ntt_small_mode", 3); ctx.ntt(a, b3, ids, inverse=inv)
    ctx.set_option("ntt_small_mode", 15); ctx.ntt(a, b15, ids, inverse=inv)
    print("inverse" if inv else "forward", "paired == single:", np.array_equal(b3.download(), b15.download()), flush=True)
for n in (2, 16, 30, 35, 50, 64):
    bufs = [(ctx.alloc(n), ctx.alloc(n)) for _ in range(6)]
    ids = (list(range(50)) * 2)[:n]
    for x, _ in bufs: ctx.fill_uniform(x, ids, 1)
    out = f"n={n:3d}:"
    for rnd in range(2):
        for mode in (3, 15, 7, 11):
            ctx.set_option("ntt_small_mode", mode)
            k = [0]
            def f(inv=False):
                x, y = bufs[k[0] % 6]; k[0] += 1
                ctx.ntt(x, y, ids, inverse=inv)
            out += f"  m{mode:<2d} fwd {t(lambda: f(False), 60):5.1f} inv {t(lambda: f(True), 60):5.1f}"
        out += " |"
    print(out, flush=True)
