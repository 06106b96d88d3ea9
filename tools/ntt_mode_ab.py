"""which pass of a small transform launch should use the 8-coefficient geometry?  hm_set_option ntt_small_mode: 3 = both (default), 1 = COL only,
2 = ROW only, 0 = neither; us per hm_ntt call of n limb-polys, two interleaved rounds.  Measured: all four within the run-to-run spread."""
import os, sys
sys.path.insert(0, os.getcwd())
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
def t(fn, iters):
    for _ in range(300): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3
for n in (35, 50, 64):
    bufs = [(ctx.alloc(n), ctx.alloc(n)) for _ in range(6)]
    ids = (list(range(50)) * 2)[:n]
    for a, _ in bufs: ctx.fill_uniform(a, ids, 1)
    ctx.set_option("ntt_small_limbs", 64)
    out = f"n={n}:"
    for rnd in range(2):
        for mode in (3, 1, 2, 0):
            ctx.set_option("ntt_small_mode", mode)
            k = [0]
            def f(inv=False):
                a, b = bufs[k[0] % 6]; k[0] += 1
                ctx.ntt(a, b, ids, inverse=inv)
            out += f"  mode{mode} fwd {t(lambda: f(False), 60):6.1f} inv {t(lambda: f(True), 60):6.1f}"
        out += " |"
    print(out, flush=True)
