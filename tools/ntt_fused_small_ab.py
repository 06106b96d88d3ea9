"""both passes in one launch of the small-launch geometry (k_ntt_fused8, hm_set_option ntt_fused_small) against two kernels (k_ntt_col8 + k_ntt_row8):
us per transform launch, 6 rotating buffer pairs, interleaved rounds, forward / inverse / in place"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
def t(fn, iters=60):
    for _ in range(60): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3
for n in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "8,16,35,50,64".split(","))]:
    bufs = [(ctx.alloc(n), ctx.alloc(n)) for _ in range(6)]
    ids = [i % 50 for i in range(n)]
    for a, _ in bufs: ctx.fill_uniform(a, ids, 1)
    res = {}
    for rnd in range(3):
        for one in (448, 0):
            ctx.set_option("ntt_fused_small", one)
            for inv in (False, True):
                for inplace in (False, True):
                    k = [0]
                    def f():
                        a, b = bufs[k[0] % 6]; k[0] += 1
                        ctx.ntt(a, a if inplace else b, ids, inverse=inv)
                    res.setdefault((one, inv, inplace), []).append(t(f))
    line = f"n={n:3d}:"
    for inv in (False, True):
        for inplace in (False, True):
            a, b = sorted(res[(448, inv, inplace)])[1], sorted(res[(0, inv, inplace)])[1]
            line += f"  {'inv' if inv else 'fwd'}{' in place' if inplace else ''}: one launch {a:5.1f} / two {b:5.1f} us"
    print(line, flush=True)
    for a, b in bufs: a.free(); b.free()
ctx.set_option("ntt_fused_small", 0)
