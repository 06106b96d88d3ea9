"""one-launch transform (k_ntt_fused) against the two-kernel transform, same process, interleaved rounds: us per limb-NTT"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
def t(fn, iters):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3
print(os.path.basename(os.environ.get("HOMULATOR_HIP_LIB", "default")))
for n in (50, 128, 512, 1150):
    sets = 6 if n <= 128 else 2
    bufs = [(ctx.alloc(n), ctx.alloc(n)) for _ in range(sets)]
    ids = list(range(50)) if n == 50 else [(i // 2) % 60 for i in range(n)]
    for a, _ in bufs: ctx.fill_uniform(a, ids, 1)
    res = {}
    for rnd in range(3):
        for fused in (1, 0):
            ctx.set_option("ntt_fused", fused)
            k = [0]
            def f(inv=False):
                a, b = bufs[k[0] % sets]; k[0] += 1
                ctx.ntt(a, b, ids, inverse=inv)
            it = 48 if n <= 128 else 12
            res.setdefault((fused, 0), []).append(t(lambda: f(False), it))
            res.setdefault((fused, 1), []).append(t(lambda: f(True), it))
    line = f"n={n:5d}:"
    for fused in (1, 0):
        for inv in (0, 1):
            v = sorted(res[(fused, inv)])
            line += f"  {'fused' if fused else '2-krn'} {'inv' if inv else 'fwd'} {v[1]:8.1f} us ({v[1]/n:.3f}/limb, min {v[0]/n:.3f})"
    print(line, flush=True)
    for a, b in bufs: a.free(); b.free()
print("cross-XCD limb-polys:", ctx.counter("ntt_cross_xcd"))
