"""timing of the one-launch transform with and without the occupancy throttle against the two-kernel transform (interleaved, one process)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
def t(fn, iters):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3
for n in (8, 16, 50, 128, 512):
    sets = 6 if n <= 128 else 2
    bufs = [(ctx.alloc(n), ctx.alloc(n)) for _ in range(sets)]
    ids = [i % 50 for i in range(n)]
    for a, _ in bufs: ctx.fill_uniform(a, ids, 1)
    res = {}
    arms = [("2-kernel", 0, 0), ("fused", 1, 0), ("fused 2WG/CU", 1, 28 * 1024), ("fused 1WG/CU", 1, 100 * 1024)]
    for rnd in range(3):
        for name, fused, lds in arms:
            ctx.set_option("ntt_fused", fused); ctx.set_option("ntt_fused_lds", lds); ctx.set_option("ntt_small_limbs", 0)
            k = [0]
            def f():
                a, b = bufs[k[0] % sets]; k[0] += 1
                ctx.ntt(a, b, ids)
            res.setdefault(name, []).append(t(f, 48 if n <= 128 else 12))
    print(f"n={n:4d}: " + "  ".join(f"{name} {sorted(v)[1]:7.1f} us ({sorted(v)[1]/n:.3f}/limb)" for name, v in res.items()), flush=True)
    for a, b in bufs: a.free(); b.free()
print("cross-XCD limb-polys:", ctx.counter("ntt_cross_xcd"))
