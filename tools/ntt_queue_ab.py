"""persistent queue transform (k_ntt_queue) against the two-kernel transform, same process, interleaved rounds: us per launch and per limb-NTT.
argv: [sizes, comma separated] ; env QUEUE_ARMS="geo:wgs:la:gc,..." overrides the arms"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
def t(fn, iters):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 50, 128, 512, 1150]
arms = [(0, 0, 2, 0)] + [tuple(int(v) for v in a.split(":")) for a in os.environ.get("QUEUE_ARMS", "1:512:2:0,1:512:3:0,1:512:4:0,1:768:4:0,1:1024:4:0").split(",")]
print(os.path.basename(os.environ.get("HOMULATOR_HIP_LIB", "default")))
for n in sizes:
    sets = 6 if n <= 128 else 2
    bufs = [(ctx.alloc(n), ctx.alloc(n)) for _ in range(sets)]
    ids = [i % 50 for i in range(n)] if n != 1150 else [(i // 2) % 60 for i in range(n)]
    for a, _ in bufs: ctx.fill_uniform(a, ids, 1)
    res = {}
    for rnd in range(3):
        for arm in arms:
            geo, wgs, la, gc = arm
            ctx.set_option("ntt_queue", geo); ctx.set_option("ntt_queue_wgs", wgs); ctx.set_option("ntt_queue_lookahead", la); ctx.set_option("ntt_queue_group", gc)
            for inplace in (0, 1):
                k = [0]
                def f():
                    a, b = bufs[k[0] % sets]; k[0] += 1
                    ctx.ntt(a, a if inplace else b, ids)
                res.setdefault((arm, inplace), []).append(t(f, 48 if n <= 128 else 12))
    print(f"n={n:5d}:")
    for (arm, inplace), v in res.items():
        v = sorted(v)
        name = "two-kernel" if arm[0] == 0 else f"queue wgs={arm[1]} la={arm[2]} gc={arm[3]}"
        print(f"   {name:36s} {'in place ' if inplace else 'out-of-pl'} {v[1]:8.1f} us ({v[1]/n:.3f}/limb, min {v[0]/n:.3f})", flush=True)
    for a, b in bufs: a.free(); b.free()
