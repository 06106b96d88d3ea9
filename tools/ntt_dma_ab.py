"""persistent double-buffered passes (k_ntt_*_dma) against the one-tile-per-workgroup passes, same process, interleaved rounds"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
def t(fn, iters):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 16, 35, 50, 128, 512, 1150]
arms = [(0, 0)] + [tuple(int(v) for v in a.split(":")) for a in os.environ.get("DMA_ARMS", "1:512,1:256,1:768,2:512,2:1024").split(",")]
print(os.path.basename(os.environ.get("HOMULATOR_HIP_LIB", "default")))
for n in sizes:
    sets = 6 if n <= 128 else 2
    bufs = [(ctx.alloc(n), ctx.alloc(n)) for _ in range(sets)]
    ids = [i % 50 for i in range(n)] if n != 1150 else [(i // 2) % 60 for i in range(n)]
    for a, _ in bufs: ctx.fill_uniform(a, ids, 1)
    res = {}
    for rnd in range(3):
        for arm in arms:
            ctx.set_option("ntt_dma", arm[0]); ctx.set_option("ntt_dma_wgs", arm[1])
            for inv in (0, 1):
                k = [0]
                def f():
                    a, b = bufs[k[0] % sets]; k[0] += 1
                    ctx.ntt(a, b, ids, inverse=bool(inv))
                res.setdefault((arm, inv), []).append(t(f, 48 if n <= 128 else 12))
    print(f"n={n:5d}:")
    for (arm, inv), v in res.items():
        v = sorted(v)
        name = "one tile per workgroup" if arm[0] == 0 else f"dma geo{8 if arm[0] == 1 else 16} wgs={arm[1]}"
        print(f"   {name:28s} {'inv' if inv else 'fwd'} {v[1]:8.1f} us ({v[1]/n:.3f}/limb, min {v[0]/n:.3f})", flush=True)
    for a, b in bufs: a.free(); b.free()
