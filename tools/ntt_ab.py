"""A/B timing of the NTT launches for the library named by HOMULATOR_HIP_LIB (ablation builds: tools/ablate.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
def t(fn, iters=20):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3
out = [os.path.basename(os.environ.get("HOMULATOR_HIP_LIB", "default"))]
for n in (50, 128, 512):
    a, b = ctx.alloc(n), ctx.alloc(n)
    ids = list(range(50)) if n == 50 else [(i // 2) % 60 for i in range(n)]
    ctx.fill_uniform(a, ids, 1)
    us = t(lambda: ctx.ntt(a, b, ids)); usi = t(lambda: ctx.ntt(a, b, ids, inverse=True))
    out.append(f"n={n}: fwd {us:7.1f} ({us/n:.3f}/limb) inv {usi:7.1f} ({usi/n:.3f}/limb)")
    a.free(); b.free()
print(" | ".join(out))
