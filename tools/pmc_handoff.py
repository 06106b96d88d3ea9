"""L2 hand-off experiment (round 4): one forward hm_ntt of n limb-polys per launch, rotating over 6 buffer pairs, for the rocprofv3 --pmc
passes.  argv: n [fused 0/1] [extra LDS bytes per workgroup (occupancy throttle of k_ntt_fused)] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from homulator_amd import hip
n = int(sys.argv[1]); fused = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lds = int(sys.argv[3]) if len(sys.argv) > 3 else 0
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 24
ctx = hip.Context(16, 45, 15)
ids = [i % 50 for i in range(n)]
ctx.set_option("ntt_fused", fused)
ctx.set_option("ntt_fused_lds", lds)
if not fused:
    ctx.set_option("ntt_small_limbs", 0)   # the wide geometry in both arms
sets = 6 if n <= 128 else 2
bufs = [(ctx.alloc(n), ctx.alloc(n)) for _ in range(sets)]
for i, (a, _) in enumerate(bufs):
    ctx.fill_uniform(a, ids, 1 + i)
for i in range(launches):
    a, b = bufs[i % sets]
    ctx.ntt(a, b, ids)
ctx.sync()
print("cross-XCD limb-polys:", ctx.counter("ntt_cross_xcd"))
