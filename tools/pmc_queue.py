"""one forward hm_ntt of n limb-polys per launch through the persistent queue kernel (or the two-kernel transform), rotating buffer pairs,
for rocprofv3 --pmc passes.  argv: n geo(0 = two-kernel) [wgs] [lookahead] [in place 0/1] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from homulator_amd import hip
n, geo = int(sys.argv[1]), int(sys.argv[2])
wgs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
la = int(sys.argv[4]) if len(sys.argv) > 4 else 2
inplace = int(sys.argv[5]) if len(sys.argv) > 5 else 0
launches = int(sys.argv[6]) if len(sys.argv) > 6 else 24
ctx = hip.Context(16, 45, 15)
ids = [i % 50 for i in range(n)]
ctx.set_option("ntt_queue", geo); ctx.set_option("ntt_queue_wgs", wgs); ctx.set_option("ntt_queue_lookahead", la)
if not geo:
    ctx.set_option("ntt_small_limbs", 0)
sets = 6 if n <= 128 else 2
bufs = [(ctx.alloc(n), ctx.alloc(n)) for _ in range(sets)]
for i, (a, _) in enumerate(bufs):
    ctx.fill_uniform(a, ids, 1 + i)
for i in range(launches):
    a, b = bufs[i % sets]
    ctx.ntt(a, a if inplace else b, ids)
ctx.sync()
