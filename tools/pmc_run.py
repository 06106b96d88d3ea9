"""Small fixed workload for rocprofv3 --pmc passes: forward NTT (70 limbs), inverse NTT (35 limbs), ModUp-shaped
base conversion batch, one EWE MAC2; 3 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip

L, K = 45, 15
ctx = hip.Context(16, L, K)
ext = ctx.ext_ids(35)
ids70 = list(range(35)) * 2
a, b = ctx.alloc(115), ctx.alloc(115)
ctx.fill_uniform(a, [i % 60 for i in range(115)], 1)
for _ in range(3):
    ctx.ntt(a, b, ids70)
    ctx.ntt(a, b, list(range(35)), inverse=True, scale=[3] * 35)
    ctx.bconv(a, list(range(15)), b, list(range(15, 35)) + [L + i for i in range(K)])
    ctx.ewe(hip.OP_MAC2, b, ext, a=a, b=a, c=a, d=a)
ctx.sync()
