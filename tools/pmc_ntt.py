"""forward + inverse NTT of 128 limb-polys (64 same-modulus pairs) and of 50 distinct moduli, for rocprofv3 passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
ids = [(i // 2) % 60 for i in range(128)]
a, b = ctx.alloc(128), ctx.alloc(128)
ctx.fill_uniform(a, ids, 1)
for _ in range(5):
    ctx.ntt(a, b, ids)
    ctx.ntt(a, b, ids, inverse=True)
ctx.sync()
