"""forward NTT of 128 limb-polys (64 same-modulus pairs: the launch size of the batched plan) for rocprofv3 --pmc passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
ids = [i % 50 if i % 50 < 45 else i % 50 for i in range(128)]
ids = sorted([i % 35 for i in range(128)])
a, b = ctx.alloc(128), ctx.alloc(128)
ctx.fill_uniform(a, ids, 1)
for _ in range(6):
    ctx.ntt(a, b, ids)
ctx.sync()
