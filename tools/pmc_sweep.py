"""the roofline leg of bench.py alone (forward NTT sweep over the 50 limbs of the extended basis), for the rocprofv3 --pmc passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
ids = ctx.ext_ids(35)
a, b = ctx.alloc(50), ctx.alloc(50)
ctx.fill_uniform(a, ids, 1)
for _ in range(10):
    ctx.ntt(a, b, ids)
ctx.sync()
