"""the roofline leg of bench.py alone (forward NTT sweep over the 50 limbs of the extended basis), for the rocprofv3 --pmc passes.
Rotates over 6 buffer pairs (315 MB: past the 256 MiB Infinity Cache) exactly as bench.py's measure_ntt_sweep does, so that the
counters and the timing describe the same launches.  HOMULATOR_NTT_FUSED=0 selects the two-kernel transform."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from homulator_amd import hip
ctx = hip.Context(16, 45, 15)
ids = ctx.ext_ids(35)
bufs = [(ctx.alloc(50), ctx.alloc(50)) for _ in range(6)]
for i, (a, _) in enumerate(bufs):
    ctx.fill_uniform(a, ids, 1 + i)
for i in range(24):
    a, b = bufs[i % 6]
    ctx.ntt(a, b, ids)
ctx.sync()
