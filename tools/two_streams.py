"""throughput of 1 vs 2 vs 3 independent hmult ops in flight (each op has its own context / stream / HBM pool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from homulator_amd import host
for k in (1, 2, 3):
    ops = [host.Op("config_4.cfg", "hmult", 45, 35, 15) for _ in range(k)]
    for o in ops: o.enqueue(5)
    for o in ops: o.sync()
    steps = 100
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        for o in ops: o.enqueue(1)
    for o in ops: o.sync()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{k} concurrent ops: {k*steps/dt:8.1f} ops/s  ({dt/steps*1e6:7.1f} us per round)")
    for o in ops: o.close()
