"""per-launch device time of one batched op (each launch alone on the chip), per op of the batch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import host
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
op = host.Op("config_4.cfg", sys.argv[2] if len(sys.argv) > 2 else "hmult", 45, 35, 15, overrides={"batch": B})
op.execute(2)
rows = op.stage_times(5)
tot = 0
for kind, name, ns in rows:
    print(f"{kind:13s} {ns*1e-3/B:8.2f} us/op   {name[:70]}")
    tot += ns
print(f"sum {tot*1e-3/B:.1f} us/op; whole op {op.execute(10)*1e-3/B:.1f} us/op at batch {B}")
