import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import host
for b in (1, 2, 3, 4, 6):
    op = host.Op("config_4.cfg", "hmult", 45, 35, 15, overrides={"batch": b})
    op.execute(3)
    ns = op.execute(30)
    print(f"batch {b}: {ns*1e-3/b:.1f} us per op  ({1e9/ns*b:.0f} ops/s single stream)", flush=True)
    op.close()
