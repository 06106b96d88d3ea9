"""Per-launch device time of one op (each launch bracketed by its own HIP event pair).
usage: python tools/stage_times.py [op] [L] [l] [alpha] [cfg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import host
opn = sys.argv[1] if len(sys.argv) > 1 else "hmult"
L, l, al = (int(x) for x in (sys.argv[2:5] if len(sys.argv) > 4 else (45, 35, 15)))
cfg = sys.argv[5] if len(sys.argv) > 5 else "config_4.cfg"
op = host.Op(cfg, opn, L, l, al)
op.execute(3)
rows = op.stage_times(20)
tot = sum(r[2] for r in rows)
for kind, name, ns in rows:
    print(f"{kind:13s} {ns*1e-3:8.1f} us  {100*ns/tot:5.1f}%  {name[:110]}")
print(f"sum of launches {tot*1e-3:.1f} us; back-to-back {op.execute(50)*1e-3:.1f} us per op")
