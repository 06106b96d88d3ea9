"""per-launch device time of one batched op (each launch alone on the chip), per op of the batch
    python3 tools/stage_times_batch.py [batch] [op] [cfg L l alpha] [key=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import host
a = sys.argv[1:]
kv = dict(x.split("=") for x in a if "=" in x)
a = [x for x in a if "=" not in x]
B = int(a[0]) if len(a) > 0 else 10
opn = a[1] if len(a) > 1 else "hmult"
cfg, L, ell, alpha = (a[2], int(a[3]), int(a[4]), int(a[5])) if len(a) > 5 else ("config_4.cfg", 45, 35, 15)
op = host.Op(cfg, opn, L, ell, alpha, overrides={"batch": B, **{k: int(v) for k, v in kv.items()}})
op.execute(2)
rows = op.stage_times(5)
tot = 0
print(f"# {cfg} {opn} {L} {ell} {alpha} batch {B} {kv}")
for kind, name, ns in rows:
    print(f"{kind:13s} {ns*1e-3/B:8.2f} us/op   {name[:70]}")
    tot += ns
print(f"sum {tot*1e-3/B:.1f} us/op; whole op {op.execute(10)*1e-3/B:.1f} us/op at batch {B}")
