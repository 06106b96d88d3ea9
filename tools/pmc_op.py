"""ops at a given launch shape, for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (whole-op HBM traffic).
usage: python3 tools/pmc_op.py [op] [rounds] [batch] [instances]
  `instances` ops in flight (own HBM pool / stream each), every launch carrying `batch` ops, enqueued alternately for `rounds`
  rounds: the timed region of bench.py is (hmult, *, 10, 2).  Ops executed = rounds x batch x instances."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from homulator_amd import host
opn = sys.argv[1] if len(sys.argv) > 1 else "hmult"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
inst = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ops = [host.Op("config_4.cfg", opn, 45, 35, 15, overrides={"seed": host.SEED + 7 * i, **({"batch": batch} if batch > 1 else {})}) for i in range(inst)]
for _ in range(rounds):
    for o in ops:
        o.enqueue(1)
for o in ops:
    o.sync()
    o.close()
