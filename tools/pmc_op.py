"""one op a few times, for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (whole-op HBM traffic).
usage: python3 tools/pmc_op.py [op] [iters] ; HOMULATOR_BATCH applies"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from homulator_amd import host
opn = sys.argv[1] if len(sys.argv) > 1 else "hmult"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
op = host.Op("config_4.cfg", opn, 45, 35, 15)
op.execute(iters)
op.close()
