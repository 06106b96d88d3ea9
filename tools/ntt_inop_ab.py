"""the in-op transform launch (bench.measure_ntt_inop) for the library named by HOMULATOR_HIP_LIB: us per limb-NTT"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for r in range(3):
    ns, n = bench.measure_ntt_inop(10)
    print(os.path.basename(os.environ.get("HOMULATOR_HIP_LIB", "default")), n, round(ns * 1e-3 / n, 4), flush=True)
