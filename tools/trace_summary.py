"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) median duration, and per-op totals.
usage: python tools/trace_summary.py <rocprof output dir> [ops in the trace]"""
import collections
import csv
import glob
import sys

path = sys.argv[1]
ops = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = (glob.glob(path + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].replace("void ", "")[:46]
    grid = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    agg[(name, grid, r['VGPR_Count'], r['LDS_Block_Size'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in agg.values())
print(f"total kernel time {tot/1e3:.1f} us; per op ({ops:g} ops) {tot/1e3/ops:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{k[0]:46s} wgs={str(k[1]):16s} vgpr={k[2]:>3s} lds={k[3]:>6s} n={len(v):4d} med={v[len(v)//2]/1e3:8.2f}us sum/op={sum(v)/1e3/ops:8.1f}us")
