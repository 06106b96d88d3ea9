#!/bin/bash
# per-kernel durations of ONE hmult at a time (batch 1), kernels of the op in stream order
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_trace1; mkdir -p $OUT
export TMPDIR=/tmp HOMULATOR_BATCH=1
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT/kt -o kt --output-format csv -- python3 $ROOT/tools/pmc_op.py hmult 8 1 1 > $OUT/kt.log 2>&1 || echo "failed"
python3 - <<P
import csv, glob, collections
rows=[]
for f in glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True): rows+=list(csv.DictReader(open(f)))
rows=[r for r in rows if not r['Kernel_Name'].startswith(("k_fill","__amd","void k_fill"))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
agg=collections.defaultdict(list); order=[]
for r in rows:
    k=r['Kernel_Name'].replace("void ","")[:48]
    if k not in agg: order.append(k)
    agg[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))*1e-3)
tot=0
for k in order:
    t=sorted(agg[k]); n=len(t)//8 or 1; g=[r for r in rows if r['Kernel_Name'].replace('void ','')[:48]==k][0]
    print(f"  {k:48s} x{n} per op  median {t[len(t)//2]:7.1f} us  {int(g['Grid_Size_X'])//int(g['Workgroup_Size_X'])} workgroups of {g['Workgroup_Size_X']}"); tot+=t[len(t)//2]*n
print("  sum", round(tot,1))
# gaps: op span
ops=[rows[i:i+len(rows)//8] for i in range(0,len(rows),len(rows)//8)]
spans=[(int(o[-1]['End_Timestamp'])-int(o[0]['Start_Timestamp']))*1e-3 for o in ops if o]
print("  span per op (first kernel start to last kernel end), us:", [round(s,1) for s in spans])
P
