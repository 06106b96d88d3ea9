#!/bin/bash
# per-kernel durations of the ModUp path (tools/bcol_ab.py at batch 10, kernels alone on the GPU) for the default library and A/B builds
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r03u}; shift; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for v in default "$@"; do
  if [ $v = default ]; then unset HOMULATOR_HIP_LIB; else export HOMULATOR_HIP_LIB=$ROOT/ab_builds/$v/libhomulator_hip.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT/kt_$v -o kt --output-format csv -- python3 $ROOT/tools/bcol_ab.py 10 3 > $OUT/kt_$v.log 2>&1 || echo "$v failed"
  python3 - <<P >> $OUT/kernels.txt
import csv, glob, collections
rows=[]
for f in glob.glob("$OUT/kt_$v/**/*kernel_trace.csv", recursive=True): rows+=list(csv.DictReader(open(f)))
agg=collections.defaultdict(list)
for r in rows: agg[r['Kernel_Name'].replace("void ","")[:40]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))*1e-3)
print("== $v")
for k,t in sorted(agg.items()):
    if k.startswith(("k_fill","__amd")): continue
    t.sort(); print(f"  {k:40s} n={len(t):3d} median {t[len(t)//2]:8.1f} us  min {t[0]:8.1f}")
P
done
cat $OUT/kernels.txt
