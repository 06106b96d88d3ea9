#!/bin/bash
# per-kernel durations of one batched hmult (batch 10, kernels alone on the chip) with and without the packed conversion inputs
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_pack_ab; mkdir -p $OUT
export TMPDIR=/tmp HOMULATOR_BATCH=10
cd /tmp
for v in 1 0; do
  HOMULATOR_PACK_BCONV_IN=$v timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT/kt_$v -o kt --output-format csv -- python3 $ROOT/tools/pmc_op.py hmult 4 > $OUT/kt_$v.log 2>&1 || echo "$v failed"
  python3 - <<P
import csv, glob, collections
rows=[]
for f in glob.glob("$OUT/kt_$v/**/*kernel_trace.csv", recursive=True): rows+=list(csv.DictReader(open(f)))
agg=collections.defaultdict(list)
for r in rows: agg[r['Kernel_Name'].replace("void ","")[:44]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))*1e-3)
print("== pack_bconv_in = $v")
tot=0
for k,t in sorted(agg.items()):
    if k.startswith(("k_fill","__amd")): continue
    t.sort(); print(f"  {k:44s} n={len(t):3d} median {t[len(t)//2]:8.1f} us"); tot+=t[len(t)//2]
print("  sum of medians", round(tot,1))
P
done
