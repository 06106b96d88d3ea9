#!/bin/bash
# per-kernel counters of one batched hmult (batch 10): tools/pmc_kernels.sh <tag> [env assignments...]
ROOT=$(pwd); TAG=$1; shift; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HOMULATOR_BATCH=10 "$@"
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p --output-format csv -- python3 $ROOT/tools/pmc_op.py hmult 2 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 > $OUT/summary.txt 2>&1
python3 - <<P
import csv, glob, collections
rows=[]
for f in glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True):
    rows+=list(csv.DictReader(open(f)))
agg=collections.defaultdict(list)
for r in rows:
    agg[r['Kernel_Name'].replace("void ","")[:44]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))*1e-3)
for k,v in sorted(agg.items()):
    if k.startswith("k_fill") or k.startswith("__amd"): continue
    print(f"{k:44s} n={len(v):3d} median {sorted(v)[len(v)//2]:9.1f} us  total {sum(v):9.1f}")
P
cat $OUT/summary.txt
