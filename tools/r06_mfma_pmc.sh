#!/bin/bash
# round 6: counters of the stand-alone MFMA base conversion (tools/bconv_mfma) beside k_bconv<15>: where its time goes
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_mfma; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p --output-format csv -- $ROOT/tools/bconv_mfma 20 15 35 4 > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
done
python3 - <<P
import csv, glob, collections
for i in range(1, 7):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$OUT/p%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "")[:28]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    for k, v in sorted(agg.items()):
        if k.startswith("k_fill"): continue
        print(f"pass {i} {k:28s} " + "  ".join(f"{c}={x / max(1, cnt[(k, c)]):.4g}" for c, x in sorted(v.items())) + "  (per launch)")
P
