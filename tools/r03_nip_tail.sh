#!/bin/bash
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03p; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 python3 $ROOT/tools/nip_tail.py 20 > $OUT/nip_tail.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt --output-format csv -- python3 $ROOT/tools/nip_tail.py 3 > $OUT/kt.log 2>&1
python3 - <<P > $OUT/nip_tail_kernels.txt
import csv, glob
f = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    n = r["Kernel_Name"]
    if "row_ip" in n or "ntt_col" in n:
        print(n[:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
P
cat $OUT/nip_tail.txt
