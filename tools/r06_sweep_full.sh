#!/bin/bash
# round 6: the reference's parameter sets A / B / C / D + `motivation` (script/README.md:17-22, script/motivation) x hmult / hrotate x EVERY level x
# {mont32, survey} at HEAD (device us per op, SURVEY.md 8(d) bytes, fraction of the HBM peak), then the wide-digit sets again with the plan of
# round 5 on the same box -> gpurun_out/r06_sweep/
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_sweep; mkdir -p $OUT
export TMPDIR=/tmp
for set in A B C D motivation; do
  timeout -k 10 600 python3 script/sweep.py --bench --set $set --ops hmult,hrotate --chains mont32,survey > $OUT/head_$set.txt 2> $OUT/head_$set.err; echo "head $set rc=$? $(wc -l < $OUT/head_$set.txt) lines"
done
for set in A motivation; do
  timeout -k 10 600 python3 script/sweep.py --bench --set $set --ops hmult,hrotate --chains mont32 --plan r5 > $OUT/r5_$set.txt 2> $OUT/r5_$set.err; echo "r5 $set rc=$?"
  timeout -k 10 600 python3 script/sweep.py --bench --set $set --ops hmult,hrotate --chains mont32 > $OUT/head2_$set.txt 2> $OUT/head2_$set.err; echo "head2 $set rc=$?"
done
