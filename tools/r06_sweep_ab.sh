#!/bin/bash
# round 6: the wide-digit parameter sets (A: N = 2^15, alpha = 28; motivation: N = 2^16, alpha = 28) at HEAD against the plan of round 5 on ONE box,
# interleaved, a handful of levels; then the dist tests' rehearsal of the plain `python bench.py --gpus 2`
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_sweep_ab; mkdir -p $OUT
export TMPDIR=/tmp
for r in 1 2; do
for plan in head r5; do
  timeout -k 10 400 python3 script/sweep.py --bench --set A,motivation --ops hmult,hrotate --levels 28,20,16,8 --chains mont32 --plan $plan > $OUT/${plan}_$r.txt 2> $OUT/${plan}_$r.err; echo "$plan $r rc=$?"
done
done
paste -d'|' $OUT/head_1.txt $OUT/r5_1.txt | awk -F'|' '{print $1; print "   r5: " $2}' | head -80
