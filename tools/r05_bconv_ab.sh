#!/bin/bash
# round 5: what an instruction diet of the base conversions can buy at HEAD — timing-only ablation builds (wrong values, same
# traffic) against the shipped library: `packed` = no 30-bit split of the conversion's inputs in k_bconv_col (the most a pre-split input format or
# a split amortised over more outputs could save), `bhalf` = half of the multiply-adds of every conversion (more than a Karatsuba middle column saves)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_bconv_ab; mkdir -p $OUT
export TMPDIR=/tmp
rm -f $OUT/kernels.txt
bash tools/r03_kab.sh r05_bconv_ab packed bhalf > $OUT/kab.log 2>&1
cat $OUT/kernels.txt
PARITY=0 TIME_WRONG=1 ROUNDS=2 bash tools/r03_bench_ab.sh r05_bconv_ab packed bhalf 2>&1 | tee $OUT/bench_ab.txt
