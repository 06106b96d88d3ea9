#!/bin/bash
# A/B of cache policies in k_ntt_row_ip: bench rates, then FETCH_SIZE per kernel at batch 10 for each variant
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r03r}; shift; mkdir -p $OUT
export TMPDIR=/tmp
ROUNDS=${ROUNDS:-2} tools/r03_bench_ab.sh $(basename $OUT) "$@" > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
cd /tmp
for v in default "$@"; do
  if [ $v = default ]; then unset HOMULATOR_HIP_LIB; else export HOMULATOR_HIP_LIB=$ROOT/ab_builds/libhm_$v.so; fi
  export HOMULATOR_BATCH=10
  for set in FETCH_SIZE "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $set | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/${v}_$n -o p --output-format csv -- python3 $ROOT/tools/pmc_op.py hmult 2 > $OUT/${v}_$n.log 2>&1 || echo "pass $v $n failed"
  done
  echo "== $v" >> $OUT/pmc.txt
  python3 $ROOT/tools/pmc_summary.py $OUT/${v}_FETCH_SIZE $OUT/${v}_WRITE_SIZE 2>&1 | grep -E "row_ip|bconv_col|ntt_row<false, 3>" >> $OUT/pmc.txt
done
cat $OUT/pmc.txt
