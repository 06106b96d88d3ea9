#!/bin/bash
# round 4: parity, timing and fabric counters of the persistent queue transform
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r04q}; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_ntt_queue.py -x -v -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; echo "PARITY FAILED"; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 400 python3 tools/ntt_queue_ab.py > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
cd /tmp
rm -f $OUT/pmc.txt
for cfg in "two50 50 0 0 2 0" "q50 50 1 512 2 0" "q50ip 50 1 512 2 1" "two512 512 0 0 2 0" "q512 512 1 512 2 0" "q512ip 512 1 512 2 1" "two512ip 512 0 0 2 1"; do
  set -- $cfg
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    c=$(echo $set | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/$1_$c -o p --output-format csv -- python3 $ROOT/tools/pmc_queue.py $2 $3 $4 $5 $6 12 > $OUT/$1_$c.log 2>&1 || echo "pass $1 $c failed"
  done
  echo "## $1: n=$2 geo=$3 wgs=$4 la=$5 inplace=$6" >> $OUT/pmc.txt
  python3 $ROOT/tools/pmc_summary.py $OUT/$1_FETCH_SIZE $OUT/$1_WRITE_SIZE $OUT/$1_TCC_HIT_sum >> $OUT/pmc.txt 2>&1
done
cat $OUT/pmc.txt
