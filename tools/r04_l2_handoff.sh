#!/bin/bash
# Round 4, capacity-controlled L2 hand-off experiment: k_ntt_fused (COL pass -> per-limb rendezvous -> ROW pass on the siblings' stores) at
# 8 and 16 limb-polys per launch (1-2 per XCD: < 3 MiB live per 4 MiB L2) and at 50 with the occupancy throttled to one workgroup per CU
# (two limb-polys in flight per XCD), against the two-kernel transform.  Counters per kernel, separate --pmc passes.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04_l2; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 python3 $ROOT/tools/handoff_ab.py > $OUT/timing.txt 2>&1; cat $OUT/timing.txt
run() {  # tag n fused lds
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_REQ_sum TCC_READ_sum"; do
    n=$(echo $set | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/$1_$n -o p --output-format csv -- python3 $ROOT/tools/pmc_handoff.py $2 $3 $4 > $OUT/$1_$n.log 2>&1 || echo "pass $1 $n failed"
  done
  echo "## $1: n=$2 fused=$3 extra_lds=$4" >> $OUT/summary.txt
  python3 $ROOT/tools/pmc_summary.py $OUT/$1_FETCH_SIZE $OUT/$1_WRITE_SIZE $OUT/$1_TCC_HIT_sum $OUT/$1_TCC_EA0_RDREQ_sum $OUT/$1_TCC_REQ_sum >> $OUT/summary.txt 2>&1
}
rm -f $OUT/summary.txt
run two8 8 0 0
run fused8 8 1 0
run two16 16 0 0
run fused16 16 1 0
run two50 50 0 0
run fused50 50 1 0
run fused50_t1 50 1 102400
run fused50_t2 50 1 28672
cat $OUT/summary.txt
