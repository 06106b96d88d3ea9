#!/bin/bash
# second pass of the L2 hand-off experiment: the hand-off loads plain and nt instead of sc1 (ab_builds/libhm_mid_plain.so, libhm_mid_nt.so),
# at 8 limb-polys per launch (one per XCD) and at 50 throttled to one workgroup per CU
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04_l2b; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -f $OUT/summary.txt
for v in mid_plain mid_nt; do
  export HOMULATOR_HIP_LIB=$ROOT/ab_builds/libhm_$v.so
  (cd $ROOT && HOMULATOR_NTT_FUSED=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_ntt_fused.py -x -q 2>&1 | tail -1)
  for cfg in "f8 8 1 0" "f50t1 50 1 102400"; do
    set -- $cfg
    for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
      n=$(echo $set | cut -d' ' -f1)
      timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/${v}_$1_$n -o p --output-format csv -- python3 $ROOT/tools/pmc_handoff.py $2 $3 $4 > $OUT/${v}_$1_$n.log 2>&1 || echo "pass $v $1 $n failed"
    done
    echo "## $v $1: n=$2 fused=$3 extra_lds=$4" >> $OUT/summary.txt
    python3 $ROOT/tools/pmc_summary.py $OUT/${v}_$1_FETCH_SIZE $OUT/${v}_$1_WRITE_SIZE $OUT/${v}_$1_TCC_HIT_sum $OUT/${v}_$1_TCC_EA0_RDREQ_sum >> $OUT/summary.txt 2>&1
  done
done
cat $OUT/summary.txt
