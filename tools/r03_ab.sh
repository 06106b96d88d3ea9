#!/bin/bash
# A/B of the one-launch transform variants in ab_builds/: parity of the fused tests, timing, HBM counters of the 50-limb sweep
#   tools/r03_ab.sh <tag> <variant> [<variant> ...]
set -e
ROOT=$(pwd); TAG=$1; shift; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for v in "$@"; do
  export HOMULATOR_HIP_LIB=$ROOT/ab_builds/libhm_$v.so HOMULATOR_NTT_FUSED=1
  timeout -k 10 300 python3 -m pytest tests/test_gpu_ntt_fused.py -x -q > $OUT/tests_$v.log 2>&1 || { tail -30 $OUT/tests_$v.log; echo "variant $v: PARITY FAILED"; }
  tail -1 $OUT/tests_$v.log
  timeout -k 10 300 python3 tools/ntt_fused_ab.py > $OUT/ab_$v.txt 2>&1; cat $OUT/ab_$v.txt
  (cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/sweep_${v}_$c -o p --output-format csv -- python3 $ROOT/tools/pmc_sweep.py > $OUT/sweep_${v}_$c.log 2>&1
  done)
  python3 $ROOT/tools/pmc_summary.py $OUT/sweep_${v}_FETCH_SIZE $OUT/sweep_${v}_WRITE_SIZE > $OUT/pmc_sweep_$v.txt 2>&1; cat $OUT/pmc_sweep_$v.txt
done
