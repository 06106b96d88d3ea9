#!/bin/bash
# L2 behaviour of k_bconv_col for the default library and A/B builds: FETCH_SIZE and TCC hit / miss per launch (tools/bcol_ab.py, batch 10)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r03bp}; shift; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for v in default "$@"; do
  if [ $v = default ]; then unset HOMULATOR_HIP_LIB; else export HOMULATOR_HIP_LIB=$ROOT/ab_builds/$v/libhomulator_hip.so; fi
  for set in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    n=$(echo $set | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/${v}_$n -o p --output-format csv -- python3 $ROOT/tools/bcol_ab.py 10 2 > $OUT/${v}_$n.log 2>&1 || echo "$v $n failed"
  done
  echo "== $v"; python3 $ROOT/tools/pmc_summary.py $OUT/${v}_FETCH_SIZE $OUT/${v}_TCC_HIT_sum $OUT/${v}_TCC_EA0_RDREQ_sum 2>&1 | grep -E "bconv_col|k_bconv<"
done
