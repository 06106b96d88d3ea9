#!/bin/bash
# round 6: HBM traffic and vector instructions of hrotate (45/35/15, batch 10 x 2 instances, 3 rounds = 60 ops) with its automorphisms folded into their
# readers (pass 12) and with the AUTO launch (HOMULATOR_FUSE_AUTO=0): separate --pmc passes over tools/pmc_op.py, the program directly after `--`
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_auto_pmc; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for v in 1 0; do
  export HOMULATOR_FUSE_AUTO=$v
  for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d $OUT/f${v}_$n -o p --output-format csv -- python3 $ROOT/tools/pmc_op.py hrotate 3 10 2 > $OUT/f${v}_$n.log 2>&1 || echo "pass $v $n failed"
  done
  echo "== fuse_auto = $v"
  python3 $ROOT/tools/pmc_op_sum.py $OUT/f${v}_FETCH_SIZE FETCH_SIZE 60 | tail -1
  python3 $ROOT/tools/pmc_op_sum.py $OUT/f${v}_WRITE_SIZE WRITE_SIZE 60 | tail -1
  python3 $ROOT/tools/pmc_op_sum.py $OUT/f${v}_SQ_INSTS_VALU SQ_INSTS_VALU 60 | tail -1
done
