#!/bin/bash
# round 5: whole-op HBM bytes and VALU instructions of hmult 45/35/15 on the GENERIC arithmetic back-end (SURVEY 8d's chain: chain_bits 60 through
# HOMULATOR_CHAIN_BITS is not needed — forcing the back-end on the default chain runs the same kernels), at the timed region's launch shape
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_pmc_generic; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp HOMULATOR_ARITH=generic
ROUNDS=3; BATCH=10; INST=2
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $OUT/op_$n -o p --output-format csv -- python3 $ROOT/tools/pmc_op.py hmult $ROUNDS $BATCH $INST > $OUT/op_$n.log 2>&1 || true
done
OPS=$((ROUNDS * BATCH * INST))
(echo "# GENERIC arithmetic back-end (HOMULATOR_ARITH=generic): hmult 45/35/15, shape: batch $BATCH x instances $INST, $ROUNDS rounds = $OPS ops; KiB per op"
 python3 $ROOT/tools/pmc_op_sum.py $OUT/op_FETCH_SIZE FETCH_SIZE $OPS; python3 $ROOT/tools/pmc_op_sum.py $OUT/op_WRITE_SIZE WRITE_SIZE $OPS
 python3 $ROOT/tools/pmc_op_sum.py $OUT/op_SQ_INSTS_VALU SQ_INSTS_VALU $OPS) > $OUT/r05_pmc_whole_op_generic.txt 2>&1
cat $OUT/r05_pmc_whole_op_generic.txt
