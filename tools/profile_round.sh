#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r03      -> gpurun_out/prof_r03/*  (copy the summaries into profiles/ afterwards, then
#   python tools/make_roofline_inputs.py gpurun_out/prof_r03 r03)
# Counters go in their own runs with --kernel-trace only (one --pmc set per pass); the program goes directly after `--`.
# Every figure of the bench line that comes from a profiler is derived from THIS run; the whole-op passes use the timed
# region's launch shape (batch 10, 2 instances), the sweep passes the same 6 rotating buffer pairs as bench.py.
set -e
R=${1:-r04}
BATCH=${BATCH:-10}; INST=${INST:-2}; ROUNDS=${ROUNDS:-3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
$ROOT/tools/bflyrate > $OUT/${R}_bflyrate.txt 2>&1
# 1. kernel trace + stats of the bench command at the driver's settings
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/kt.log 2>&1
python3 $ROOT/tools/trace_summary.py $OUT/kt > $OUT/${R}_bench_kernel_trace_summary.txt 2>&1 || true
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/${R}_bench_kernel_stats.csv \; 2>/dev/null || true
# 2. HBM traffic of the roofline leg (50-limb forward sweep): FETCH_SIZE and WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/sweep_$c -o p --output-format csv -- python3 $ROOT/tools/pmc_sweep.py > $OUT/sweep_$c.log 2>&1
done
python3 $ROOT/tools/pmc_summary.py $OUT/sweep_FETCH_SIZE $OUT/sweep_WRITE_SIZE > $OUT/${R}_pmc_ntt_sweep.txt 2>&1 || true
# 3. issue / occupancy counters of the 128-limb forward + inverse launches
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $OUT/ntt128_$i -o p --output-format csv -- python3 $ROOT/tools/pmc_ntt.py > $OUT/ntt128_$i.log 2>&1 || true
done
python3 $ROOT/tools/pmc_summary.py $OUT/ntt128_1 $OUT/ntt128_2 $OUT/ntt128_3 $OUT/ntt128_4 $OUT/ntt128_5 > $OUT/${R}_pmc_ntt128.txt 2>&1 || true
# 4. whole-op HBM bytes and issue counters at the timed region's launch shape
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $OUT/op_$n -o p --output-format csv -- python3 $ROOT/tools/pmc_op.py hmult $ROUNDS $BATCH $INST > $OUT/op_$n.log 2>&1 || true
done
OPS=$((ROUNDS * BATCH * INST))
(echo "# hmult 45/35/15, shape: batch $BATCH x instances $INST, $ROUNDS rounds = $OPS ops; KiB per op"
 python3 $ROOT/tools/pmc_op_sum.py $OUT/op_FETCH_SIZE FETCH_SIZE $OPS; python3 $ROOT/tools/pmc_op_sum.py $OUT/op_WRITE_SIZE WRITE_SIZE $OPS
 python3 $ROOT/tools/pmc_op_sum.py $OUT/op_SQ_INSTS_VALU SQ_INSTS_VALU $OPS) > $OUT/${R}_pmc_whole_op.txt 2>&1 || true
ls $OUT | head -50
