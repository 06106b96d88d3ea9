#!/bin/bash
# round 6: how busy the vector ALUs are in the contract's roofline kernel (the 50-limb sweep, k_ntt_fused8): issue counters of the same launches
# bench.py times (tools/pmc_sweep.py), + the sweep at 16 / 35 / 50 / 100 limb-polys (tools/ntt_ab50.py: the marginal cost of 256 more workgroups)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_sweep_issue; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p --output-format csv -- python3 $ROOT/tools/pmc_sweep.py > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
cd $ROOT
for r in 1 2; do timeout -k 10 120 python3 tools/ntt_ab50.py 2>&1 | tail -1; done
