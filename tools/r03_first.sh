#!/bin/bash
# first GPU pass of round 3: parity of the one-launch transform, A/B timing, HBM counters of the 50-limb sweep (fused / two-kernel)
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03a; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_ntt_fused.py tests/test_gpu_kernels.py -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
timeout -k 10 300 python3 tools/ntt_fused_ab.py > $OUT/ab_default.txt 2>&1; cat $OUT/ab_default.txt
for v in nt_in nt_io; do HOMULATOR_HIP_LIB=$ROOT/ab_builds/libhm_$v.so timeout -k 10 300 python3 tools/ntt_fused_ab.py > $OUT/ab_$v.txt 2>&1; cat $OUT/ab_$v.txt; done
cd /tmp
for f in 1 0; do for c in FETCH_SIZE WRITE_SIZE; do
  HOMULATOR_NTT_FUSED=$f timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/sweep_f${f}_$c -o p --output-format csv -- python3 $ROOT/tools/pmc_sweep.py > $OUT/sweep_f${f}_$c.log 2>&1
done; python3 $ROOT/tools/pmc_summary.py $OUT/sweep_f${f}_FETCH_SIZE $OUT/sweep_f${f}_WRITE_SIZE > $OUT/pmc_sweep_f$f.txt 2>&1; cat $OUT/pmc_sweep_f$f.txt; done
for v in nt_in nt_io; do for c in FETCH_SIZE WRITE_SIZE; do
  HOMULATOR_HIP_LIB=$ROOT/ab_builds/libhm_$v.so timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/sweep_${v}_$c -o p --output-format csv -- python3 $ROOT/tools/pmc_sweep.py > $OUT/sweep_${v}_$c.log 2>&1
done; python3 $ROOT/tools/pmc_summary.py $OUT/sweep_${v}_FETCH_SIZE $OUT/sweep_${v}_WRITE_SIZE > $OUT/pmc_sweep_$v.txt 2>&1; cat $OUT/pmc_sweep_$v.txt; done
