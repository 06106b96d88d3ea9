#!/bin/bash
# round 6 closing evidence at HEAD on ONE lease: the whole GPU suite on both arithmetic back-ends, the soak, the one-op kernel trace, per-kernel counters
# of the batched op, set A's op through the kernel trace (N = 2^15: the small-launch forms in use)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_final; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -q --timeout 400 > $OUT/suite.txt 2>&1; echo "suite rc=$?"; tail -2 $OUT/suite.txt
HOMULATOR_ARITH=generic timeout -k 10 900 python3 -m pytest tests -m gpu -q --timeout 400 > $OUT/suite_generic.txt 2>&1; echo "generic suite rc=$?"; tail -2 $OUT/suite_generic.txt
(echo "== mont32, batch 1 (one-launch transforms, merged conversion launches, two instances interleaved), 100000 hmults"; timeout -k 10 300 python3 tools/soak.py 100000 1 2>&1 | tail -3
 echo "== mont32, batch 4, 20000 launches = 80000 hmults"; timeout -k 10 300 python3 tools/soak.py 20000 4 2>&1 | tail -2
 echo "== generic back-end forced, batch 1, 40000 hmults"; HOMULATOR_ARITH=generic timeout -k 10 300 python3 tools/soak.py 40000 1 2>&1 | tail -2) > $OUT/soak.txt 2>&1; echo "soak rc=$?"; cat $OUT/soak.txt
sed -i 's/r05_trace1/r06_trace1/' tools/r05_trace1.sh; bash tools/r05_trace1.sh > $OUT/trace1.txt 2>&1; echo "trace1 rc=$?"; cat $OUT/trace1.txt
bash tools/pmc_kernels.sh r06_pmc_kernels > $OUT/pmc_kernels.log 2>&1; echo "pmc_kernels rc=$?"; tail -25 $OUT/pmc_kernels.log
