#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r04d}; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_ntt_dma.py -x -v -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; echo "PARITY FAILED"; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 500 python3 -u tools/ntt_dma_ab.py > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
