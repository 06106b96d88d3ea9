#!/bin/bash
# round 6: digits of a small fused-conversion call merged into one launch (option bconv_col_merge) against one launch per digit width, one box, interleaved:
# parity first, then one op at a time (bench.py --streams 1 --batch 1) and the stage times
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_merge_ab; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_ops.py tests/test_gpu_param_sets.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for r in 1 2 3; do
  for m in 1 0; do
    HOMULATOR_BCOL_MERGE=$m timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --streams 1 --batch 1 --no-cpu-baseline > $OUT/b_${m}_$r.json 2> $OUT/b_${m}_$r.err
    python3 -c "
import json;d=json.load(open('$OUT/b_${m}_$r.json'));print('merge=$m one op at a time', round(d['value'],1), 'NTT_IP', [x[2] for x in d['stage_us'] if x[0]=='NTT_IP'], 'hrotate', round(d['hrotate']['ops_per_s'],1))"
  done
done
