#!/bin/bash
# round 5: the transform x key kernel of small launches on half tiles (k_ntt_row_ip8h): parity, then one op at a time with and without
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_half; mkdir -p $OUT
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_ops.py -x -q -m gpu -k "inner_product or hmult or hrotate" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
run() { name=$1; shift; env "$@" timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "
import json;d=json.load(open('$OUT/$name.json'));print('$name', round(d['value'],1), round(d['sustained_ops_per_s'],1), 'single', round(d['single_stream_ops_per_s'],1), 'hrotate', round(d['hrotate']['ops_per_s'],1), [ (k,u) for k,n,u in d['stage_us']])"; }
for r in 1 2 3; do
  run half_$r HOMULATOR_NIP_HALF=1
  run whole_$r HOMULATOR_NIP_HALF=0
done
