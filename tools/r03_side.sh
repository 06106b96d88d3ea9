#!/bin/bash
# one op at a time with / without side-by-side conversion launches; parity first
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r03w}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_kernels.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do
  for v in 1 0; do
    HOMULATOR_SIDE_LAUNCHES=$v timeout -k 10 200 python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/b_${v}_${r}.json 2>/dev/null
    python3 -c "
import json;d=json.load(open('$OUT/b_${v}_${r}.json'));print('side=$v', round(d['value'],1), round(d['single_stream_ops_per_s'],1), [(k,t) for k,_,t in d['stage_us'] if k=='NTT_IP'])"
  done
done
