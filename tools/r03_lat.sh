#!/bin/bash
# one-op-at-a-time figures of bench.py, N rounds: value / single-stream rate / NTT_IP stage / sum of stages
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r03z}; mkdir -p $OUT; export TMPDIR=/tmp
for r in 1 2 3; do
  timeout -k 10 200 python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/b_$r.json 2>/dev/null
  python3 -c "
import json;d=json.load(open('$OUT/b_$r.json'));print(round(d['value'],1), round(d['single_stream_ops_per_s'],1), [(k,t) for k,_,t in d['stage_us']], round(sum(t for _,_,t in d['stage_us']),1))"
done
