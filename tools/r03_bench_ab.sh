#!/bin/bash
# bench.py for the default library and A/B builds, interleaved, N rounds each (ROUNDS, default 3), 400 steps: `value` and sustained rate
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r03j}; mkdir -p $OUT; shift
export TMPDIR=/tmp
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in default "$@"; do
    if [ $v = default ]; then unset HOMULATOR_HIP_LIB; else export HOMULATOR_HIP_LIB=$ROOT/ab_builds/libhm_$v.so; fi
    timeout -k 10 200 python3 bench.py --steps 400 --warmup 20 --no-cpu-baseline > $OUT/b_${v}_$r.json 2>/dev/null
    python3 -c "
import json;d=json.load(open('$OUT/b_${v}_$r.json'));print('$v', round(d['value'],1), round(d['sustained_ops_per_s'],1), round(d['single_stream_ops_per_s'],1), d.get('hip_library'))"
  done
done
