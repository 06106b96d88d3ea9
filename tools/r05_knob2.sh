#!/bin/bash
# round 5: one more knob against the default on one box: fuse_bconv = 0 (ModUp conversions as launches of their own: pass 8 off)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_knob2; mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "
import json;d=json.load(open('$OUT/$name.json'));print('$name', round(d['value'],1), round(d['sustained_ops_per_s'],1), 'single', round(d['single_stream_ops_per_s'],1), 'hrotate', round(d['hrotate']['ops_per_s'],1), 'launches', d['config']['launches_per_op'], [ (k,u) for k,n,u in (d.get('stage_us_per_op_batched') or [])])"; }
for r in 1 2; do
  run default_$r A=0
  run nobconv_$r HOMULATOR_FUSE_BCONV=0
done
