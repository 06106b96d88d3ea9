#!/bin/bash
# round 5: the driver's K = 20 timed region (one launch per instance): instances x batch = 2 x 10 (the default so far) against 4 x 5, 5 x 4, 10 x 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_k20; mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "
import json;d=json.load(open('$OUT/$name.json'));print('$name', round(d['value'],1), 'sustained', round(d['sustained_ops_per_s'],1), 'single', round(d['single_stream_ops_per_s'],1), 'hrotate', round(d['hrotate']['ops_per_s'],1), 'generic', round(d['generic_chain_ops_per_s'],1), 'streams', d['config']['streams'], 'batch', d['config']['batch'])"; }
for r in 1 2 3; do
  run s2b10_$r
  run s4b5_$r --streams 4 --batch 5
  run s5b4_$r --streams 5 --batch 4
  run s10b2_$r --streams 10 --batch 2
done
