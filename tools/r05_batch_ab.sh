#!/bin/bash
# round 5: ops per launch against the same-modulus grouping of the block map (G = 8 needs the ops of a batch to fill groups of 8: 10 ops do
# not, and fall back to groups of 2): 8 / 10 / 12 / 16 ops per launch, two instances, 20 launches per arm
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_batch; mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; b=$2; timeout -k 10 200 python3 bench.py --steps $((b * 20)) --warmup 20 --batch $b --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "
import json;d=json.load(open('$OUT/$name.json'));print('$name', round(d['value'],1), round(d['sustained_ops_per_s'],1), 'batch', d['config']['batch'], 'hrotate', round(d['hrotate']['ops_per_s'],1), [ (k,u) for k,n,u in (d.get('stage_us_per_op_batched') or [])])"; }
for r in 1 2; do
  for b in 10 8 16 12; do run b${b}_$r $b; done
done
