#!/bin/bash
# round 5: limb-polys per launch pair of the two-kernel transform (hand-off per pair = entries x 512 KiB) against the Infinity Cache (256 MiB):
# the in-op transform leg (1 150 limb-polys in one hm_ntt call) and the whole bench, default 448 against smaller pairs, interleaved
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_entries; mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "
import json;d=json.load(open('$OUT/$name.json'));print('$name', round(d['value'],1), round(d['sustained_ops_per_s'],1), 'single', round(d['single_stream_ops_per_s'],1), 'hrotate', round(d['hrotate']['ops_per_s'],1), 'in_op us/limb', round(d['roofline']['in_op']['us_per_limb'],4), [ (k,u) for k,n,u in (d.get('stage_us_per_op_batched') or [])])"; }
for r in 1 2; do
  for e in 448 256 128 64; do run e${e}_$r HOMULATOR_NTT_LAUNCH_ENTRIES=$e; done
done
