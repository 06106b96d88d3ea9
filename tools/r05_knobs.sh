#!/bin/bash
# round 5: cheap knobs of the plan and of the bench, interleaved with the default on one box (bench.py --steps 200, value / sustained / one at a time):
#   fuse_moddown = 1 (the ModDown conversion inside the merged transform's first pass: level in round 4), 3 instances in flight, nip_small 0
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_knobs; mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline $EXTRA > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "
import json;d=json.load(open('$OUT/$name.json'));print('$name', round(d['value'],1), round(d['sustained_ops_per_s'],1), round(d['single_stream_ops_per_s'],1), 'hrotate', round(d['hrotate']['ops_per_s'],1), 'generic', round(d.get('generic_chain_ops_per_s') or 0,1))"; }
for r in 1 2; do
  EXTRA="" run default_$r A=0
  EXTRA="" run moddown_$r HOMULATOR_FUSE_MODDOWN=1
  EXTRA="--streams 3" run streams3_$r A=0
  EXTRA="" run nopack_$r HOMULATOR_PACK_BCONV_IN=0
  EXTRA="" run noipinv_$r HOMULATOR_FUSE_IP_INV=0
done
