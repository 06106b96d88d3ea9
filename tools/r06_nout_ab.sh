#!/bin/bash
# one op at a time: outputs per workgroup of the fused conversion (1 | 2) x digits merged into one launch (0 | 1 | 2 = also with two outputs), interleaved
export TMPDIR=/tmp
for r in 1 2; do
for cfg in "0 1" "1 1" "2 1" "2 2" "2 0"; do set -- $cfg
HOMULATOR_BCOL_OUTS=$1 HOMULATOR_BCOL_MERGE=$2 timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --streams 1 --batch 1 --no-cpu-baseline > /tmp/b.json 2>/dev/null
python3 -c "
import json;d=json.load(open('/tmp/b.json'));print('outs=$1 merge=$2 one op at a time', round(d['value'],1), 'NTT_IP', [x[2] for x in d['stage_us'] if x[0]=='NTT_IP'], 'hrotate', round(d['hrotate']['ops_per_s'],1))"
done; done
