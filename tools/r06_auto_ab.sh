#!/bin/bash
# round 6: hrotate's automorphisms folded into their readers (pass 12, fuse_auto: ModUp INTT + key product + final add gather) against the plan with the AUTO launch:
# parity (every hrotate test of the suite + the new kernel test), then stage times and bench.py's hrotate leg, interleaved on ONE box
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_auto_ab; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "hrotate or rotate or automorph or chain or sharded or param_sets" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for v in 1 0; do timeout -k 10 200 python3 tools/stage_times_batch.py 10 hrotate config_4.cfg 45 35 15 fuse_auto=$v; done
for r in 1 2 3; do
  for v in 1 0; do
    HOMULATOR_FUSE_AUTO=$v timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/b_${v}_$r.json 2> $OUT/b_${v}_$r.err
    python3 -c "
import json;d=json.load(open('$OUT/b_${v}_$r.json'));print('fuse_auto=$v', 'hrotate', round(d['hrotate']['ops_per_s'],1), [round(x,1) for x in d['hrotate']['ops_per_s_min_median_max']], 'value', round(d['value'],1))"
  done
done
