#!/bin/bash
# round 5: pass 9 (ModDown conversion inside the merged transform's first pass) now that it takes the split-30 packed inputs too:
# parity of the new kernel combination, then default against fuse_moddown = 1 interleaved on one box (bench.py --steps 200)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_moddown; mkdir -p $OUT
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_ops.py -x -q -m gpu -k "packed or moddown or mix_sub_scale" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
run() { name=$1; shift; env "$@" timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "
import json;d=json.load(open('$OUT/$name.json'));print('$name', round(d['value'],1), round(d['sustained_ops_per_s'],1), round(d['single_stream_ops_per_s'],1), 'hrotate', round(d['hrotate']['ops_per_s'],1), 'generic', round(d.get('generic_chain_ops_per_s') or 0,1), 'launches', d['config']['launches_per_op'], [ (k,n,u) for k,n,u in (d.get('stage_us_per_op_batched') or [])])"; }
for r in 1 2 3; do
  run default_$r A=0
  run moddown_$r HOMULATOR_FUSE_MODDOWN=1
done
