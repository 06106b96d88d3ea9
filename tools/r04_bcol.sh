#!/bin/bash
# two outputs per workgroup of the fused conversion + first pass against one: parity, kernel time alone, bench
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r04b}; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python3 -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_ops.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; echo "PARITY FAILED"; exit 1; }
tail -2 $OUT/tests.log
for r in 1 2; do for v in 2 1; do
  HOMULATOR_BCOL_OUTS=$v timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_${v}_$r.json 2> $OUT/bench_${v}_$r.err
  python3 -c "
import json;d=json.load(open('$OUT/bench_${v}_$r.json'));print('outs=$v', round(d['value'],1), 'single', round(d['single_stream_ops_per_s'],1), [(k,t) for k,_,t in d['stage_us_per_op_batched'] if k=='NTT_IP'], [(k,t) for k,_,t in d['stage_us'] if k=='NTT_IP'])"
done; done
