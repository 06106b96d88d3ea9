#!/bin/bash
# round 4: ModDown conversion inside the merged transform (pass 9), residue in the conversion's epilogue (pass 10): parity, then bench A/B
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r04f}; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1000 python3 -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_ops.py tests/test_gpu_param_sets.py tests/test_gpu_chain.py tests/test_gpu_real_data.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; echo "PARITY FAILED"; exit 1; }
tail -2 $OUT/tests.log
for r in 1 2; do for v in "HOMULATOR_FUSE_MODDOWN=1" "HOMULATOR_FUSE_MODDOWN=0" "HOMULATOR_FUSE_BCONV=0" "X=1"; do
  env $v timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_${v}_$r.json 2> $OUT/bench_${v}_$r.err
  python3 -c "
import json;d=json.load(open('$OUT/bench_${v}_$r.json'));print('$v', round(d['value'],1), 'single', round(d['single_stream_ops_per_s'],1), 'launches', d['config']['launches_per_op'], [(k,round(t,1)) for k,_,t in d['stage_us_per_op_batched']], 'single:', [(k,round(t,1)) for k,_,t in d['stage_us']])"
done; done 2>&1 | tee $OUT/ab.txt
