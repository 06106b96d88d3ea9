#!/bin/bash
# round 6: every ROW round on the row's private twiddles (-DHM_NO_TWIST: no twist multiplies, 192 more twiddle words per row) against the shipped
# form (last ROW round on shared twiddles + 8 / 16 twist multiplies per thread), re-measured now that the op is known to be bound by instructions
# and package power, and twiddles are one 8-byte word: parity gate, then bench.py interleaved on ONE box
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_notwist_ab; mkdir -p $OUT
export TMPDIR=/tmp
HOMULATOR_HIP_LIB=$ROOT/ab_builds/notwist/libhomulator_hip.so timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_ops.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for r in 1 2 3; do
  for v in default notwist; do
    if [ $v = default ]; then unset HOMULATOR_HIP_LIB; else export HOMULATOR_HIP_LIB=$ROOT/ab_builds/$v/libhomulator_hip.so; fi
    timeout -k 10 200 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > $OUT/b_${v}_$r.json 2> $OUT/b_${v}_$r.err
    python3 -c "
import json;d=json.load(open('$OUT/b_${v}_$r.json'));print('$v', 'value', round(d['value'],1), [round(x,1) for x in d['value_min_median_max']], 'generic', round(d['generic_chain_ops_per_s'],1), 'hrotate', round(d['hrotate']['ops_per_s'],1), 'one at a time', round(d['single_stream_ops_per_s'],1), 'sweep us', round(d['roofline']['us_per_launch'],2), 'in op us/limb', round(d['roofline']['in_op']['us_per_limb'],4))"
  done
done
