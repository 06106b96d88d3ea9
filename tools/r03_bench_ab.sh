#!/bin/bash
# bench.py for the default library and A/B builds, interleaved, N rounds each (ROUNDS, default 3; set it INSIDE the gpurun command), 400 steps:
# `value`, sustained and one-at-a-time rate, and the library the process really mapped.  Every build first passes a parity gate (the kernel
# and op tests against the oracle with that build loaded): a variant that is fast because it is wrong is reported as such, not timed.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r03j}; mkdir -p $OUT; shift
export TMPDIR=/tmp
declare -A OK
for v in default "$@"; do
  if [ $v = default ]; then unset HOMULATOR_HIP_LIB; else export HOMULATOR_HIP_LIB=$ROOT/ab_builds/$v/libhomulator_hip.so; fi
  if [ "${PARITY:-1}" = 0 ]; then OK[$v]=2; elif timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ops.py -x -q -m gpu -k "not cli and not param" > $OUT/parity_$v.log 2>&1; then OK[$v]=1; else OK[$v]=0; echo "$v PARITY FAILED (timing-only ablation?): $(tail -1 $OUT/parity_$v.log)"; fi
done
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in default "$@"; do
    if [ ${OK[$v]} = 0 ] && [ "${TIME_WRONG:-0}" = 0 ]; then continue; fi
    if [ $v = default ]; then unset HOMULATOR_HIP_LIB; else export HOMULATOR_HIP_LIB=$ROOT/ab_builds/$v/libhomulator_hip.so; fi
    timeout -k 10 200 python3 bench.py --steps 400 --warmup 20 --no-cpu-baseline > $OUT/b_${v}_$r.json 2>/dev/null
    python3 -c "
import json;d=json.load(open('$OUT/b_${v}_$r.json'));print('$v', round(d['value'],1), round(d['sustained_ops_per_s'],1), round(d['single_stream_ops_per_s'],1), d.get('hip_library'), {0: 'WRONG RESULTS', 1: 'parity ok', 2: 'parity not checked'}[${OK[$v]}])"
  done
done
