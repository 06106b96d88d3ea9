#!/bin/bash
# round 6: the ROW pass's LDS swizzle per pass direction / geometry (hm_lds_idx sets 1, 2; tools/lds_banks.py) against the build before it
# (ab_builds/head), interleaved on ONE box: parity gate first, then bench.py (value, one op at a time, the 50-limb sweep), then the LDS counters
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_lds_ab; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_ops.py tests/test_gpu_param_sets.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for r in 1 2 3; do
  for v in default head; do
    if [ $v = default ]; then unset HOMULATOR_HIP_LIB; else export HOMULATOR_HIP_LIB=$ROOT/ab_builds/$v/libhomulator_hip.so; fi
    timeout -k 10 200 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > $OUT/b_${v}_$r.json 2> $OUT/b_${v}_$r.err
    python3 -c "
import json;d=json.load(open('$OUT/b_${v}_$r.json'));print('$v', 'value', round(d['value'],1), 'generic', round(d['generic_chain_ops_per_s'],1), 'hrotate', round(d['hrotate']['ops_per_s'],1), 'one at a time', round(d['single_stream_ops_per_s'],1), 'sweep us', round(d['roofline']['us_per_launch'],2), 'in place', round(d['roofline']['in_place']['us_per_launch'],2), 'in op us/limb', round(d['roofline']['in_op']['us_per_limb'],4))"
  done
done
unset HOMULATOR_HIP_LIB
bash tools/pmc_kernels.sh r06_lds_ab/pmc > $OUT/pmc.txt 2>&1; grep -E "LDS_BANK" $OUT/pmc.txt
