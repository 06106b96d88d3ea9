ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_ewe; mkdir -p $OUT; export TMPDIR=/tmp
for v in ewe2 ewe4; do HOMULATOR_HIP_LIB=$ROOT/ab_builds/$v/libhomulator_hip.so timeout -k 10 200 python3 -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "ewe or tensor" 2>&1 | tail -1; done
for r in 1 2; do for v in default ewe2 ewe4; do
  if [ $v = default ]; then unset HOMULATOR_HIP_LIB; else export HOMULATOR_HIP_LIB=$ROOT/ab_builds/$v/libhomulator_hip.so; fi
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v', {k:(round(v['ops_per_s']),round(v['frac_of_hbm_peak'],3)) for k,v in d['elementwise'].items()}, round(d['value'],1))"
done; done
