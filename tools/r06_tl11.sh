ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_tl11; mkdir -p $OUT; export TMPDIR=/tmp
HOMULATOR_HIP_LIB=$ROOT/ab_builds/tl11/libhomulator_hip.so timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "ntt" > $OUT/parity.log 2>&1; echo "parity rc=$?"; tail -3 $OUT/parity.log
for r in 1 2 3; do
  for v in default tl11; do
    if [ $v = default ]; then unset HOMULATOR_HIP_LIB; else export HOMULATOR_HIP_LIB=$ROOT/ab_builds/$v/libhomulator_hip.so; fi
    timeout -k 10 120 python3 tools/ntt_ab50.py 2>&1 | tail -1
  done
done
