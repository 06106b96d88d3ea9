#!/bin/bash
# round 6: the scalar-register diet of the fused conversion kernels (one buffer descriptor + per-unit table rows) against the build before it
# (ab_builds/prediet), interleaved on ONE box: parity gate of the new kernels first, then bench.py at 400 steps, then the wide-digit sets
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_diet_ab; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_ops.py tests/test_gpu_param_sets.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
PARITY=0 ROUNDS=3 bash tools/r03_bench_ab.sh r06_diet_ab prediet 2>&1 | tee $OUT/bench_ab.txt
for r in 1 2; do
  for v in default prediet; do
    if [ $v = default ]; then unset HOMULATOR_HIP_LIB; else export HOMULATOR_HIP_LIB=$ROOT/ab_builds/$v/libhomulator_hip.so; fi
    timeout -k 10 300 python3 script/sweep.py --bench --set A,motivation --ops hmult --levels 28,20,16 --chains mont32,survey > $OUT/sweep_${v}_$r.txt 2> $OUT/sweep_${v}_$r.err; echo "sweep $v $r rc=$?"
  done
done
unset HOMULATOR_HIP_LIB
paste -d'|' $OUT/sweep_default_1.txt $OUT/sweep_prediet_1.txt $OUT/sweep_default_2.txt $OUT/sweep_prediet_2.txt | awk -F'|' '{split($1,a," "); split($2,b," "); split($3,c," "); split($4,d," "); print a[1],a[2],a[5],a[7], "diet", a[10], c[10], "prediet", b[10], d[10]}'
