#!/bin/bash
# batched per-launch times of the hmult plan: fused NTT x key MAC variants against the separate transform + inner product
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r03e}; mkdir -p $OUT; shift
export TMPDIR=/tmp
echo "== fuse_hpip=0"; HOMULATOR_FUSE_HPIP=0 timeout -k 10 200 python3 tools/stage_times_batch.py 10 2>/dev/null | tee $OUT/st_nohpip.txt
echo "== default"; timeout -k 10 200 python3 tools/stage_times_batch.py 10 2>/dev/null | tee $OUT/st_default.txt
for v in "$@"; do echo "== $v"; HOMULATOR_HIP_LIB=$ROOT/ab_builds/$v/libhomulator_hip.so timeout -k 10 200 python3 tools/stage_times_batch.py 10 2>/dev/null | tee $OUT/st_$v.txt; done
