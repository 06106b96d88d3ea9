#!/bin/bash
# timing-only ablation builds of the HIP backend into build_variants/ (git-ignored; travels with gpurun):
#   tools/ablate.sh <name> "<extra hipcc flags>"     e.g.  tools/ablate.sh nocompute "-DHM_ABL_NOCOMPUTE"
# use with HOMULATOR_HIP_LIB=build_variants/libhm_<name>.so python tools/ntt_scale.py
set -e
cd "$(dirname "$0")/../homulator_amd/csrc"
mkdir -p ../../build_variants
make OUT=../../build_variants/libhm_$1.so OBJ=../../build_variants/obj_$1 EXTRA="$2" >/dev/null
echo "built build_variants/libhm_$1.so"
