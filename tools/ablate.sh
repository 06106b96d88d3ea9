#!/bin/bash
# A/B and timing-only ablation builds of the HIP backend into ab_builds/<name>/ (git-ignored; the .so files travel with gpurun, the
# objects stay in /tmp; delete a variant when its question is answered: every file here is pushed with every lease):
#   tools/ablate.sh <name> "<extra hipcc flags>"     e.g.  tools/ablate.sh nocompute "-DHM_ABL_NOCOMPUTE"
# builds the dispatcher and both arithmetic back-ends: use with HOMULATOR_HIP_LIB=ab_builds/<name>/libhomulator_hip.so
set -e
cd "$(dirname "$0")/../homulator_amd/csrc"
mkdir -p ../../ab_builds/$1
make -j8 LIB=../../ab_builds/$1 OBJ=/tmp/hm_ab_obj_$1 EXTRA="$2" >/dev/null
echo "built ab_builds/$1/"
