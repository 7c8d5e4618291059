#!/bin/bash
# Builds rnnoise_amd/librnnoise_amd_<tag>.so = the product library with ONE source recompiled with extra flags (A/B builds for tools/ab_libs.sh):
#   tools/build_tag.sh <tag> <source under rnnoise_amd/csrc> [extra hipcc flags ...]      e.g.  tools/build_tag.sh c8 nn_mfma.hip -DRN_C1WD=8
# The product objects must be current (make -C rnnoise_amd/csrc).  Tagged libraries are git-ignored scratch.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/rnnoise_amd/csrc
tag=$1; src=$2; shift 2
base=$(basename "${src%.*}")
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
case $base in dsp_kernels|hp_kernel) F="$F -fno-slp-vectorize";; esac
mkdir -p /tmp/build_tag
( cd "$C" && /opt/rocm/bin/hipcc $F "$@" -c "$src" -o /tmp/build_tag/$base.$tag.o )
objs=""
for o in model tables batch host_io dropin hp_kernel state_kernels dsp_kernels nn_kernels nn_mfma nn_layers; do
  if [ $o = $base ]; then objs="$objs /tmp/build_tag/$base.$tag.o"; else objs="$objs $C/build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/exports.map -o "$R/rnnoise_amd/librnnoise_amd_$tag.so" $objs -lhsa-runtime64
python "$R/tools/kernel_resources.py" /tmp/build_tag/$base.$tag.o
