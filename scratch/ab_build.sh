#!/bin/bash
# Builds an alternative bf16 library with extra -D flags for an in-call A/B (devices differ by up to 12 % in wall time: variants are
# only ever compared inside ONE gpurun call, on one device).   usage: scratch/ab_build.sh NAME -DFLAG=..   -> scratch/libvqa_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
objs=""
for f in vqa_model_builder_amd/csrc/*.hip; do
  o=/tmp/ab_${name}_$(basename $f .hip).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ivqa_model_builder_amd/csrc -Wno-unused-result "$@" -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libvqa_${name}.so $objs
echo scratch/libvqa_${name}.so
