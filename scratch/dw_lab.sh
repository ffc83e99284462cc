#!/bin/bash
# where the 256 x 256 weight-gradient kernel's k-tile goes: the same loop without its LDS reads / its MFMAs / its DMA (lab builds, one device)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
for v in base NO_READS NO_MMA NO_DMA; do
  lib=scratch/libvqa_dw$v.so
  [ -f $lib ] || { echo "missing $lib"; continue; }
  VQA_HIP_LIB=$R/$lib timeout -k 10 200 python scratch/dw256_bench.py 2>&1 | grep "dw256=1" | sed "s/^/$v: /"
done
