#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r2_adamw_variants.log
for rep in 1 2; do for lib in "" scratch/libvqa_rev.so; do
  if [ -n "$lib" ]; then export VQA_HIP_LIB=$GRAFT_REPO_ROOT/$lib; else unset VQA_HIP_LIB; fi
  timeout -k 10 200 python scratch/adamw_bench.py cfg2_xattn 2>/dev/null | grep "M params" >> gpurun_out/r2_adamw_variants.log
done; done
cat gpurun_out/r2_adamw_variants.log
unset VQA_HIP_LIB
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_graph_gpu.py -m gpu -q -p no:cacheprovider -k "adamw or AdamW or optim" > gpurun_out/r2_t34.log 2>&1; grep -E "passed|failed" gpurun_out/r2_t34.log | tail -2
